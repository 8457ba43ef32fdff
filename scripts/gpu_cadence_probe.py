"""Per-frame completion times of a short pipelined run (where do the first / last frames of a K = 20 run lose time?)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from real3dportrait_amd.frames import PipelinedClipRenderer
dev = torch.device("cuda", 0)
G, clip, dec, scene = bench.build_scene(torch, dev, n_frames=64)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ring = torch.zeros(K, 512, 512, 3, dtype=torch.uint8, device=dev)
cano, residuals, cams = scene
pipe = PipelinedClipRenderer(G, cano, residuals, cams, clip.ws, base_seed=clip.base_seed, n_streams=3)
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6): pipe.render_u8(i % K, out=ring[(i % K):(i % K)+1])
pipe.sync(); torch.cuda.synchronize()
for rep in range(3):
    ev0 = torch.cuda.Event(enable_timing=True); evs = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
    torch.cuda.synchronize(); t0 = time.perf_counter(); ev0.record()
    for i in range(K):
        pipe.render_u8(i, out=ring[i:i+1])
        evs[i].record(pipe.streams[i % 3])
    pipe.sync(); torch.cuda.synchronize(); wall = time.perf_counter() - t0
    done = sorted(ev0.elapsed_time(e) for e in evs)
    gaps = [done[0]] + [done[i] - done[i-1] for i in range(1, K)]
    print("rep %d wall %.3f ms (%.4f/frame); completion gaps ms: %s" % (rep, wall * 1e3, wall * 1e3 / K, " ".join("%.2f" % g for g in gaps)))
