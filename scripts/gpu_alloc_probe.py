import sys, os, time
sys.path.insert(0, "/root/repo")
import torch, bench
from real3dportrait_amd.frames import PipelinedClipRenderer
dev = torch.device("cuda", 0)
G, clip, dec, scene = bench.build_scene(torch, dev, n_frames=64)
cano, residuals, cams = scene
ring = torch.zeros(40, 512, 512, 3, dtype=torch.uint8, device=dev)
pipe = PipelinedClipRenderer(G, cano, residuals, cams, clip.ws, base_seed=clip.base_seed, n_streams=3)
prev = 0
for i in range(40):
    t0 = time.perf_counter()
    pipe.render_u8(i, out=ring[i:i+1])
    dt = time.perf_counter() - t0
    st = torch.cuda.memory_stats()
    na = st.get("num_device_alloc", 0); seg = st.get("segment.all.current", 0)
    print("frame %2d: enqueue %.3f ms  device allocs so far %d (+%d) segments %d reserved %.0f MB" % (i, dt * 1e3, na, na - prev, seg, st["reserved_bytes.all.current"] / 1e6))
    prev = na
pipe.sync(); torch.cuda.synchronize()
