"""Soak: N frames rendered sequentially on one stream vs the same frames through the 3-stream pipeline, several passes; every uint8
frame must be identical (co-residency of ray and SR kernels of different frames must not change a single byte)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from real3dportrait_amd.frames import PipelinedClipRenderer
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 10
G, clip, dec, scene = bench.build_scene(torch, dev, n_frames=n)
cano, residuals, cams = scene
ref = torch.stack([clip.render_u8(t).clone() for t in range(n)])
torch.cuda.synchronize()
pipe = PipelinedClipRenderer(G, cano, residuals, cams, clip.ws, base_seed=clip.base_seed, n_streams=3)
ring = torch.zeros(n, 512, 512, 3, dtype=torch.uint8, device=dev)
bad = 0
for p in range(passes):
    ring.zero_()
    for t in range(n): pipe.render_u8(t, out=ring[t:t + 1])
    pipe.sync(); torch.cuda.synchronize()
    diff = (ring != ref).flatten(1).any(dim=1)
    bad += int(diff.sum())
    if diff.any():
        t = int(torch.nonzero(diff)[0]); d = (ring[t].int() - ref[t].int()).abs()
        print("pass %d: frame %d differs in %d bytes (max %d)" % (p, t, int((d > 0).sum()), int(d.max())))
print("soak: %d of %d pipelined frames differ from the sequential render" % (bad, n * passes))
