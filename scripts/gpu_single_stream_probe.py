import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from real3dportrait_amd.frames import PipelinedClipRenderer
dev = torch.device("cuda", 0)
G, clip, dec, scene = bench.build_scene(torch, dev, n_frames=64)
ring = torch.zeros(40, 512, 512, 3, dtype=torch.uint8, device=dev)
def single(tag):
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(40): clip.render_u8(i, out=ring[i:i+1])
    te = time.perf_counter() - t
    torch.cuda.synchronize(); print(tag, "single-stream: enqueue %.3f ms/frame, wall %.3f ms/frame" % (te / 40 * 1e3, (time.perf_counter() - t) / 40 * 1e3))
for _ in range(3): clip.render_u8(0, out=ring[0:1])
single("before pipe")
cano, residuals, cams = scene
pipe = PipelinedClipRenderer(G, cano, residuals, cams, clip.ws, base_seed=clip.base_seed, n_streams=3)
for i in range(6): pipe.render_u8(i, out=ring[i:i+1])
pipe.sync(); torch.cuda.synchronize()
t = time.perf_counter()
for i in range(40): pipe.render_u8(i, out=ring[i:i+1])
te = time.perf_counter() - t
pipe.sync(); torch.cuda.synchronize(); print("pipe: enqueue %.3f wall %.3f ms/frame" % (te / 40 * 1e3, (time.perf_counter() - t) / 40 * 1e3))
single("after pipe 1"); single("after pipe 2")
print(torch.cuda.memory_stats()["num_alloc_retries"], torch.cuda.memory_reserved() / 2**30)
