"""How long until the 3-stream frame pipeline reaches its steady rate, and does it go 'cold' again?  Chunks of 20 frames (sync before and
after each), optional idle sleeps / other GPU work between them."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from real3dportrait_amd.frames import PipelinedClipRenderer
dev = torch.device("cuda", 0)
G, clip, dec, scene = bench.build_scene(torch, dev, n_frames=64)
cano, residuals, cams = scene
ring = torch.zeros(20, 512, 512, 3, dtype=torch.uint8, device=dev)
pipe = PipelinedClipRenderer(G, cano, residuals, cams, clip.ws, base_seed=clip.base_seed, n_streams=3)
a = torch.randn(4096, 4096, device=dev, dtype=torch.float16)
def chunk(tag):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20): pipe.render_u8(i, out=ring[i:i + 1])
    pipe.sync(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-34s %7.1f frames/s" % (tag, 20 / dt)); sys.stdout.flush()
for i in range(5): pipe.render_u8(i, out=ring[i:i + 1])
pipe.sync()
for c in range(5): chunk("chunk %d after 5 warm-up frames" % c)
time.sleep(0.05); chunk("after 50 ms idle")
time.sleep(0.5); chunk("after 500 ms idle"); chunk("next")
for _ in range(200): a @ a
chunk("after 200 matmuls"); chunk("next")
torch.cuda.synchronize(); time.sleep(0.5)
for _ in range(200): a @ a
chunk("after 500 ms idle + 200 matmuls")
