"""Kernel list of the drop-in API path: TriPlaneGenerator.synthesis() per frame through patch_model()'d operators (bench.py's value_synthesis_api loop), for
rocprofv3 --kernel-trace --stats: which launches a frame costs beyond ClipRenderer's."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from real3dportrait_amd import patch_model
from real3dportrait_amd.frames import clone_generator_shell, frame_seed
dev = torch.device("cuda:0")
G, clip, dec, (cano, residuals, cams) = bench.build_scene(torch, dev, n_frames=64)
G_api = patch_model(clone_generator_shell(G))
G_api.renderer.noise_mode = "hash"
ws = torch.ones(1, 14, 512, device=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50


def frame(t):
    G_api.renderer.seed = frame_seed(clip.base_seed, t)
    G_api._last_planes = (cano + residuals[t % len(residuals)]).view(1, 96, 256, 256)
    return G_api.synthesis(ws, cams[t:t + 1], use_cached_backbone=True, noise_mode="none")


for t in range(10):
    frame(t)
torch.cuda.synchronize(); t0 = time.perf_counter()
for t in range(n):
    frame(t % 64)
torch.cuda.synchronize()
print("synthesis API: %.4f ms per frame over %d frames" % ((time.perf_counter() - t0) / n * 1e3, n))
