"""Time the torso/background fusion convs at the reference size (256 x 256, sr_with_ref.py:24-63)."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from real3dportrait_amd import synth
from real3dportrait_amd.superresolution import Conv2d, ConvStack, SynthesisBlockNoUp, blend_cat

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
R = 256
dev = "cuda"
stacks = {}
for n, (k, plan) in enumerate(synth.FUSION_STACKS.items()):
    mods = []
    for (ci, co, ks, lrelu) in plan:
        mods.append(Conv2d(ci, co, ks, 1, padding=ks // 2))
        if lrelu:
            mods.append(torch.nn.LeakyReLU())
    stacks[k] = ConvStack(*mods).to(dev)
blk = SynthesisBlockNoUp(256, 256, w_dim=512, resolution=R, img_channels=3, is_last=False, conv_clamp=None).to(dev)
g = torch.Generator(device=dev).manual_seed(1)
r = lambda *s: torch.randn(*s, device=dev, generator=g)
x_head, hid, bg, rgb, rgb_t = r(1, 256, R, R), r(1, 64, R, R), r(1, 3, R, R), r(1, 3, R, R), r(1, 3, R, R)
alpha, occ = torch.rand(1, 1, R, R, device=dev), torch.rand(1, 1, R, R, device=dev)
ws = torch.ones(1, 3, 512, device=dev)


def step():
    x_torso = stacks["torso_encoder"](hid)
    x_bg = stacks["bg_encoder"](bg)
    rgb1 = rgb * alpha + rgb_t * (1 - alpha)
    x1 = stacks["fuse_head_torso_convs"](blend_cat(x_head, x_torso, alpha, stacks["fuse_head_torso_convs"]))
    x2, rgb2 = blk(x1, rgb1, ws, noise_mode="none")
    return stacks["fuse_fg_bg_convs"](blend_cat(x2, x_bg, occ, stacks["fuse_fg_bg_convs"])), rgb2


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / iters * 1e3
# algorithmic flops: 2 * Cin * Cout * k*k * H*W per conv (real channels)
fl = 0
for plan in synth.FUSION_STACKS.values():
    fl += sum(2 * ci * co * k * k * R * R for ci, co, k, _ in plan)
fl += 2 * (2 * 256 * 256 * 9 * R * R) + 2 * 256 * 3 * R * R
print("fusion stacks @256: %.3f ms / frame, %.1f GFLOP algorithmic -> %.1f TFLOP/s" % (ms, fl / 1e9, fl / ms / 1e9))
