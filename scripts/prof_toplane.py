"""Time the per-frame plane producer tail at the reference size: to_plane_cnn 128^2 -> 256^2 + flips +
cano add + channel-last layout (segformer.py:691-700,721-729; secc_img2plane.py:76-77)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from real3dportrait_amd import ImportanceRenderer, synth
from real3dportrait_amd.superresolution import Conv2d, ConvStack
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
mods = []
for i, (ci, co, k, lrelu) in enumerate(synth.TO_PLANE_CNN):
    if i == synth.TO_PLANE_CNN_UP_BEFORE:
        mods.append(torch.nn.UpsamplingBilinear2d(scale_factor=2.))
    mods.append(Conv2d(ci, co, k, 1, padding=1))
    if lrelu:
        mods.append(torch.nn.LeakyReLU(0.01))
cnn = ConvStack(*mods).cuda()
ren = ImportanceRenderer(hp={})
feat = torch.randn(1, 256, 128, 128, device="cuda"); cano = torch.randn(1, 3, 32, 256, 256, device="cuda")
def step():
    return ren.prepare_planes(cano, add=cnn(feat), add_flip=ren.SECC_PLANE_FLIPS)
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(iters): step()
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / iters * 1e3
fl = 3 * 2 * 9 * 256 * 256 * 128 * 128 + 2 * 9 * 256 * 96 * 256 * 256
print("to_plane_cnn + flips + add + layout @128->256: %.3f ms / frame, %.1f GFLOP -> %.1f TFLOP/s algorithmic" % (ms, fl / 1e9, fl / ms / 1e9))
