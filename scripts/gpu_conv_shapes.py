"""Time the plain 3x3 conv kernel (f16x3, via Conv2d) over (Cin, Cout, H) shapes: achieved fraction of the f16 peak per shape."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from real3dportrait_amd import synth, _lib
from real3dportrait_amd.superresolution import Conv2d
lib = _lib.load()
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for (ci, co, H) in ((256, 256, 256), (128, 128, 512), (128, 128, 256), (256, 256, 128), (256, 128, 256), (128, 256, 256), (512, 256, 256), (128, 128, 384), (64, 128, 512)):
    c = Conv2d(ci, co, 3, 1, padding=1).cuda()
    with torch.no_grad():
        c.weight.copy_(T(synth.hash_unitvar(1, (co, ci, 3, 3), stream=2) / np.float32(np.sqrt(ci * 9.0)))); c.bias.zero_()
    x = T(synth.hash_unitvar(2, (1, ci, H, H), stream=1))
    nxt = Conv2d(co, 128, 3, 1, padding=1).cuda()
    y = c(x, negative_slope=0.2, out_format="cb8")
    torch.cuda.synchronize()
    lib.r3d_profile_configure(1 << 1); lib.r3d_profile_reset()
    for _ in range(20): c(x, negative_slope=0.2, out_format="cb8")
    torch.cuda.synchronize()
    ms, cnt = ctypes.c_double(0), ctypes.c_int(0); lib.r3d_profile_read(1, ctypes.byref(ms), ctypes.byref(cnt)); lib.r3d_profile_configure(0)
    t = ms.value / max(1, cnt.value)
    fl = 2 * 9 * ci * co * H * H
    blocks = (H // 16) ** 2 * (co // 128)
    print("conv %3d -> %3d at %3d^2: %7.4f ms  %6.1f TFLOP/s alg = %.3f of peak  (%d blocks = %.2f rounds of 512, %d sub-stages)" % (ci, co, H, t, fl / t / 1e9, fl / t / 1e9 / 2500, blocks, blocks / 512.0, ci // 16 * 9 // 2))
