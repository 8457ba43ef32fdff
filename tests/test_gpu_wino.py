"""GPU parity (`-m gpu`) of the Winograd F(2,3) form of the plain 3x3 convolutions (csrc/r3d_sr_wino.h, round 6).

Which calls take it is a property of the shape and the process-wide switch R3D_CONV_WINO (read once per process: 0 never | 1 both precisions |
2 f16mx only | 3 f16x3 only, the default): whole 16 x 16 output tiles, Cin % 16 == 0, a plain SPLIT operand.  So
  * in-process tests here use shapes that qualify and the operands the default sends to it (f16x3 blocks; any conv layer fed with fp32 / plain SPLIT),
    against float64 -- the reference's semantics: modules/eg3ds/models/networks_stylegan2.py:37-94 (modulated_conv2d) and torch.nn.Conv2d of
    modules/real3d/super_resolution/sr_with_ref.py:24-63;
  * the f16mx instantiation (not a default: it is slower than the direct kernel) and the DIRECT f16x3 kernel (no longer what the default
    runs on these shapes) are kept from rotting by re-running existing test files in a subprocess under R3D_CONV_WINO=1 / 0."""
import os
import subprocess
import sys

import numpy as np
import pytest

from test_gpu_parity import SR_TOL, T, load_block, torch_cuda  # noqa: F401  (fixture)
from test_gpu_range_and_sizes import _block_fp64

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("N,Cin,Cout,H,W,slope", [(1, 16, 128, 16, 16, None), (2, 48, 72, 32, 48, 0.01), (1, 256, 256, 64, 32, 0.2), (3, 32, 12, 16, 64, None),
                                                   (1, 7, 64, 48, 16, 0.2)])
def test_conv_layer_on_the_winograd_kernel_vs_fp64(torch_cuda, N, Cin, Cout, H, W, slope):
    """r3d_conv_forward(k = 3) on shapes that take the Winograd kernel (fp32 input -> plain SPLIT operand): one K stage and many, a batch, Cout padded
    to 128 (72, 12), Cin padded to 16 (7), H != W, with and without the activation -- vs torch float64, at the fp32-class tolerance of the direct form."""
    torch = torch_cuda
    from real3dportrait_amd import synth
    from real3dportrait_amd.superresolution import Conv2d
    x = T(torch, synth.hash_unitvar(15, (N, Cin, H, W), stream=1))
    c = Conv2d(Cin, Cout, 3, 1, padding=1).cuda()
    c.precision = "f16x3"
    with torch.no_grad():
        c.weight.copy_(T(torch, synth.hash_unitvar(15, (Cout, Cin, 3, 3), stream=2) / np.float32(np.sqrt(Cin * 9))))
        c.bias.copy_(T(torch, synth.hash_unitvar(15, (Cout,), stream=3)))
    y = c(x, negative_slope=slope)
    ref = torch.nn.functional.conv2d(x.double().cpu(), c.weight.detach().double().cpu(), c.bias.detach().double().cpu(), padding=1)
    if slope is not None:
        ref = torch.nn.functional.leaky_relu(ref, slope)
    e = (y.cpu().double() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    print("conv layer on the Winograd kernel N=%d %d->%d %dx%d: %.2e of max|ref|" % (N, Cin, Cout, H, W, e))
    assert y.shape == ref.shape and e <= 2e-6


@pytest.mark.parametrize("k", [-12, 0, 14])
def test_sr_block_on_the_winograd_kernel_vs_fp64(torch_cuda, k):
    """SynthesisBlock at f16x3 whose conv1 output is whole 16 x 16 tiles (32 x 48): the block's second layer runs the Winograd kernel on the operand the
    up-sampling conv's epilogue wrote; inputs scaled by 2^k (the transform halves V and the epilogue takes the factor 4 out: exact), vs float64."""
    torch = torch_cuda
    from real3dportrait_amd import synth
    from real3dportrait_amd.superresolution import SynthesisBlock
    N, Cin, Cout, H, W = 2, 32, 128, 16, 24
    p = {kk: tuple(np.array(a) for a in v) for kk, v in synth.synth_sr_block(97, Cin, Cout, 512, 720).items()}
    x = synth.hash_unitvar(98, (N, Cin, H, W), stream=1) * np.float32(2.0 ** k)
    img = synth.hash_unitvar(98, (N, 3, H, W), stream=2) * np.float32(0.5)
    ws = np.ones((N, 3, 512), np.float32) + synth.hash_unitvar(98, (N, 3, 512), stream=3) * np.float32(0.2)
    blk = SynthesisBlock(Cin, Cout, w_dim=512, resolution=2 * H, img_channels=3, is_last=False, conv_clamp=None).cuda()
    load_block(torch, blk, p)
    blk.precision = "f16x3"
    xo, io = blk(T(torch, x), T(torch, img), T(torch, ws), noise_mode="none")
    rx, ri = _block_fp64(torch, p, torch.from_numpy(x), torch.from_numpy(img), torch.from_numpy(ws), True, None)
    ex = float((xo.cpu().double() - rx).abs().max() / rx.abs().max()); ei = float((io.cpu().double() - ri).abs().max() / max(1.0, float(ri.abs().max())))
    print("SR block on the Winograd kernel, input x 2^%d: x %.2e img %.2e of max|ref|" % (k, ex, ei))
    assert max(ex, ei) <= 4e-6


def _rerun(env, args):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"] + args, cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    tail = "\n".join(r.stdout.strip().splitlines()[-6:])
    assert r.returncode == 0, tail + r.stderr[-2000:]
    return tail


def test_f16mx_instantiation_forced_on_passes_the_sr_goldens_and_heavy_tails():
    """R3D_CONV_WINO=1: both SR blocks' conv1 on conv_wino_f16x3_kernel<true> (e5m2 records of V computed in the kernel, cross products on the fp8 MFMA):
    the f16mx goldens at their 5e-5 tier, the 2^k sweeps and the heavy-tail tiers."""
    print(_rerun({"R3D_CONV_WINO": "1"}, ["tests/test_gpu_mx.py", "tests/test_gpu_pinned_config.py::test_sr_block_heavy_tail",
                                            "tests/test_gpu_pinned_config.py::test_sr_block_dense_heavy_tail", "tests/test_gpu_pinned_config.py::test_benchmarked_frame_vs_oracle"]))


def test_direct_f16x3_kernel_still_passes_with_the_winograd_form_off():
    """R3D_CONV_WINO=0: the direct f16x3 conv (what rounds 1-5 shipped, and what shapes that are not whole tiles still run) on the tests of the fp32-class tier."""
    print(_rerun({"R3D_CONV_WINO": "0"}, ["tests/test_gpu_f16x3.py", "tests/test_gpu_wino.py::test_conv_layer_on_the_winograd_kernel_vs_fp64",
                                            "tests/test_gpu_wino.py::test_sr_block_on_the_winograd_kernel_vs_fp64"]))
