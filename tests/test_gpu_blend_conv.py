"""GPU parity (`-m gpu`) of r3d_conv_forward_blend (round 6): `x = torch.cat([x * person_occlusion, x_bg * (1 - person_occlusion)], dim=1)` fused into
the 1x1 conv that opens fuse_fg_bg_convs (modules/real3d/super_resolution/sr_with_ref.py:113-114, the Sequential of :56-62).

Two checks per case: the fused kernel is BIT-IDENTICAL to the two-step form it replaces (r3d_blend_cat_to_split + r3d_conv_forward: the same operand
arithmetic, the same MFMA order), and both agree with torch float64 at the fp32-class tolerance of the conv layers."""
import numpy as np
import pytest

from test_gpu_parity import T, torch_cuda  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def _cb8(torch, x):
    """NCHW fp32 -> the channel-blocked [N, C/8, H, W, 8] layout a Conv2d(out_format='cb8') writes, tagged like one (bound = max|x| per sample)."""
    N, C, H, W = x.shape
    y = x.view(N, C // 8, 8, H, W).permute(0, 1, 3, 4, 2).contiguous()
    y._r3d_fmt = "cb8"
    y._r3d_bound, y._r3d_depth = x.abs().amax(dim=(1, 2, 3)).contiguous(), 0
    return y


@pytest.mark.parametrize("N,Ca,Cb,Cout,H,W,slope,k", [(1, 256, 256, 64, 64, 48, 0.01, 0), (2, 24, 40, 12, 37, 21, None, 0), (1, 64, 64, 256, 16, 16, 0.2, 9),
                                                     (3, 8, 120, 64, 20, 33, 0.01, -11), (1, 128, 64, 128, 256, 256, 0.01, 0)])
@pytest.mark.parametrize("out_format", ["nchw", "cb8"])
def test_blend_conv_bit_identical_to_two_steps_and_vs_fp64(torch_cuda, N, Ca, Cb, Cout, H, W, slope, k, out_format):
    """Cin 512 / 64 / 128 / 192 (one ring of four stages, many), batch, ragged tiles, Cout padded to 128 (12, 64) and two cout tiles (256), sources
    scaled by 2^k (the fold's in-multiplier is a power of two: the fused operand is the same bits)."""
    torch = torch_cuda
    from real3dportrait_amd import synth
    from real3dportrait_amd.superresolution import Conv2d, _fold_single, blend_cat
    if out_format == "cb8" and Cout % 8:
        pytest.skip("blocked output needs Cout % 8 == 0")
    Cin = Ca + Cb
    a = T(torch, synth.hash_unitvar(31, (N, Ca, H, W), stream=1) * np.float32(2.0 ** k))
    b = T(torch, synth.hash_unitvar(31, (N, Cb, H, W), stream=2) * np.float32(2.0 ** k) * np.float32(0.7))
    m = T(torch, (np.abs(synth.hash_unitvar(31, (N, 1, H, W), stream=3)) * np.float32(0.6)).clip(0, 1))
    m[:, :, : H // 3] = 1.0
    m[:, :, -(H // 4):] = 0.0
    c = Conv2d(Cin, Cout, 1, 1, padding=0).cuda()
    c.precision = "f16x3"
    with torch.no_grad():
        c.weight.copy_(T(torch, synth.hash_unitvar(31, (Cout, Cin, 1, 1), stream=4) / np.float32(np.sqrt(Cin))))
        c.bias.copy_(T(torch, synth.hash_unitvar(31, (Cout,), stream=5)))
    a8, b8 = _cb8(torch, a), _cb8(torch, b)
    assert c.can_blend(a8, b8)
    _fold_single(c, N, a.device, [a8._r3d_bound, b8._r3d_bound], negative_slope=slope)
    y2 = c(blend_cat(a8, b8, m, c, _folded_head=c), negative_slope=slope, out_format=out_format)
    y1 = c(None, negative_slope=slope, out_format=out_format, _folded=True, _blend=(a8, b8, m))
    torch.cuda.synchronize()
    assert y1.shape == y2.shape and torch.equal(y1, y2), "fused blend conv differs from blend_cat_to_split + conv by %.3e" % (y1 - y2).abs().max().item()
    if out_format == "cb8":
        y1 = y1.permute(0, 1, 4, 2, 3).reshape(N, Cout, H, W)
    xd = torch.cat([a.double() * m.double(), b.double() * (1 - m.double())], dim=1).cpu()
    ref = torch.nn.functional.conv2d(xd, c.weight.detach().double().cpu(), c.bias.detach().double().cpu())
    if slope is not None:
        ref = torch.nn.functional.leaky_relu(ref, slope)
    e = (y1.cpu().double() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    print("blend conv N=%d %d+%d->%d %dx%d -> %s: %.2e of max|ref|, bit-identical to the two-step form" % (N, Ca, Cb, Cout, H, W, out_format, e))
    assert e <= 2e-6


def test_blend_conv_inside_a_stack_feeds_the_next_layer(torch_cuda):
    """ConvStack.forward(None, _blend=...) on the shape of fuse_fg_bg_convs (512 -> 64 1x1, LeakyReLU, 64 -> 256 3x3, LeakyReLU, 256 -> 256 3x3): the fused first
    layer writes the SPLIT operand of the second; the stack's output equals the two-step flow bit for bit at both precisions of the inner layers."""
    torch = torch_cuda
    from torch import nn
    from real3dportrait_amd import synth
    from real3dportrait_amd.superresolution import ConvStack, blend_cat
    N, H, W = 2, 48, 32
    seq = nn.Sequential(nn.Conv2d(512, 64, 1, 1, padding=0), nn.LeakyReLU(), nn.Conv2d(64, 256, 3, 1, padding=1), nn.LeakyReLU(), nn.Conv2d(256, 256, 3, 1, padding=1))
    with torch.no_grad():
        for j, mod in enumerate(seq):
            if isinstance(mod, nn.Conv2d):
                mod.weight.copy_(torch.from_numpy(synth.hash_unitvar(41, tuple(mod.weight.shape), stream=j) / np.float32(np.sqrt(mod.weight[0].numel()))))
                mod.bias.copy_(torch.from_numpy(synth.hash_unitvar(41, tuple(mod.bias.shape), stream=20 + j) * np.float32(0.1)))
    a = T(torch, synth.hash_unitvar(42, (N, 256, H, W), stream=1))
    b = T(torch, synth.hash_unitvar(42, (N, 256, H, W), stream=2))
    m = T(torch, np.abs(synth.hash_unitvar(42, (N, 1, H, W), stream=3)).clip(0, 1))
    xd = torch.cat([a.double() * m.double(), b.double() * (1 - m.double())], dim=1).cpu()
    ref = seq.double()(xd)
    for prec, tol in (("f16x3", 4e-6), ("f16mx", 1e-4)):
        stack = ConvStack.from_torch(seq.float()).cuda()
        for mod in stack:
            if hasattr(mod, "precision"):
                mod.precision = prec
        a8, b8 = _cb8(torch, a), _cb8(torch, b)
        head = stack.fold_for_input(N, a.device, [a8._r3d_bound, b8._r3d_bound])
        y2 = stack(blend_cat(a8, b8, m, stack, _folded_head=head))
        y1 = stack(None, _blend=(a8, b8, m))
        torch.cuda.synchronize()
        assert torch.equal(y1, y2), prec
        e = (y1.cpu().double() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
        print("fuse_fg_bg_convs-shaped stack with the fused blend, %s: %.2e of max|ref|" % (prec, e))
        assert e <= tol, prec


@pytest.mark.parametrize("prec,N,Cin,Ca,Cb,H,W", [("f16x3", 1, 64, 256, 256, 64, 48), ("f16mx", 1, 64, 256, 256, 64, 48), ("f16mx", 2, 24, 32, 48, 37, 21), ("f16x3", 3, 16, 16, 128, 16, 20)])
def test_conv_into_concatenation_bit_identical_to_two_steps(torch_cuda, prec, N, Cin, Ca, Cb, H, W):
    """r3d_conv_forward_cat (the 1x1 torso_encoder writes `x_torso * (1 - alpha)` as the second part of the concatenated SPLIT / SPLIT_MX operand) +
    r3d_blend_cat_to_split(b = NULL) (the first part) against conv -> fp32 -> r3d_blend_cat_to_split: every byte of both planes equal -- hi words, lo words or
    e5m2 records -- on whole and ragged tiles, batches, Cout one / two cout blocks."""
    torch = torch_cuda
    from real3dportrait_amd import synth
    from real3dportrait_amd.superresolution import Conv2d, blend_cat, chain_fold
    hid = T(torch, synth.hash_unitvar(51, (N, Cin, H, W), stream=1) * np.float32(3.0))
    a = T(torch, synth.hash_unitvar(51, (N, Ca, H, W), stream=2))
    m = T(torch, np.abs(synth.hash_unitvar(51, (N, 1, H, W), stream=3)).clip(0, 1))
    tconv, head = Conv2d(Cin, Cb, 1, 1, padding=0).cuda(), Conv2d(Ca + Cb, 32, 3, 1, padding=1).cuda()
    tconv.precision = head.precision = prec
    with torch.no_grad():
        tconv.weight.copy_(T(torch, synth.hash_unitvar(51, (Cb, Cin, 1, 1), stream=4) / np.float32(np.sqrt(Cin))))
        tconv.bias.copy_(T(torch, synth.hash_unitvar(51, (Cb,), stream=5)))
    a8 = _cb8(torch, a)
    tconv.prepare(N, hid.device); head.prepare(N, hid.device)
    tconv._depth_in = head._depth_in = 0
    bh = hid.abs().amax(dim=(1, 2, 3)).contiguous()
    chain_fold([tconv.chain_op(-1), head.chain_op(-2, 0)], N, [bh, a8._r3d_bound])
    xt = tconv(hid, out_format="cb8", _folded=True)
    ref = blend_cat(a8, xt, m, head, _folded_head=head)
    xs = torch.full((N, 2, (Ca + Cb) // 8, H, W, 8), float("nan"), device=hid.device, dtype=torch.float16)
    xs._r3d_fmt, xs._r3d_for = ref._r3d_fmt, head
    assert ref._r3d_fmt == ("split_mx" if prec == "f16mx" else "split")
    tconv.forward_cat(hid, xs, Ca, m, True, head)
    blend_cat(a8, None, m, head, _folded_head=head, _b_channels=Cb, _dst=xs)
    torch.cuda.synchronize()
    same = torch.equal(xs.view(torch.int16), ref.view(torch.int16))
    print("conv into the concatenation (%s) N=%d %d -> %d + %d, %dx%d: %s" % (prec, N, Cin, Ca, Cb, H, W, "every byte equal" if same else "DIFFERENT"))
    assert same


def test_blend_conv_argument_errors(torch_cuda):
    """The C ABI refuses what the kernel does not cover (a channel sum that is not a multiple of 64, a missing pointer) with R3D_ERR_INVALID_ARG and a message."""
    torch = torch_cuda
    from real3dportrait_amd import _lib
    lib = _lib.load()
    t = torch.zeros(4096, device="cuda")
    p = _lib.ptr(t)
    rc = lib.r3d_conv_forward_blend(p, p, None, 1, 8, 40, 64, 4, 4, p, p, p, 0, 0.0, 1.0, -1.0, p, 0, None, 0, None, _lib.stream_ptr())
    assert rc != 0 and b"multiple of 64" in lib.r3d_last_error()
    rc = lib.r3d_conv_forward_blend(p, p, None, 1, 32, 32, 64, 4, 4, p, None, p, 0, 0.0, 1.0, -1.0, p, 0, None, 0, None, _lib.stream_ptr())
    assert rc != 0 and b"bad argument" in lib.r3d_last_error()
