"""GPU parity (`-m gpu`) of the fp32-class SR precision 'f16x3' on every test that takes the library default.

The library default is 'f16mx' since round 5 (superresolution.py "Precision policy": its e5m2 activation records hold the 2e-4 tier on the
heavy-tail sweeps too); 'f16x3' (every product as three fp16 MFMA terms, <= 1.3e-6) is selected by name.  The tests of test_gpu_parity /
test_gpu_warp_sr / test_gpu_range_and_sizes that build SR modules without naming a precision therefore exercise f16mx; this file runs the same
test bodies again with R3D_SR_PRECISION=f16x3 (read when a block is constructed), so BOTH shipped precisions meet every reference golden,
every bit-exactness property and the fused-path equalities."""
import pytest

import test_gpu_parity as tp
from test_gpu_parity import torch_cuda  # noqa: F401  (fixture)
import test_gpu_range_and_sizes as tr
import test_gpu_warp_sr as tw

pytestmark = pytest.mark.gpu

WITH_TORCH = [tp.test_synthesis_golden, tp.test_fusion_stacks_golden, tp.test_to_plane_cnn_golden, tp.test_sr_rgb_skip_is_linear,
              tp.test_sr_concurrent_streams_are_bit_identical, tp.test_multi_stream_pipeline_is_bit_identical,
              tp.test_ray_kernel_split_output_equals_conversion_launch, tp.test_synthesis_batch_of_two_equals_two_singles,
              tr.test_sr_cfg5_golden, tr.test_fusion_stacks_full_size_golden, tr.test_to_plane_cnn_full_size_golden, tr.test_fused_u8_epilogue_equals_reference_formula,
              tr.test_synthesis_mask_invalid_rays_golden, tr.test_bounds_are_upper_bounds_and_stored_operands_fit_fp16]
PLAIN = [tw.test_warp_sr_forward_v2_golden, tw.test_warp_sr_two_stage_entry_golden, tw.test_warp_sr_forward_fuse_mode_v1_golden, tw.test_torso_frame_fused_input_equals_unfused_sequence,
         tw.test_warp_sr_forward_v2_batch_of_two_equals_two_singles]


@pytest.mark.parametrize("fn", WITH_TORCH, ids=lambda f: f.__name__)
def test_f16x3(monkeypatch, torch_cuda, fn):
    monkeypatch.setenv("R3D_SR_PRECISION", "f16x3")
    fn(torch_cuda)


@pytest.mark.parametrize("k", tr.SWEEP)
def test_f16x3_conv_stack_range_sweep(monkeypatch, torch_cuda, k):
    monkeypatch.setenv("R3D_SR_PRECISION", "f16x3")
    tr.test_conv_stack_range_sweep(torch_cuda, k)


@pytest.mark.parametrize("fn", PLAIN, ids=lambda f: f.__name__)
def test_f16x3_warp(monkeypatch, fn):
    monkeypatch.setenv("R3D_SR_PRECISION", "f16x3")
    fn()


@pytest.mark.parametrize("tag", ["down", "up"])
def test_f16x3_sr_resize(monkeypatch, torch_cuda, tag):
    monkeypatch.setenv("R3D_SR_PRECISION", "f16x3")
    tp.test_sr_resize_golden(torch_cuda, tag)


def test_precision_policy(monkeypatch):
    """Library default = 'f16mx' (round 5); the fp32-class tier 'f16x3' and the exact 'f32' are selected by name (ClipRenderer(precision=),
    patch_model(precision=), set_sr_precision) or for a process by R3D_SR_PRECISION."""
    import torch
    from real3dportrait_amd import TriPlaneGenerator
    from real3dportrait_amd.frames import ClipRenderer
    from real3dportrait_amd.superresolution import (DEFAULT_SR_PRECISION, FP32_CLASS_SR_PRECISION, THROUGHPUT_SR_PRECISION, Conv2d, SynthesisBlock,
                                                    SynthesisBlockNoUp, set_sr_precision)
    assert DEFAULT_SR_PRECISION == THROUGHPUT_SR_PRECISION == "f16mx" and FP32_CLASS_SR_PRECISION == "f16x3"
    monkeypatch.delenv("R3D_SR_PRECISION", raising=False)
    b = SynthesisBlock(32, 64, w_dim=512, resolution=32, img_channels=3, is_last=False, conv_clamp=None)
    assert b.precision == "f16mx" and b._prec() == 2 and b.wants_mx()
    G = TriPlaneGenerator()
    z = torch.zeros(1)
    ClipRenderer(G, z, None, z, z)
    assert G.superresolution.block0.precision == "f16mx"                       # None: left alone
    ClipRenderer(G, z, None, z, z, precision="f16x3")
    assert G.superresolution.block0.precision == G.superresolution.block1.precision == "f16x3"
    ClipRenderer(G, z, None, z, z, precision="throughput")
    assert G.superresolution.block0.precision == G.superresolution.block1.precision == "f16mx"
    set_sr_precision(G.superresolution, "f32")
    assert G.superresolution.block1.precision == "f32"
    with pytest.raises(ValueError):
        set_sr_precision(G.superresolution, "fp8")
    n = SynthesisBlockNoUp(64, 64, w_dim=512, resolution=32, img_channels=3, is_last=False, conv_clamp=None)
    assert n._prec() == 2 and n.wants_mx()      # SynthesisBlockNoUp has the 8-bit path too (conv0 on R3D_FMT_SPLIT_MX inputs, conv1)
    c3, c1 = Conv2d(64, 128, 3, 1, padding=1), Conv2d(64, 128, 1, 1, padding=0)
    assert c3.precision == "f16mx" and c3.wants_mx() and not c1.wants_mx()      # the 1x1 conv has no 8-bit path
    set_sr_precision(torch.nn.Sequential(c3, c1), "f16x3")
    assert not c3.wants_mx()
    monkeypatch.setenv("R3D_SR_PRECISION", "f16x3")
    assert SynthesisBlock(32, 64, w_dim=512, resolution=32, img_channels=3, is_last=False, conv_clamp=None).precision == "f16x3"
    ClipRenderer(G, z, None, z, z, precision="throughput")
    assert G.superresolution.block0.precision == "f16x3"


def test_mixed_precision_hand_off_degrades_to_plain_split(torch_cuda):
    """set_sr_precision / `.precision` allow per-module precisions: an f16x3 producer in front of an f16mx consumer hands over plain
    SPLIT (no 8-bit records) instead of tripping an assert, and the result is the all-f16x3 one within the f16mx tier."""
    torch = torch_cuda
    import numpy as np
    from real3dportrait_amd import synth
    from real3dportrait_amd.superresolution import SuperresolutionHybrid8XDC, SynthesisBlock
    b0 = SynthesisBlock(32, 128, w_dim=512, resolution=32, img_channels=3, is_last=False, conv_clamp=None).cuda()
    b1 = SynthesisBlock(128, 128, w_dim=512, resolution=64, img_channels=3, is_last=True, conv_clamp=None).cuda()
    for k, b in enumerate((b0, b1)):
        p = {kk: tuple(np.array(a) for a in v) for kk, v in synth.synth_sr_block(40 + k, b.in_channels, b.out_channels, 512, 700).items()}
        tp.load_block(torch, b, p)
    x = tp.T(torch, synth.hash_unitvar(44, (1, 32, 16, 16), stream=1))
    img = tp.T(torch, synth.hash_unitvar(44, (1, 3, 16, 16), stream=2) * np.float32(0.5))
    ws = torch.ones(1, 3, 512, device="cuda")
    outs = {}
    for p0, p1 in (("f16x3", "f16x3"), ("f16x3", "f16mx"), ("f16mx", "f16mx")):
        b0.precision, b1.precision = p0, p1
        b1.prepare(ws)
        from real3dportrait_amd.superresolution import chain_fold
        b0.prepare(ws)
        bx = torch.full((1,), float(x.abs().max()), device="cuda")
        chain_fold([b0.chain_op(-1), b1.chain_op(0)], 1, [bx])
        b0.out_format = "split_mx" if b1.wants_mx() else "split"
        x0, i0 = b0(x, img, ws, noise_mode="none", _next=b1, _folded=True)
        assert x0._r3d_fmt == ("split_mx" if (p0, p1) == ("f16mx", "f16mx") else "split")
        b0.out_format = "nchw"
        _, i1 = b1(x0, i0, ws, noise_mode="none")
        outs[(p0, p1)] = i1
    ref = outs[("f16x3", "f16x3")]
    for k, v in outs.items():
        assert (v - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item()), k
