"""GPU parity (`-m gpu`) of the throughput SR precision 'f16mx' on every test that takes the library default.

The library default is the fp32-class 'f16x3' (round 4: superresolution.py "Precision policy"); bench.py and frames.ClipRenderer(precision=
'throughput') select 'f16mx' by name.  The tests of test_gpu_parity / test_gpu_warp_sr / test_gpu_range_and_sizes that build SR modules
without naming a precision therefore exercise f16x3; this file runs the same test bodies again with R3D_SR_PRECISION=f16mx (read when a
block is constructed), so BOTH shipped precisions meet every reference golden, every bit-exactness property and the fused-path
equalities.  (The file keeps its round-3 name; then the roles were the other way round.)"""
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
PLAIN = [tw.test_warp_sr_forward_v2_golden, tw.test_torso_frame_fused_input_equals_unfused_sequence,
         tw.test_warp_sr_forward_v2_batch_of_two_equals_two_singles]


@pytest.mark.parametrize("fn", WITH_TORCH, ids=lambda f: f.__name__)
def test_f16mx(monkeypatch, torch_cuda, fn):
    monkeypatch.setenv("R3D_SR_PRECISION", "f16mx")
    fn(torch_cuda)


@pytest.mark.parametrize("fn", PLAIN, ids=lambda f: f.__name__)
def test_f16mx_warp(monkeypatch, fn):
    monkeypatch.setenv("R3D_SR_PRECISION", "f16mx")
    fn()


@pytest.mark.parametrize("tag", ["down", "up"])
def test_f16mx_sr_resize(monkeypatch, torch_cuda, tag):
    monkeypatch.setenv("R3D_SR_PRECISION", "f16mx")
    tp.test_sr_resize_golden(torch_cuda, tag)


def test_precision_policy(monkeypatch):
    """Library default = fp32-class f16x3; the throughput tier is selected by name (ClipRenderer(precision='throughput'), patch_model(precision=),
    set_sr_precision) or for a process by R3D_SR_PRECISION."""
    import torch
    from real3dportrait_amd import TriPlaneGenerator
    from real3dportrait_amd.frames import ClipRenderer
    from real3dportrait_amd.superresolution import DEFAULT_SR_PRECISION, THROUGHPUT_SR_PRECISION, SynthesisBlock, SynthesisBlockNoUp, set_sr_precision
    assert DEFAULT_SR_PRECISION == "f16x3" and THROUGHPUT_SR_PRECISION == "f16mx"
    monkeypatch.delenv("R3D_SR_PRECISION", raising=False)
    b = SynthesisBlock(32, 64, w_dim=512, resolution=32, img_channels=3, is_last=False, conv_clamp=None)
    assert b.precision == "f16x3" and b._prec() == 1
    G = TriPlaneGenerator()
    z = torch.zeros(1)
    ClipRenderer(G, z, None, z, z)
    assert G.superresolution.block0.precision == "f16x3"                       # None: left alone
    ClipRenderer(G, z, None, z, z, precision="throughput")
    assert G.superresolution.block0.precision == G.superresolution.block1.precision == "f16mx"
    set_sr_precision(G.superresolution, "f32")
    assert G.superresolution.block1.precision == "f32"
    with pytest.raises(ValueError):
        set_sr_precision(G.superresolution, "fp8")
    n = SynthesisBlockNoUp(64, 64, w_dim=512, resolution=32, img_channels=3, is_last=False, conv_clamp=None)
    n.precision = "f16mx"
    assert n._prec() == 2 and n.wants_mx()      # round 4: SynthesisBlockNoUp has the fp8 path too (conv0 on R3D_FMT_SPLIT_MX inputs, conv1)
    from real3dportrait_amd.superresolution import Conv2d
    c3, c1 = Conv2d(64, 128, 3, 1, padding=1), Conv2d(64, 128, 1, 1, padding=0)
    assert c3.precision == "f16x3" and not c3.wants_mx()
    set_sr_precision(torch.nn.Sequential(c3, c1), "f16mx")
    assert c3.wants_mx() and not c1.wants_mx()  # the 1x1 conv has no fp8 path
    monkeypatch.setenv("R3D_SR_PRECISION", "f16mx")
    assert SynthesisBlock(32, 64, w_dim=512, resolution=32, img_channels=3, is_last=False, conv_clamp=None).precision == "f16mx"
    ClipRenderer(G, z, None, z, z, precision="throughput")
    assert G.superresolution.block0.precision == "f16mx"
