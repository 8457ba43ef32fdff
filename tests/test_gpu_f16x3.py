"""GPU parity (`-m gpu`) of the fp32-class SR precision 'f16x3' on every test that takes the library default.

Since the end of round 3 the default SR precision is 'f16mx' (DESIGN 4.2c), so the tests of test_gpu_parity / test_gpu_warp_sr /
test_gpu_range_and_sizes that build SR modules without naming a precision exercise that path.  This file runs the same test bodies
again with R3D_SR_PRECISION=f16x3 (read when a block is constructed), so both shipped precisions meet every reference golden, every
bit-exactness property and the fused-path equalities."""
import pytest

import test_gpu_parity as tp
from test_gpu_parity import torch_cuda  # noqa: F401  (fixture)
import test_gpu_range_and_sizes as tr
import test_gpu_warp_sr as tw

pytestmark = pytest.mark.gpu

WITH_TORCH = [tp.test_synthesis_golden, tp.test_fusion_stacks_golden, tp.test_sr_rgb_skip_is_linear,
              tp.test_sr_concurrent_streams_are_bit_identical, tp.test_multi_stream_pipeline_is_bit_identical,
              tp.test_ray_kernel_split_output_equals_conversion_launch, tp.test_synthesis_batch_of_two_equals_two_singles,
              tr.test_sr_cfg5_golden, tr.test_fusion_stacks_full_size_golden, tr.test_fused_u8_epilogue_equals_reference_formula,
              tr.test_synthesis_mask_invalid_rays_golden, tr.test_bounds_are_upper_bounds_and_stored_operands_fit_fp16]
PLAIN = [tw.test_warp_sr_forward_v2_golden, tw.test_torso_frame_fused_input_equals_unfused_sequence,
         tw.test_warp_sr_forward_v2_batch_of_two_equals_two_singles]


@pytest.mark.parametrize("fn", WITH_TORCH, ids=lambda f: f.__name__)
def test_f16x3(monkeypatch, torch_cuda, fn):
    monkeypatch.setenv("R3D_SR_PRECISION", "f16x3")
    fn(torch_cuda)


@pytest.mark.parametrize("fn", PLAIN, ids=lambda f: f.__name__)
def test_f16x3_warp(monkeypatch, fn):
    monkeypatch.setenv("R3D_SR_PRECISION", "f16x3")
    fn()


@pytest.mark.parametrize("tag", ["down", "up"])
def test_f16x3_sr_resize(monkeypatch, torch_cuda, tag):
    monkeypatch.setenv("R3D_SR_PRECISION", "f16x3")
    tp.test_sr_resize_golden(torch_cuda, tag)


def test_default_precision_is_f16mx_and_pinnable(monkeypatch):
    from real3dportrait_amd.superresolution import DEFAULT_SR_PRECISION, SynthesisBlock, SynthesisBlockNoUp
    assert DEFAULT_SR_PRECISION == "f16mx"
    monkeypatch.delenv("R3D_SR_PRECISION", raising=False)
    b = SynthesisBlock(32, 64, w_dim=512, resolution=32, img_channels=3, is_last=False, conv_clamp=None)
    assert b.precision == "f16mx" and b._prec() == 2
    n = SynthesisBlockNoUp(64, 64, w_dim=512, resolution=32, img_channels=3, is_last=False, conv_clamp=None)
    assert n._prec() == 1                       # no fp8 path in the block without up-sampling: computes as f16x3
    monkeypatch.setenv("R3D_SR_PRECISION", "f16x3")
    assert SynthesisBlock(32, 64, w_dim=512, resolution=32, img_channels=3, is_last=False, conv_clamp=None).precision == "f16x3"
