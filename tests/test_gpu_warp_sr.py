"""GPU parity (`-m gpu`): the fused HIP SuperresolutionHybrid8XDC_Warp.forward (fuse mode v2) against the reference's own forward
run on CPU with the same stand-in torso network (tests/golden/warp_sr_a.npz, tests/warp_mock.py)."""
import numpy as np
import pytest

from conftest import load_golden
from test_gpu_parity import SR_TOL, T, load_block

pytestmark = pytest.mark.gpu


def test_warp_sr_forward_v2_golden():
    import torch
    assert torch.cuda.is_available()
    import warp_mock
    from real3dportrait_amd.sr_with_ref import SuperresolutionHybrid8XDC_Warp
    g = load_golden("warp_sr_a")
    sr = SuperresolutionHybrid8XDC_Warp(32, 512, 0, True, torso_model=warp_mock.MockTorso(),
                                        hparams={"htbsr_head_threshold": float(g["threshold"])}).cuda()
    warp_mock.load_warp_params(sr, lambda blk, p: load_block(torch, blk, p), to=lambda a: T(torch, a))
    i = {k: T(torch, v) for k, v in warp_mock.warp_inputs().items()}
    outs = []
    for _ in range(2):          # second call: clip constants (resized references, bg_encoder output, styles) come from the caches
        out, ret = sr(i["x"][:, :3].contiguous(), i["x"], i["ws"], i["ref_torso_rgb"], i["ref_bg_rgb"], i["weights_img"], None, None, None,
                      noise_mode="none")
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    assert set(ret) == {"deformed_torso_hid", "occlusion_2"}
    out = outs[0].cpu().numpy()
    tol = SR_TOL * max(1.0, np.abs(g["strided"]).max())
    assert out.shape == (1, 3, 512, 512)
    assert np.abs(out[:, :, ::4, ::4] - g["strided"]).max() <= tol
    assert np.abs(out[:, :, :96, :96] - g["corner"]).max() <= tol
    assert np.abs(out[:, :, -64:, -64:] - g["tail"]).max() <= tol
    assert abs(float(np.abs(out).mean()) - float(g["absmean"])) <= 1e-4
    # a new background image must invalidate the cached bg_encoder output
    bg2 = i["ref_bg_rgb"] * 0.5
    out2, _ = sr(i["x"][:, :3].contiguous(), i["x"], i["ws"], i["ref_torso_rgb"], bg2, i["weights_img"], None, None, None, noise_mode="none")
    assert not torch.equal(out2, outs[0])


@pytest.mark.parametrize("prec", ["f16mx", "f16x3"])
def test_warp_sr_forward_v2_round6_fusions_are_bit_identical(prec):
    """Round 6 moved two blend + concatenation steps into the convs next to them: torso_encoder writes its half of :104's operand itself
    (r3d_conv_forward_cat, R3D_FUSE_TORSO_CAT) and fuse_fg_bg_convs' 1x1 conv computes :113's operand while it stages it (r3d_conv_forward_blend,
    R3D_FUSE_BLEND).  Same arithmetic in the same order: the fused forward's image equals the round-5 sequence bit for bit, at both precisions."""
    import torch
    import warp_mock
    from real3dportrait_amd import sr_with_ref
    from real3dportrait_amd.superresolution import set_sr_precision
    g = load_golden("warp_sr_a")
    i = {k: T(torch, v) for k, v in warp_mock.warp_inputs().items()}
    outs = {}
    saved = (sr_with_ref._FUSE_TORSO_CAT, sr_with_ref._FUSE_BLEND)
    try:
        for cat, bl in ((True, True), (False, False), (True, False), (False, True)):
            sr_with_ref._FUSE_TORSO_CAT, sr_with_ref._FUSE_BLEND = cat, bl
            sr = sr_with_ref.SuperresolutionHybrid8XDC_Warp(32, 512, 0, True, torso_model=warp_mock.MockTorso(),
                                                            hparams={"htbsr_head_threshold": float(g["threshold"])}).cuda()
            warp_mock.load_warp_params(sr, lambda blk, p: load_block(torch, blk, p), to=lambda a: T(torch, a))
            set_sr_precision(sr, prec)
            out, _ = sr(i["x"][:, :3].contiguous(), i["x"], i["ws"], i["ref_torso_rgb"], i["ref_bg_rgb"], i["weights_img"], None, None, None, noise_mode="none")
            outs[(cat, bl)] = out.clone()
    finally:
        sr_with_ref._FUSE_TORSO_CAT, sr_with_ref._FUSE_BLEND = saved
    for k, v in outs.items():
        assert torch.equal(v, outs[(False, False)]), (prec, k, float((v - outs[(False, False)]).abs().max()))


def test_warp_sr_two_stage_entry_golden():
    """The reference's two-stage entry (sr_with_ref.py:164-218: infer_forward_stage1 -> infer_forward_stage2, unused by real3d_infer.py) on the
    mirror class against the reference's own outputs (tests/golden/warp_sr_two_stage_a.npz): block0's x, the dict's keys, the final image."""
    import torch
    import warp_mock
    from real3dportrait_amd.sr_with_ref import SuperresolutionHybrid8XDC_Warp
    g = load_golden("warp_sr_two_stage_a")
    sr = SuperresolutionHybrid8XDC_Warp(32, 512, 0, True, torso_model=warp_mock.MockTorso()).cuda()
    warp_mock.load_warp_params(sr, lambda blk, p: load_block(torch, blk, p), to=lambda a: T(torch, a))
    i = {k: T(torch, v) for k, v in warp_mock.warp_inputs().items()}
    ret = sr.infer_forward_stage1(i["x"][:, :3].contiguous(), i["x"], i["ws"], i["ref_torso_rgb"], i["ref_bg_rgb"], i["weights_img"], None, None, None,
                                  noise_mode="none")
    assert sorted(k for k in ret if not k.startswith("_")) == [str(k) for k in g["keys"]]
    x0 = ret["x"].cpu().numpy()
    assert x0.shape == (1, 256, 256, 256)
    assert np.abs(x0[:, ::8, ::8, ::8] - g["x0_strided"]).max() <= SR_TOL * max(1.0, np.abs(g["x0_strided"]).max())
    out, ret2 = sr.infer_forward_stage2(ret, noise_mode="none")
    assert ret2 is ret
    out = out.cpu().numpy()
    tol = SR_TOL * max(1.0, np.abs(g["strided"]).max())
    assert out.shape == (1, 3, 512, 512)
    e = max(np.abs(out[:, :, ::4, ::4] - g["strided"]).max(), np.abs(out[:, :, :96, :96] - g["corner"]).max(), np.abs(out[:, :, -64:, -64:] - g["tail"]).max())
    print("two-stage entry [%s]: final image err %.2e (tier %.1e)" % (sr.block0.precision, e, tol))
    assert e <= tol
    assert abs(float(np.abs(out).mean()) - float(g["absmean"])) <= 1e-4
    # the hand-off formats the fused forward sets per call are untouched by the two-stage entry
    assert sr.block0.out_format == "nchw" and sr.block1.out_format == "nchw" and sr.block1.return_x is True


def test_warp_sr_forward_fuse_mode_v1_golden():
    """htbsr_head_weight_fuse_mode = 'v1' (sr_with_ref.py:92-104, not a shipped configuration) on the mirror class, operator by operator, against
    the reference's forward under the same hparams (tests/golden/warp_sr_v1_a.npz); 'v3' stays a descriptive NotImplementedError."""
    import torch
    import warp_mock
    from real3dportrait_amd.sr_with_ref import SuperresolutionHybrid8XDC_Warp
    g = load_golden("warp_sr_v1_a")
    sr = SuperresolutionHybrid8XDC_Warp(32, 512, 0, True, torso_model=warp_mock.MockTorso(),
                                        hparams={"htbsr_head_threshold": float(g["threshold"]), "htbsr_head_weight_fuse_mode": "v1"}).cuda()
    warp_mock.load_warp_params(sr, lambda blk, p: load_block(torch, blk, p), to=lambda a: T(torch, a))
    i = {k: T(torch, v) for k, v in warp_mock.warp_inputs().items()}
    args = (i["x"][:, :3].contiguous(), i["x"], i["ws"], i["ref_torso_rgb"], i["ref_bg_rgb"], i["weights_img"], None, None, None)
    out, ret = sr(*args, noise_mode="none")
    out = out.cpu().numpy()
    tol = SR_TOL * max(1.0, np.abs(g["strided"]).max())
    e = max(np.abs(out[:, :, ::4, ::4] - g["strided"]).max(), np.abs(out[:, :, :96, :96] - g["corner"]).max(), np.abs(out[:, :, -64:, -64:] - g["tail"]).max())
    print("fuse mode v1 [%s]: final image err %.2e (tier %.1e)" % (sr.block0.precision, e, tol))
    assert out.shape == (1, 3, 512, 512) and e <= tol and abs(float(np.abs(out).mean()) - float(g["absmean"])) <= 1e-4
    sr._r3d_state.hparams["htbsr_head_weight_fuse_mode"] = "v3"
    with pytest.raises(NotImplementedError):
        sr(*args, noise_mode="none")


def test_resize_blend_kernels_vs_torch():
    """r3d_resize_bilinear vs F.interpolate(bilinear, align_corners=False, antialias) for the three shapes of sr_with_ref.py:77-82,110
    and odd sizes; r3d_blend / r3d_person_occlusion vs the torch expressions."""
    import torch
    import torch.nn.functional as F
    from real3dportrait_amd.sr_with_ref import blend, person_occlusion, resize_bilinear
    torch.manual_seed(3)
    for (C, H, W, OH, OW) in [(3, 128, 128, 256, 256), (3, 512, 512, 256, 256), (1, 64, 64, 256, 256), (2, 37, 53, 80, 21), (1, 300, 200, 128, 128)]:
        x = torch.randn(2, C, H, W, device="cuda")
        for aa in (True, False):
            ref = F.interpolate(x, size=(OH, OW), mode="bilinear", align_corners=False, antialias=aa)
            got = resize_bilinear(x, (OH, OW), aa)
            assert (got - ref).abs().max().item() <= 5e-6 * max(1.0, ref.abs().max().item()), (C, H, W, OH, OW, aa)
    a, b = torch.randn(2, 3, 40, 50, device="cuda"), torch.randn(2, 3, 40, 50, device="cuda")
    m = torch.rand(2, 1, 40, 50, device="cuda")
    assert (blend(a, b, m) - (a * m + b * (1 - m))).abs().max().item() <= 1e-6
    occ = torch.rand(2, 1, 40, 50, device="cuda")
    head = m.clone(); head[head > 0.9] = 1.0
    assert torch.equal(person_occlusion(m, occ, 0.9), (occ + head).clamp_(0, 1))


def test_torso_frames_on_three_streams_are_bit_identical():
    """bench.py's torso frame (to_plane_cnn -> planes -> rays -> fused SuperresolutionHybrid8XDC_Warp.forward) issued round-robin on three
    streams, each with its own module shells, must equal the single-stream frames bit for bit."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from real3dportrait_amd.frames import StreamPipeline, clone_generator_shell
    dev = torch.device("cuda", 0)
    G, clip, dec, scene = bench.build_scene(torch, dev, n_frames=8)
    frames = [bench.build_torso_frame(torch, dev, G)[0]] + [bench.build_torso_frame(torch, dev, clone_generator_shell(G))[0] for _ in range(2)]
    ref = [frames[0](t).clone() for t in range(6)]
    torch.cuda.synchronize()
    pipe = StreamPipeline(frames)
    for rep in range(3):
        out = [pipe.submit(t) for t in range(6)]
        pipe.sync(); torch.cuda.synchronize()
        for t in range(6):
            assert torch.equal(out[t], ref[t]), (rep, t, float((out[t] - ref[t]).abs().max()))
    assert float(ref[0].std()) > 1e-3


def test_torso_frame_fused_input_equals_unfused_sequence():
    """Camera-mode rays + the ray kernel's SPLIT copy as block0's operand (warp_split_input_spec) vs ray arrays + fp32 feature image +
    conversion launch: the same uint8-able frame, bit for bit."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    dev = torch.device("cuda", 0)
    G, clip, dec, scene = bench.build_scene(torch, dev, n_frames=8)
    fused = bench.build_torso_frame(torch, dev, G)[0]
    plain = bench.build_torso_frame(torch, dev, G, fused_input=False)[0]
    for t in range(4):
        a = fused(t).clone()
        b = plain(t)
        assert torch.equal(a, b), (t, float((a - b).abs().max()))
        a2 = fused(t)
        assert torch.equal(a2, b)


def test_warp_sr_forward_v2_batch_of_two_equals_two_singles():
    """N = 2 through the fused forward (the per-sample bounds of a tagged activation sit one scales record apart: consumers need them
    dense): a batch whose two samples differ in magnitude by 2^6 must equal the two samples run one by one."""
    import torch
    import warp_mock
    from real3dportrait_amd.sr_with_ref import SuperresolutionHybrid8XDC_Warp
    sr = SuperresolutionHybrid8XDC_Warp(32, 512, 0, True, torso_model=warp_mock.MockTorso(), hparams={"htbsr_head_threshold": 0.9}).cuda()
    warp_mock.load_warp_params(sr, lambda blk, p: load_block(torch, blk, p), to=lambda a: T(torch, a))
    i = {k: T(torch, v) for k, v in warp_mock.warp_inputs(N=2).items()}
    i["x"][1] *= 64.0                                                      # per-sample range folding must not couple the samples
    call = lambda sl: sr(i["x"][sl, :3].contiguous(), i["x"][sl].contiguous(), i["ws"][sl].contiguous(), i["ref_torso_rgb"][sl].contiguous(),
                         i["ref_bg_rgb"][sl].contiguous(), i["weights_img"][sl].contiguous(), None, None, None, noise_mode="none")[0]
    both = call(slice(0, 2)).cpu().numpy()
    assert both.shape == (2, 3, 512, 512) and np.isfinite(both).all()

    for n in range(2):
        sr.torso_model = _PerSample(warp_mock.MockTorso(), n)
        one = call(slice(n, n + 1)).cpu().numpy()
        tol = 2e-5 * max(1.0, float(np.abs(both[n]).max()))
        e = np.abs(one[0] - both[n]).max()
        print("sample %d: batch vs single max diff %.2e (tol %.1e)" % (n, e, tol))
        assert e <= tol


def _PerSample(inner, n):
    """Runs the stand-in torso network on a 2-sample batch built from the single sample and returns row n, so that a single-sample call
    sees exactly the tensors sample n of the batch saw (the stand-in's noise depends on the batch index)."""
    import torch

    class PerSample(torch.nn.Module):
        def forward(self, ref_torso_rgb_256, segmap, kp_s, kp_d, rgb_256, weights_256=None, cal_loss=True, target_torso_mask=None):
            rep = lambda t: t.expand(2, -1, -1, -1).contiguous()
            rgb_torso, ret = inner.forward(rep(ref_torso_rgb_256), segmap, kp_s, kp_d, rep(rgb_256), rep(weights_256), cal_loss, target_torso_mask)
            return rgb_torso[n:n + 1].contiguous(), {k: v[n:n + 1].contiguous() for k, v in ret.items()}
    return PerSample()


def test_conv_to_convstack_chain_batch_of_two():
    """Conv2d -> ConvStack with N = 2: the tagged bound of the first conv's output feeds the stack's fold (dense float[N] required)."""
    import torch
    from real3dportrait_amd.superresolution import Conv2d, ConvStack
    torch.manual_seed(3)
    c0 = Conv2d(16, 128, 3, 1, padding=1).cuda()
    st = ConvStack(Conv2d(128, 128, 3, 1, padding=1), torch.nn.LeakyReLU(0.2), Conv2d(128, 128, 3, 1, padding=1)).cuda()
    x = torch.randn(2, 16, 40, 40, device="cuda")
    x[1] *= 300.0
    y = st(c0(x, out_format="cb8"))
    ref = torch.nn.functional.conv2d(x.double(), c0.weight.double(), c0.bias.double(), padding=1)
    convs = [m for m in st if hasattr(m, "weight")]
    ref = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(ref, convs[0].weight.double(), convs[0].bias.double(), padding=1), 0.2)
    ref = torch.nn.functional.conv2d(ref, convs[1].weight.double(), convs[1].bias.double(), padding=1)
    for n in range(2):
        e = float((y[n].double() - ref[n]).abs().max()) / float(ref[n].abs().max())
        print("sample %d: rel err %.2e" % (n, e))
        assert e <= 2e-5
