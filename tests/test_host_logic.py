"""CPU: host-side logic of the operators (no kernels launched)."""
import numpy as np
import pytest
import torch

EXPECTED_SR_KEYS = {
    "block0.resample_filter": (4, 4), "block0.conv0.weight": (256, 32, 3, 3), "block0.conv0.noise_strength": (),
    "block0.conv0.bias": (256,), "block0.conv0.resample_filter": (4, 4), "block0.conv0.noise_const": (256, 256),
    "block0.conv0.affine.weight": (32, 512), "block0.conv0.affine.bias": (32,),
    "block0.conv1.weight": (256, 256, 3, 3), "block0.conv1.noise_strength": (), "block0.conv1.bias": (256,),
    "block0.conv1.resample_filter": (4, 4), "block0.conv1.noise_const": (256, 256),
    "block0.conv1.affine.weight": (256, 512), "block0.conv1.affine.bias": (256,),
    "block0.torgb.weight": (3, 256, 1, 1), "block0.torgb.bias": (3,), "block0.torgb.affine.weight": (256, 512),
    "block0.torgb.affine.bias": (256,),
    "block1.resample_filter": (4, 4), "block1.conv0.weight": (128, 256, 3, 3), "block1.conv0.noise_strength": (),
    "block1.conv0.bias": (128,), "block1.conv0.resample_filter": (4, 4), "block1.conv0.noise_const": (512, 512),
    "block1.conv0.affine.weight": (256, 512), "block1.conv0.affine.bias": (256,),
    "block1.conv1.weight": (128, 128, 3, 3), "block1.conv1.noise_strength": (), "block1.conv1.bias": (128,),
    "block1.conv1.resample_filter": (4, 4), "block1.conv1.noise_const": (512, 512),
    "block1.conv1.affine.weight": (128, 512), "block1.conv1.affine.bias": (128,),
    "block1.torgb.weight": (3, 128, 1, 1), "block1.torgb.bias": (3,), "block1.torgb.affine.weight": (128, 512),
    "block1.torgb.affine.bias": (128,),
}


def test_sr_state_dict_keys_match_reference():
    """Key/shape list dumped from the reference's SuperresolutionHybrid8XDC: a reference
    checkpoint must load strict=True."""
    from real3dportrait_amd import SuperresolutionHybrid8XDC
    sr = SuperresolutionHybrid8XDC(channels=32, img_resolution=512, sr_num_fp16_res=0, sr_antialias=True,
                                   channel_base=32768, channel_max=512, fused_modconv_default="inference_only")
    got = {k: tuple(v.shape) for k, v in sr.state_dict().items()}
    assert got == EXPECTED_SR_KEYS
    f = sr.block0.resample_filter
    assert torch.allclose(f, torch.outer(torch.tensor([1., 3, 3, 1]), torch.tensor([1., 3, 3, 1])) / 64)


def test_decoder_and_generator_keys():
    from real3dportrait_amd import OSGDecoder, TriPlaneGenerator
    dec = OSGDecoder(32, {"decoder_lr_mul": 1, "decoder_output_dim": 32})
    assert {k: tuple(v.shape) for k, v in dec.state_dict().items()} == {
        "net.0.weight": (64, 32), "net.0.bias": (64,), "net.2.weight": (33, 64), "net.2.bias": (33,)}
    G = TriPlaneGenerator()
    keys = set(G.state_dict().keys())
    assert {"decoder.net.0.weight", "decoder.net.2.bias", "superresolution.block1.torgb.affine.weight"} <= keys
    assert not any(k.startswith(("renderer.", "ray_sampler.")) for k in keys)      # parameter-free like upstream
    assert G.rendering_kwargs["depth_resolution"] == 48 and G.rendering_kwargs["depth_resolution_importance"] == 48


def test_conv_stack_mirrors_torch_sequential():
    """ConvStack.from_torch keeps the reference's state_dict keys/shapes (the fusion stacks and to_plane_cnn are plain
    nn.Sequential upstream: sr_with_ref.py:24-63, segformer.py:691-700) and rejects what the HIP conv does not cover."""
    from real3dportrait_amd import synth
    from real3dportrait_amd.superresolution import Conv2d, ConvStack, SynthesisBlockNoUp
    nn = torch.nn
    seq = nn.Sequential(nn.Conv2d(256, 256, 3, 1, padding=1), nn.LeakyReLU(0.01, inplace=True), nn.UpsamplingBilinear2d(scale_factor=2.),
                        nn.Conv2d(256, 96, 3, 1, padding=1))
    st = ConvStack.from_torch(seq)
    assert list(st.state_dict().keys()) == list(seq.state_dict().keys())
    for k, v in seq.state_dict().items():
        assert torch.equal(st.state_dict()[k], v)
    assert isinstance(st[0], Conv2d) and st[1].negative_slope == 0.01
    with pytest.raises(NotImplementedError):
        ConvStack.from_torch(nn.Sequential(nn.Conv2d(8, 8, 3, 2, padding=1)))           # stride 2
    with pytest.raises(NotImplementedError):
        ConvStack.from_torch(nn.Sequential(nn.Conv2d(8, 8, 3, 1, padding=1), nn.ReLU()))
    with pytest.raises(NotImplementedError):
        Conv2d(32, 1, 3, 1, padding=1)                                                   # Cout % 4 (alpha predictor head)
    blk = SynthesisBlockNoUp(256, 256, w_dim=512, resolution=256, img_channels=3, is_last=False, conv_clamp=None)
    assert blk._UP == 0 and blk.conv0.up == 1 and blk.conv0.weight.shape == (256, 256, 3, 3)
    assert [c for c, *_ in synth.FUSION_STACKS["fuse_fg_bg_convs"]] == [512, 64, 256]


def test_noup_block_keys_match_reference():
    """The reference's SynthesisBlockNoUp (superresolution.py:159) and ours expose the same state_dict."""
    import os, sys
    ref = os.environ.get("R3D_REFERENCE", "/root/reference")
    if not os.path.isdir(ref):
        pytest.skip("reference checkout not present (GPU box)")
    sys.path.insert(0, ref)
    try:
        from modules.eg3ds.models.superresolution import SynthesisBlockNoUp as RefNoUp
    finally:
        sys.path.remove(ref)
    from real3dportrait_amd.superresolution import SynthesisBlockNoUp
    kw = dict(w_dim=512, resolution=256, img_channels=3, is_last=False, use_fp16=False, conv_clamp=None)
    a = RefNoUp(256, 256, channel_base=32768, channel_max=512, fused_modconv_default="inference_only", **kw).state_dict()
    b = SynthesisBlockNoUp(256, 256, **kw).state_dict()
    assert {k: tuple(v.shape) for k, v in a.items()} == {k: tuple(v.shape) for k, v in b.items()}


def test_patch_model_converts_fusion_stacks_of_a_warp_sr():
    """patch_model on a model whose superresolution has the SuperresolutionHybrid8XDC_Warp attributes (sr_with_ref.py:24-63):
    Sequential stacks -> ConvStack (keys preserved), unsupported stacks (the 1-channel alpha predictor) left untouched."""
    from real3dportrait_amd import patch_model
    from real3dportrait_amd.superresolution import ConvStack
    nn = torch.nn

    class WarpSR(nn.Module):
        def __init__(self):
            super().__init__()
            self.torso_encoder = nn.Sequential(nn.Conv2d(64, 256, 1, 1, padding=0))
            self.fuse_fg_bg_convs = nn.Sequential(nn.Conv2d(512, 64, 1, 1, padding=0), nn.LeakyReLU(), nn.Conv2d(64, 256, 3, 1, padding=1))
            self.head_torso_alpha_predictor = nn.Sequential(nn.Conv2d(7, 32, 3, 1, padding=1), nn.LeakyReLU(), nn.Conv2d(32, 1, 3, 1, padding=1), nn.Sigmoid())

    class Backbone(nn.Module):
        def __init__(self):
            super().__init__()
            self.to_plane_cnn = nn.Sequential(nn.Conv2d(256, 256, 3, 1, padding=1), nn.LeakyReLU(0.01, inplace=True),
                                              nn.UpsamplingBilinear2d(scale_factor=2.), nn.Conv2d(256, 96, 3, 1, padding=1))

    class Model(nn.Module):
        def __init__(self):
            super().__init__()
            self.renderer = nn.Identity(); self.ray_sampler = nn.Identity()
            self.superresolution = WarpSR(); self.secc_img2plane_backbone = Backbone()
            self.p = nn.Parameter(torch.zeros(1))
    m = Model()
    keys = {k: v.clone() for k, v in m.state_dict().items()}
    patch_model(m)
    assert isinstance(m.superresolution.torso_encoder, ConvStack) and isinstance(m.superresolution.fuse_fg_bg_convs, ConvStack)
    assert isinstance(m.secc_img2plane_backbone.to_plane_cnn, ConvStack)
    assert type(m.superresolution.head_torso_alpha_predictor).__name__ == "Sequential"       # Cout = 1: not covered, untouched
    after = m.state_dict()
    assert set(after) == set(keys) and all(torch.equal(after[k], keys[k]) for k in keys)


def test_operators_refuse_cpu_tensors_and_bad_options():
    from real3dportrait_amd import ImportanceRenderer, RaySampler
    with pytest.raises(AssertionError):
        RaySampler()(torch.eye(4)[None], torch.eye(3)[None], 8)          # device tensors only, no CPU fallback
    ren = ImportanceRenderer(hp={"triplane_feature_type": "3dgrid"})
    with pytest.raises(NotImplementedError):
        ren._check_options({"ray_start": "auto", "ray_end": "auto"})
    tri = ImportanceRenderer(hp={"triplane_feature_type": "trigrid_v2", "triplane_depth": 3})
    assert tri.triplane_depth == 3 and ImportanceRenderer(hp={"triplane_depth": 3}).triplane_depth == 1   # depth only for trigrids
    tri._check_options({"ray_start": "auto", "ray_end": "auto"})
    ren = ImportanceRenderer(hp={})
    with pytest.raises(NotImplementedError):
        ren._check_options({"ray_start": 2.25, "ray_end": 3.3})
    with pytest.raises(NotImplementedError):
        ren._check_options({"ray_start": "auto", "ray_end": "auto", "disparity_space_sampling": True})
    with pytest.raises(AssertionError):
        ren._check_options({"ray_start": "auto", "ray_end": "auto", "clamp_mode": "relu"})


def test_missing_library_fails_loudly(monkeypatch):
    from real3dportrait_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libr3d_hip.so")
    with pytest.raises(RuntimeError, match="no CPU/eager fallback"):
        _lib.load()


def test_synth_is_deterministic_and_sane():
    from real3dportrait_amd import synth
    a = synth.synth_planes(3, N=1, C=4, H=8, W=8)
    b = synth.synth_planes(3, N=1, C=4, H=8, W=8)
    assert np.array_equal(a, b) and a.dtype == np.float32
    big = synth.hash_unitvar(1, (200000,))
    assert abs(float(big.mean())) < 0.01 and abs(float(big.std()) - 1.0) < 0.01
    u = synth.synth_noise(5, (1000,))
    assert 0.0 <= float(u.min()) and float(u.max()) < 1.0
    cam = synth.look_at_camera(0.3, -0.1)
    c2w = cam[:16].reshape(4, 4)
    Rm = c2w[:3, :3]
    assert np.allclose(Rm.T @ Rm, np.eye(3), atol=1e-6) and abs(np.linalg.norm(c2w[:3, 3] - [0, 0, 0.2]) - 2.7) < 1e-5
    # the camera looks at the pivot: forward axis points from origin to lookat
    fwd = (np.array([0, 0, 0.2]) - c2w[:3, 3]); fwd /= np.linalg.norm(fwd)
    assert np.allclose(Rm[:, 2], fwd, atol=1e-6)


def test_frame_sharding_arithmetic():
    from real3dportrait_amd.frames import frame_seed, shard_frames
    for T, W in [(125, 8), (7, 8), (16, 4), (1, 2), (0, 3)]:
        chunks = [shard_frames(T, W, r) for r in range(W)]
        covered = [t for lo, hi in chunks for t in range(lo, hi)]
        assert covered == list(range(T))
        assert max(hi - lo for lo, hi in chunks) == (T + W - 1) // W if T else True
    assert shard_frames(125, 8, 0) == (0, 16) and shard_frames(125, 8, 7) == (112, 125)
    seeds = {frame_seed(7, t) for t in range(1000)}
    assert len(seeds) == 1000 and frame_seed(7, 3) == frame_seed(7, 3) != frame_seed(8, 3)


@pytest.mark.skipif(not __import__("os").path.isdir("/root/reference/modules/eg3ds"), reason="reference tree not present")
def test_patch_model_swaps_operators_of_the_reference_generator():
    """In the build container the reference's TriPlaneGenerator is importable: patch_model must replace its
    ray_sampler / renderer / superresolution with the HIP operators and copy the SR parameters strict=True."""
    import os, sys
    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference")
    try:
        from utils.commons.hparams import set_hparams, hparams
        from modules.eg3ds.models.triplane import TriPlaneGenerator as RefG
    finally:
        sys.path.remove("/root/reference")
    set_hparams("/root/reference/egs/egs_bases/eg3d/base.yaml", print_hparams=False)
    hparams.update(ray_near="auto", ray_far="auto", ones_ws_for_sr=True, enable_rescale_plane_regulation=False)
    G = RefG().eval()
    ref_sr_state = {k: v.clone() for k, v in G.superresolution.state_dict().items()}
    import real3dportrait_amd as r3d
    r3d.patch_model(G)
    assert type(G.renderer).__module__.startswith("real3dportrait_amd")
    assert type(G.ray_sampler).__module__.startswith("real3dportrait_amd")
    assert type(G.superresolution).__module__.startswith("real3dportrait_amd")
    new_state = G.superresolution.state_dict()
    assert set(new_state) == set(ref_sr_state)
    for k, v in ref_sr_state.items():
        assert torch.equal(new_state[k], v), k
    # the generator's own state_dict keys are unchanged (checkpoints keep loading strict=True)
    assert "decoder.net.0.weight" in G.state_dict() and "superresolution.block0.conv0.affine.weight" in G.state_dict()


@pytest.mark.skipif(not __import__("os").path.isdir("/root/reference/modules/real3d"), reason="reference tree not present")
def test_patch_model_on_the_real_warp_superresolution():
    """The reference's SuperresolutionHybrid8XDC_Warp (sr_with_ref.py:16-63; constructible here with sys.modules stand-ins for the
    uninstalled cv2 / torchvision / ...) inside a model shell: patch_model must convert block0 / block1 / head_torso_block /
    the four conv stacks, keep every parameter (strict state_dict equality), leave torso_model and the 1-channel alpha predictor
    alone, and install the fused forward for fuse mode v2."""
    import os, sys
    sys.dont_write_bytecode = True
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import ref_stubs
    ref_stubs.install()
    sys.path.insert(0, "/root/reference")
    try:
        from utils.commons.hparams import set_hparams, hparams
        set_hparams("/root/reference/egs/os_avatar/real3d_orig/secc_img2plane_torso_orig.yaml", print_hparams=False)
        from modules.real3d.super_resolution.sr_with_ref import SuperresolutionHybrid8XDC_Warp as RefWarp
    finally:
        sys.path.remove("/root/reference")
    import real3dportrait_amd as r3d
    from real3dportrait_amd.superresolution import ConvStack, SynthesisBlock, SynthesisBlockNoUp

    class Shell(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.renderer, self.ray_sampler = torch.nn.Identity(), torch.nn.Identity()
            self.superresolution = RefWarp(channels=32, img_resolution=512, sr_num_fp16_res=0, sr_antialias=True, channel_base=32768,
                                           channel_max=512, fused_modconv_default="inference_only")
    m = Shell().eval()
    before = {k: v.clone() for k, v in m.state_dict().items()}
    torso_model = m.superresolution.torso_model
    r3d.patch_model(m)
    sr = m.superresolution
    assert type(sr).__name__ == "SuperresolutionHybrid8XDC_Warp" and type(sr).__module__.startswith("modules.real3d")   # same object, same class
    assert type(sr.block0) is SynthesisBlock and type(sr.block1) is SynthesisBlock and type(sr.head_torso_block) is SynthesisBlockNoUp
    for name in ("torso_encoder", "bg_encoder", "fuse_head_torso_convs", "fuse_fg_bg_convs"):
        assert isinstance(getattr(sr, name), ConvStack), name
    assert type(sr.head_torso_alpha_predictor).__name__ == "Sequential" and sr.torso_model is torso_model
    after = m.state_dict()
    assert set(after) == set(before) and all(torch.equal(after[k], before[k]) for k in before)
    assert sr.forward.__func__.__name__ == "forward_v2" and sr._r3d_state.hparams["htbsr_head_threshold"] == hparams["htbsr_head_threshold"]
    # weight_fuse=False: the reference calls block1(x, None, ws) (sr_with_ref.py:161) -> block1 stays the reference's, no fused forward
    hparams["weight_fuse"] = False
    try:
        m2 = Shell().eval()
        r3d.patch_model(m2)
        assert type(m2.superresolution.block1).__module__.startswith("modules.eg3ds") and type(m2.superresolution.block0) is SynthesisBlock
        assert "forward" not in m2.superresolution.__dict__
    finally:
        hparams["weight_fuse"] = True
    # our mirror class exposes the same state_dict keys for everything it owns (torso_model is passed in)
    from real3dportrait_amd.sr_with_ref import SuperresolutionHybrid8XDC_Warp as OurWarp
    ours = {k: tuple(v.shape) for k, v in OurWarp(32, 512, 0, True).state_dict().items()}
    ref = {k: tuple(v.shape) for k, v in before.items() if k.startswith("superresolution.") and ".torso_model." not in k}
    assert ours == {k[len("superresolution."):]: v for k, v in ref.items()}


def test_full_size_goldens_are_present_and_consistent():
    """The full-size fixtures (config 5 SR, fusion stacks at 256^2, to_plane_cnn at 128^2 -> 256^2, masked synthesis, fused warp SR) are
    replayed on the GPU only (the C oracle would need minutes for them); here: they load and have the shapes the GPU tests expect."""
    from conftest import load_golden
    g = load_golden("sr_cfg5_a")
    assert g["strided"].shape == (1, 3, 128, 128) and g["mid"].shape == (1, 3, 64, 64)
    g = load_golden("fusion_full_a")
    assert int(g["R"]) == 256 and g["x3"].shape == (1, 16, 32, 32) and g["x1_crop"].shape == (1, 8, 32, 24)
    g = load_golden("toplane_full_a")
    assert int(g["r"]) == 128 and g["strided"].shape == (1, 3, 32, 32, 32)
    g = load_golden("synthesis_mask_a")
    assert 0.2 < float(g["masked_frac"]) < 0.6 and g["image_raw"].shape == (1, 3, 128, 128)
    g = load_golden("warp_sr_a")
    assert g["strided"].shape == (1, 3, 128, 128) and float(g["threshold"]) == 0.9
    g = load_golden("warp_sr_v1_a")                 # fuse mode v1 of the same forward (round 5)
    assert g["strided"].shape == (1, 3, 128, 128) and float(g["threshold"]) == 0.9
    g = load_golden("warp_sr_two_stage_a")          # the reference's two-stage entry of the same module (round 5)
    assert g["strided"].shape == (1, 3, 128, 128) and g["x0_strided"].shape == (1, 32, 32, 32)
    assert {"deformed_torso_hid", "occlusion_2", "ref_bg_rgb_256", "weights_256", "x", "ws", "rgb"} <= set(str(k) for k in g["keys"])


def test_write_frames_round_trips(tmp_path):
    """frames.write_frames: raw rgb24 / npy / ppm / png dumps of a uint8 clip decode back to the same bytes."""
    import struct, zlib
    from real3dportrait_amd.frames import write_frames
    rng = np.random.RandomState(3)
    clip = rng.randint(0, 256, size=(3, 20, 28, 3)).astype(np.uint8)
    (raw,) = write_frames(torch.from_numpy(clip), str(tmp_path / "clip.raw"), "raw")
    assert np.array_equal(np.fromfile(raw, np.uint8).reshape(clip.shape), clip)
    (npy,) = write_frames(clip, str(tmp_path / "clip"), "npy")
    assert np.array_equal(np.load(npy), clip)
    ppm = write_frames(clip, str(tmp_path / "ppm"), "ppm")
    data = open(ppm[1], "rb").read()
    assert data.startswith(b"P6\n28 20\n255\n") and np.array_equal(np.frombuffer(data[len(b"P6\n28 20\n255\n"):], np.uint8).reshape(20, 28, 3), clip[1])
    png = write_frames(clip, str(tmp_path / "png"), "png")
    data = open(png[2], "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, ihdr = 8, b"", None
    while pos < len(data):
        (n,), tag = struct.unpack(">I", data[pos:pos + 4]), data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + body) & 0xFFFFFFFF
        if tag == b"IHDR": ihdr = struct.unpack(">IIBBBBB", body)
        if tag == b"IDAT": idat += body
        pos += 12 + n
    assert ihdr == (28, 20, 8, 2, 0, 0, 0)
    rows = zlib.decompress(idat)
    img = np.frombuffer(rows, np.uint8).reshape(20, 1 + 28 * 3)
    assert (img[:, 0] == 0).all() and np.array_equal(img[:, 1:].reshape(20, 28, 3), clip[2])
    with pytest.raises(ValueError):
        write_frames(clip, str(tmp_path / "x"), "gif")


def test_committed_bench_line_keeps_the_contract():
    """profiles/r0{2,4}/bench_n1.json are bench.py lines from the MI355X: the keys the driver and the judge read must be there."""
    import json
    import os
    for rnd in ("r02", "r04"):
        _check_bench_line(json.loads(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", rnd, "bench_n1.json")).read().strip().splitlines()[-1]))


def _check_bench_line(d):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port")
    assert abs(d["value"] - d["steps"] * d["n_gpus"] / (d["ms_per_step"] * d["steps"] * 1e-3)) / d["value"] < 1e-3
