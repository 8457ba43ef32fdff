"""CPU: the C oracle (oracle/r3d_oracle.c) must reproduce the golden vectors that were produced by the
reference's own PyTorch modules (tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest

from conftest import golden_planes, load_golden

RENDER_CASES = ["render_a_r16_16p16", "render_b_n2_r16_48p48", "render_c_invalid_r16_16p0",
                "render_e_white_r12_32p16_bw", "render_d_cfg1_r64_16p16", "render_f_trigrid_d3_r12_20p12"]

# fp32 tolerances (rgb in [-1,1]); SURVEY 8(d): rgb <= 2e-4, depth <= 1e-4
RGB_TOL, DEPTH_TOL = 2e-4, 1e-4


def dec_of(g):
    return g["dec_w1"], g["dec_b1"], g["dec_w2"], g["dec_b2"]


@pytest.mark.parametrize("name", RENDER_CASES)
def test_raygen_and_limits(oracle, name):
    g = load_golden(name)
    cams = g["cams"]
    o, d = oracle.raygen(cams[:, :16], cams[:, 16:], int(g["R"]))
    assert np.abs(o - g["origins"]).max() <= 1e-6
    assert np.abs(d - g["dirs"]).max() <= 2e-6
    rs, re, valid = oracle.ray_limits(g["origins"], g["dirs"], float(g["box_warp"]))
    raw_valid = g["raw_end"][..., 0] > g["raw_start"][..., 0]
    assert np.array_equal(valid, raw_valid)
    assert np.array_equal(valid, g["valid"][..., 0])
    assert np.abs(rs[valid] - g["raw_start"][..., 0][valid]).max() <= 2e-6
    assert np.abs(re[valid] - g["raw_end"][..., 0][valid]).max() <= 2e-6


@pytest.mark.parametrize("name", RENDER_CASES)
def test_render_matches_reference(oracle, name):
    g = load_golden(name)
    planes = golden_planes(g)
    rgb, depth, wsum, valid = oracle.render(planes, dec_of(g), g["origins"], g["dirs"], int(g["Nc"]), int(g["Nf"]),
                                            g["noise_c"], g["u_f"], float(g["box_warp"]), bool(g["white_back"]),
                                            triplane_depth=int(g["triplane_depth"]))
    assert np.array_equal(valid, g["valid"])
    assert np.abs(rgb - g["rgb"]).max() <= RGB_TOL
    assert np.abs(wsum - g["wsum"]).max() <= RGB_TOL
    assert np.abs(depth - g["depth"]).max() <= DEPTH_TOL


@pytest.mark.parametrize("name", ["run_model_a", "run_model_b_trigrid_d3"])
def test_run_model_matches_reference(oracle, name):
    g = load_golden(name)
    rgb, sigma = oracle.run_model(g["planes"], dec_of(g), g["coords"], float(g["box_warp"]), triplane_depth=int(g["triplane_depth"]))
    assert np.abs(rgb - g["rgb"]).max() <= 2e-5
    assert np.abs(sigma - g["sigma"]).max() <= 2e-4     # sigma is unbounded (|sigma| ~ 10)


def test_sr_blocks_match_reference(oracle):
    from real3dportrait_amd import synth
    g = load_golden("sr_small_a")
    params = synth.synth_sr_params(int(g["seed"]))
    x0, r0 = oracle.sr_block(g["x"][0], g["rgb"][0], params[0], g["ws"][0])
    x1, r1 = oracle.sr_block(x0, r0, params[1], g["ws"][0])
    scale = np.abs(g["rgb1"]).max()
    assert np.abs(x0[::4] - g["x0"][0]).max() <= 2e-5 * max(1.0, np.abs(g["x0"]).max())
    assert np.abs(r0 - g["rgb0"][0]).max() <= 2e-5 * max(1.0, np.abs(g["rgb0"]).max())
    assert np.abs(x1[::8] - g["x1"][0]).max() <= 2e-5 * max(1.0, np.abs(g["x1"]).max())
    assert np.abs(r1 - g["rgb1"][0]).max() <= 2e-5 * max(1.0, scale)


def _fusion_inputs(seed, R):
    from real3dportrait_amd import synth
    t = lambda shape, s, g=1.0: synth.hash_unitvar(seed, shape, stream=s) * np.float32(g)
    return dict(x_head=t((1, 256, R, R), 1), hid=t((1, 64, R, R), 2), bg=t((1, 3, R, R), 3, 0.5),
                rgb=t((1, 3, R, R), 4, 0.5), rgb_torso=t((1, 3, R, R), 5, 0.5),
                alpha=synth.synth_noise(seed, (1, 1, R, R), stream=6), occ=synth.synth_noise(seed, (1, 1, R, R), stream=8),
                ws=np.ones((1, 3, 512), np.float32) + synth.hash_unitvar(seed, (1, 3, 512), stream=9) * np.float32(0.1))


def test_fusion_stacks_match_reference(oracle):
    """Torso / background fusion convs + SynthesisBlockNoUp (sr_with_ref.py:24-63, :101-123) vs the reference."""
    from real3dportrait_amd import synth
    g = load_golden("fusion_a")
    seed, R = int(g["seed"]), int(g["R"])
    i = _fusion_inputs(seed, R)
    P = {k: synth.synth_conv_stack(seed, plan, 300 + 20 * n) for n, (k, plan) in enumerate(synth.FUSION_STACKS.items())}
    S = synth.FUSION_STACKS
    x_torso = oracle.conv_stack(i["hid"][0], S["torso_encoder"], P["torso_encoder"])
    x_bg = oracle.conv_stack(i["bg"][0], S["bg_encoder"], P["bg_encoder"])
    a, occ = i["alpha"][0], i["occ"][0]
    rgb1 = i["rgb"][0] * a + i["rgb_torso"][0] * (1 - a)
    x1 = oracle.conv_stack(np.concatenate([i["x_head"][0] * a, x_torso * (1 - a)]), S["fuse_head_torso_convs"], P["fuse_head_torso_convs"])
    x2, rgb2 = oracle.sr_block(x1, rgb1, synth.synth_sr_block(seed, 256, 256, 512, 400), i["ws"][0], up=False)
    x3 = oracle.conv_stack(np.concatenate([x2 * occ, x_bg * (1 - occ)]), S["fuse_fg_bg_convs"], P["fuse_fg_bg_convs"])
    for got, key in ((x_torso[::4], "x_torso"), (x_bg[::4], "x_bg"), (x1[::4], "x1"), (x2[::4], "x2"), (rgb2, "rgb2"), (x3[::4], "x3")):
        ref = g[key][0]
        assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), key


def test_to_plane_cnn_matches_reference(oracle):
    """Per-frame plane producer tail: to_plane_cnn + flips + cano add (segformer.py:691-700,721-729; secc_img2plane.py:76-77)."""
    from real3dportrait_amd import synth
    g = load_golden("toplane_a")
    seed, r = int(g["seed"]), int(g["r"])
    feat = synth.hash_unitvar(seed, (1, 256, r, r), stream=1)
    cano = synth.hash_unitvar(seed, (1, 3, 32, 2 * r, 2 * r), stream=2)
    secc = oracle.to_plane_cnn(feat[0], synth.TO_PLANE_CNN, synth.synth_conv_stack(seed, synth.TO_PLANE_CNN, 500),
                               synth.TO_PLANE_CNN_UP_BEFORE)
    got = cano[0] + secc
    assert np.abs(got - g["planes"][0]).max() <= 2e-5 * max(1.0, np.abs(g["planes"]).max())


def test_edge_cases_run(oracle):
    """No valid ray at all (camera looks away): fix-up is skipped, depths run backwards, outputs stay finite."""
    from real3dportrait_amd import synth
    planes = synth.synth_planes(3, N=1, H=16, W=16)
    cam = synth.look_at_camera(0.0, 0.0)
    cam[3] += 5.0
    o, d = oracle.raygen(cam[None, :16], cam[None, 16:], 8)
    rs, re, valid = oracle.ray_limits(o, d, 1.0)
    assert not valid.any() and (rs == -1).all() and (re == -2).all()
    noise = synth.synth_noise(1, (1, 64, 16, 1))
    u = synth.synth_noise(2, (64, 16))
    rgb, depth, wsum, v = oracle.render(planes, synth.synth_decoder(1), o, d, 16, 16, noise, u)
    assert np.isfinite(rgb).all() and np.isfinite(depth).all() and not v.any()


def test_sr_full_matches_reference(oracle):
    """SuperresolutionHybrid8XDC 128^2 -> 512^2, all-ones ws: the full-size SR golden (about 12 s of CPU)."""
    from real3dportrait_amd import synth
    g = load_golden("sr_full_a")
    seed = int(g["seed"])
    x = synth.hash_unitvar(seed, (1, 32, 128, 128), stream=1)[0]
    out = oracle.superresolution(np.ascontiguousarray(x[:3]), x, synth.synth_sr_params(seed), np.ones((14, 512), np.float32))
    tol = 2e-5 * max(1.0, np.abs(g["strided"]).max())
    assert np.abs(out[None][:, :, ::4, ::4] - g["strided"]).max() <= tol
    assert np.abs(out[None][:, :, :96, :96] - g["corner"]).max() <= tol
    assert np.abs(out[None][:, :, -64:, -64:] - g["tail"]).max() <= tol


def test_synthesis_matches_reference(oracle):
    """TriPlaneGenerator.synthesis at the reference default (R=128, 48+48, SR -> 512^2) restated end to end
    with the oracle pieces (raygen -> render -> image assembly -> SR -> clamp)."""
    from real3dportrait_amd import synth
    g = load_golden("synthesis_ref_a")
    seed, R, Nc, Nf = int(g["seed"]), int(g["R"]), int(g["Nc"]), int(g["Nf"])
    cam = g["cam"]
    o, d = oracle.raygen(cam[:, :16], cam[:, 16:], R)
    rgb, depth, wsum, valid = oracle.render(synth.synth_planes(seed, N=1), synth.synth_decoder(seed, sigma_bias=4.0), o, d, Nc, Nf,
                                            synth.synth_noise(seed, (1, R * R, Nc, 1), stream=7),
                                            synth.synth_noise(seed, (R * R, Nf), stream=8))
    feat = np.ascontiguousarray(rgb[0].T.reshape(32, R, R))          # triplane.py:120-122
    assert np.abs(np.clip(feat[:3], -1, 1) - g["image_raw"][0]).max() <= RGB_TOL
    assert np.abs(depth[0].reshape(R, R) - g["image_depth"][0, 0]).max() <= DEPTH_TOL
    img = oracle.superresolution(np.ascontiguousarray(feat[:3]), feat, synth.synth_sr_params(seed), np.ones((14, 512), np.float32))
    img = np.clip(img, -1, 1)[None]
    assert np.abs(img[:, :, ::4, ::4] - g["image_strided"]).max() <= 5e-4
    assert np.abs(img[:, :, :96, :96] - g["image_corner"]).max() <= 5e-4
