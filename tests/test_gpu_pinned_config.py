"""GPU parity (`-m gpu`) of the BENCHMARKED configuration itself, and of operands the synthetic data never has (round 4).

(a) bench.py renders with noise_mode='hash' (frames.ClipRenderer): the ray kernel derives the two sampling draws of the reference
    (torch.rand_like at modules/eg3ds/volumetric_rendering/renderer.py:226, torch.rand at :281) from (seed, ray, sample) in-kernel.  Every other
    parity test injects noise tensors, so until round 4 the timed path had determinism tests only.  Here the host mirror of the hash
    (synth.render_hash_noise, held against the device source by tests/test_hash_noise.py) feeds the SAME jitter to the oracle:
      * hash mode == the mirror's arrays injected, bit for bit (camera mode and explicit rays);
      * ClipRenderer's frame t (cano + residual planes, camera mode, SPLIT hand-off, SR, uint8) vs the oracle's render + SR of the same frame,
        both SR precisions;
      * a clip's frames do not depend on which shard / stream / order renders them.
(b) heavy tails: one spike of 2^{6,10,14} sigma per sample in the SR input, in a conv-stack input, in the weight rows and in the planes,
    for BOTH shipped SR precisions, vs float64.  Two figures per case: err / max|ref| (the tolerance SURVEY 8d states) and the stricter
    `far` figure = the error over the outputs the spike cannot reach / max|ref| over the same outputs (a spike inflates max|ref| by
    itself; this one does not let it hide anything).
(c) BASELINE config 5 at FULL size (N = 8, R = 256, 96 + 96 samples: 100 M samples, 524 288 rays) on a strided ray subset vs the oracle.

Tolerances: rgb / wsum <= 2e-4, depth <= 1e-4; SR <= 2e-4 * max(1, max|ref|); final uint8 frame <= 1 count."""
import numpy as np
import pytest

from test_gpu_parity import DEPTH_TOL, RGB_TOL, SR_TOL, T, load_block, make_decoder, opts
from test_gpu_range_and_sizes import _block_fp64, _run_model_fp64

pytestmark = pytest.mark.gpu
PRECISIONS = ["f16x3", "f16mx"]


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from real3dportrait_amd import _lib
    _lib.load()
    return torch


# ------------------------------------------------------------------------------------------------
# (a) the in-kernel hash noise
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("camera_mode", [True, False])
def test_hash_mode_equals_injected_mirror_arrays(torch_cuda, camera_mode):
    """noise_mode='hash' and noise_override = (host mirror arrays) are the same render, bit for bit: the mirror IS the device's noise."""
    torch = torch_cuda
    from real3dportrait_amd import ImportanceRenderer, RaySampler, synth
    from real3dportrait_amd.frames import frame_seed
    N, R, Nc, Nf = 2, 48, 48, 48
    planes = T(torch, synth.synth_planes(201, N=N, H=128, W=128))
    dec = make_decoder(torch, synth.synth_decoder(202, sigma_bias=3.0))
    cams = T(torch, synth.camera_sweep(N, -0.2, 0.3))
    c2w, K = cams[:, :16].view(-1, 4, 4), cams[:, 16:].view(-1, 3, 3)
    seed = frame_seed(7, 5)
    nc, uf = synth.render_hash_noise(seed, np.arange(N * R * R), Nc, Nf)
    outs = []
    for mode in ("hash", "inject"):
        ren = ImportanceRenderer(hp={})
        if mode == "hash":
            ren.noise_mode, ren.seed = "hash", seed
        else:
            ren.noise_override = (T(torch, nc.reshape(N, R * R, Nc, 1)), T(torch, uf))
        if camera_mode:
            outs.append(ren.forward_camera(planes, dec, c2w, K, R, opts(Nc, Nf)))
        else:
            o, d = RaySampler()(c2w, K, R)
            outs.append(ren(planes, dec, o, d, opts(Nc, Nf)))
    torch.cuda.synchronize()
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert float(outs[0][0].std()) > 0.05


def _clip_scene(torch, seed=31, frames=6):
    from real3dportrait_amd import TriPlaneGenerator, synth
    G = TriPlaneGenerator().cuda().eval()
    dec = synth.synth_decoder(seed, sigma_bias=4.0)
    with torch.no_grad():
        G.decoder.net[0].weight.copy_(T(torch, dec[0])); G.decoder.net[0].bias.copy_(T(torch, dec[1]))
        G.decoder.net[2].weight.copy_(T(torch, dec[2])); G.decoder.net[2].bias.copy_(T(torch, dec[3]))
    params = synth.synth_sr_params(seed)
    load_block(torch, G.superresolution.block0, params[0]); load_block(torch, G.superresolution.block1, params[1])
    cano = synth.synth_planes(seed, N=1)
    res = [synth.synth_planes(seed + 1 + i, N=1, scale=0.1) for i in range(2)]
    cams = synth.camera_sweep(frames, -0.3, 0.3)
    return G, dec, params, cano, res, cams


@pytest.mark.parametrize("precision", PRECISIONS)
def test_benchmarked_frame_vs_oracle(torch_cuda, oracle, precision):
    """Frame t of frames.ClipRenderer -- the object bench.py times: planes = cano + residual_t through the layout kernel, rays generated in
    the kernels, in-kernel hash noise (seed = frame_seed(base, t)), 128^2 x (48 + 48), the ray kernel's SPLIT hand-off, SR -> 512^2, uint8
    ring -- against the oracle rendering the same frame from the same planes, camera and (mirrored) jitter."""
    torch = torch_cuda
    from real3dportrait_amd import synth
    from real3dportrait_amd.frames import ClipRenderer, frame_seed
    G, dec, params, cano, res, cams = _clip_scene(torch)
    for b in (G.superresolution.block0, G.superresolution.block1):
        b.precision = precision
    ws = torch.ones(1, 14, 512, device="cuda")
    clip = ClipRenderer(G, T(torch, cano), [T(torch, r) for r in res], T(torch, cams), ws, base_seed=5)
    t, R, Nc, Nf = 3, 128, 48, 48
    feat = clip._features(t)                                           # [1,32,R,R]
    img = clip.render_image(t).cpu().numpy()
    u8 = clip.render_u8(t).cpu().numpy()
    torch.cuda.synchronize()
    planes_t = (cano + res[t % 2]).astype(np.float32)
    o, d = oracle.raygen(cams[t:t + 1, :16], cams[t:t + 1, 16:], R)
    nc, uf = synth.render_hash_noise(frame_seed(5, t), np.arange(R * R), Nc, Nf)
    ref = oracle.render(planes_t, dec, o, d, Nc, Nf, nc.reshape(1, R * R, Nc, 1), uf)
    rfeat = ref[0][0].T.reshape(32, R, R)
    e_feat = np.abs(feat[0].cpu().numpy() - rfeat).max()
    rimg = oracle.superresolution(rfeat[:3].copy(), rfeat, params, np.ones((14, 512), np.float32))
    e_img = np.abs(img[0] - rimg).max() / max(1.0, np.abs(rimg).max())
    ru8 = ((np.clip(rimg, -1, 1).transpose(1, 2, 0) + 1) / 2 * 255).astype(np.int32).astype(np.uint8)
    d8 = np.abs(u8.astype(np.int32) - ru8.astype(np.int32))
    print("benchmarked frame [%s]: feature err %.2e, SR image err %.2e of max|ref| %.2f, uint8 frame: %.4f %% of the bytes differ (max %d)"
          % (precision, e_feat, e_img, np.abs(rimg).max(), 100.0 * (d8 > 0).mean(), d8.max()))
    assert e_feat <= RGB_TOL
    assert e_img <= SR_TOL
    assert d8.max() <= 1 and (d8 > 0).mean() < 0.02
    assert np.abs(rfeat).std() > 0.1 and u8.std() > 5


def test_clip_frames_do_not_depend_on_shard_stream_or_order(torch_cuda):
    """Frame t's bytes are a function of t only: one renderer in order == two 'ranks' rendering their shard_frames() chunks with their own
    module shells == reverse order == three HIP streams.  (What lets N GPUs render one clip with no collective but the final gather.)"""
    torch = torch_cuda
    from real3dportrait_amd.frames import ClipRenderer, PipelinedClipRenderer, clone_generator_shell, shard_frames
    Tn = 7
    G, dec, params, cano, res, cams = _clip_scene(torch, seed=37, frames=Tn)
    cano_t, res_t, cams_t, ws = T(torch, cano), [T(torch, r) for r in res], T(torch, cams), torch.ones(1, 14, 512, device="cuda")
    serial = ClipRenderer(G, cano_t, res_t, cams_t, ws, base_seed=9)
    ref = torch.stack([serial.render_u8(t).clone() for t in range(Tn)])
    world = 2
    clip = torch.zeros_like(ref)
    for rank in range(world):
        lo, hi = shard_frames(Tn, world, rank)
        shard = ClipRenderer(clone_generator_shell(G), cano_t, res_t, cams_t, ws, base_seed=9)
        for t in reversed(range(lo, hi)):                                # a rank is free to render its chunk in any order
            clip[t] = shard.render_u8(t)
    pipe = PipelinedClipRenderer(G, cano_t, res_t, cams_t, ws, base_seed=9, n_streams=3)
    ring = torch.zeros_like(ref)
    for t in (4, 0, 6, 2, 5, 1, 3):
        pipe.render_u8(t, out=ring[t:t + 1])
    pipe.sync(); torch.cuda.synchronize()
    assert torch.equal(clip, ref) and torch.equal(ring, ref)
    assert not torch.equal(ref[0], ref[1])


# ------------------------------------------------------------------------------------------------
# (c) BASELINE config 5 at full size
# ------------------------------------------------------------------------------------------------
def test_render_cfg5_full_size_subset_vs_oracle(torch_cuda, oracle):
    """N = 8 cameras of one tri-plane, R = 256, 96 + 96 samples per ray (BASELINE config 5: 524 288 rays, 100 M samples per launch, the
    <6,6> instantiation on its real grid), in-kernel hash noise.  The oracle renders every 8th row and column of every image (8 192 rays,
    1.6 M samples) from the mirrored jitter.  All rays hit the box here, so the only global coupling of the call is the depth clamp
    [min, max of ALL sample depths] (ray_marcher.py:46-50), which cannot bind on a ray whose weights are finite."""
    torch = torch_cuda
    from real3dportrait_amd import ImportanceRenderer, synth
    N, R, Nc, Nf, step = 8, 256, 96, 96, 8
    planes1 = synth.synth_planes(611, N=1)
    dec_np = synth.synth_decoder(612, sigma_bias=4.0)
    cams = synth.camera_sweep(N, -0.4, 0.4)
    seed = 0xC0FFEE123456789
    ren = ImportanceRenderer(hp={})
    ren.noise_mode, ren.seed = "hash", seed
    planes = T(torch, planes1).expand(N, -1, -1, -1, -1).contiguous()
    camt = T(torch, cams)
    rgb, depth, wsum, valid = ren.forward_camera(planes, make_decoder(torch, dec_np), camt[:, :16].view(-1, 4, 4), camt[:, 16:].view(-1, 3, 3),
                                                 R, opts(Nc, Nf))
    torch.cuda.synchronize()
    assert bool(valid.all()) and torch.isfinite(rgb).all() and torch.isfinite(depth).all()
    M = R * R
    rows = np.arange(step // 2, R, step)
    pix = (rows[:, None] * R + rows[None, :]).reshape(-1)                       # 1 024 rays per image
    o, d = oracle.raygen(cams[:, :16], cams[:, 16:], R)
    o_s, d_s = o[:, pix].reshape(1, -1, 3), d[:, pix].reshape(1, -1, 3)         # the 8 cameras share the tri-plane: one oracle call, N = 1
    gray = (np.arange(N)[:, None] * M + pix[None, :]).reshape(-1)
    nc, uf = synth.render_hash_noise(seed, gray, Nc, Nf)
    ref = oracle.render(planes1, dec_np, o_s, d_s, Nc, Nf, nc.reshape(1, -1, Nc, 1), uf)
    g_rgb = rgb.cpu().numpy()[:, pix].reshape(1, -1, 32)
    g_dep = depth.cpu().numpy()[:, pix].reshape(1, -1, 1)
    g_ws = wsum.cpu().numpy()[:, pix].reshape(1, -1, 1)
    e = (np.abs(g_rgb - ref[0]).max(), np.abs(g_dep - ref[1]).max(), np.abs(g_ws - ref[2]).max())
    print("cfg5 full size (%d rays checked of %d): rgb %.2e depth %.2e wsum %.2e" % (gray.size, N * M, e[0], e[1], e[2]))
    assert e[0] <= RGB_TOL and e[2] <= RGB_TOL and e[1] <= DEPTH_TOL
    # the last image's last rays sit at the top of the 32-bit index ranges the kernel uses (ray * Nc, colour offsets n * 32 * M + ...)
    assert gray.max() >= N * M - step * R


# ------------------------------------------------------------------------------------------------
# (b) heavy tails
# ------------------------------------------------------------------------------------------------
SPIKES = [6, 10, 14]
# Asserted tiers.  `max` figure (err / max|ref|, the tolerance SURVEY 8d states): f16x3 2e-5, f16mx 2e-4 for every spike.  `far` figure
# (outputs no spike reaches): f16x3 2e-5 (measured <= 2.3e-6: the exact power-of-two fold keeps 22+ bits at max / rms = 2^14); f16mx 2e-4 for
# EVERY spike since round 5: its activation records are OCP e5m2 (a per-element exponent, 29 binades of normal range under the fold's bound),
# so the typical outputs no longer depend on how far the tensor's bound sits above them.  Until round 4 the records were e4m3 with one exponent
# per tensor and went subnormal from max / rms = 2^10 on (far 1.6e-4 at 2^10, 3.6e-4 at 2^14: asserted 1e-3 then, and the reason 'f16mx' was
# not the library default); profiles/r05/mx_format_model.txt is the numpy model of both formats, `make EXTRA=-DR3D_MX_ACT_E4M3=1` the A/B build.
# Round 6: the f16mx tiers are asserted at 1e-4 near / 5e-5 far instead of at the stated tolerance itself (measured 6.3e-5 / 2.5e-5
# in round 5: a 3 x regression would have passed unnoticed).  The far figure only means something where a spike has a far field (`where == "input"`).
_TIER = {"f16x3": 2e-5, "f16mx": 1e-4}
_TIER_FAR = {("f16x3", 6): 2e-5, ("f16x3", 10): 2e-5, ("f16x3", 14): 2e-5, ("f16mx", 6): 5e-5, ("f16mx", 10): 5e-5, ("f16mx", 14): 5e-5}


def _far_mask(shape_hw, centers, radius):
    H, W = shape_hw
    yy, xx = np.mgrid[0:H, 0:W]
    m = np.ones((H, W), bool)
    for (cy, cx) in centers:
        m &= (np.abs(yy - cy) > radius) | (np.abs(xx - cx) > radius)
    return m


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("k", SPIKES)
@pytest.mark.parametrize("where", ["input", "input_channel", "weight_rows"])
def test_sr_block_heavy_tail(torch_cuda, precision, k, where):
    """SynthesisBlock (up-sampling conv + FIR, 3x3 conv, toRGB) vs float64 with an outlier of 2^k sigma:
      input         one element per sample (a spatial spike: max/rms of the block input = 2^k);
      input_channel one whole input channel times 2^k (a hot channel);
      weight_rows   one element of every weight row of conv0 and conv1 times 2^k (demodulation then shrinks the rest of the row by 2^-k)."""
    torch = torch_cuda
    from real3dportrait_amd import synth
    from real3dportrait_amd.superresolution import SynthesisBlock
    N, Cin, Cout, H, W = 2, 32, 128, 24, 20
    p = {kk: tuple(np.array(a) for a in v) for kk, v in synth.synth_sr_block(91, Cin, Cout, 512, 700).items()}
    x = synth.hash_unitvar(92, (N, Cin, H, W), stream=1)
    img = synth.hash_unitvar(92, (N, 3, H, W), stream=2) * np.float32(0.5)
    ws = np.ones((N, 3, 512), np.float32) + synth.hash_unitvar(92, (N, 3, 512), stream=3) * np.float32(0.2)
    s = np.float32(2.0 ** k)
    centers = [(7, 5), (15, 12)]
    if where == "input":
        for n, (cy, cx) in enumerate(centers):
            x[n, 3 + 9 * n, cy, cx] = s * (1 if n == 0 else -1)
    elif where == "input_channel":
        x[:, 11] *= s
    else:
        for layer in ("conv0", "conv1"):
            w_ = p[layer][0].copy()
            co = np.arange(w_.shape[0])
            w_[co, (co * 7) % w_.shape[1], co % 3, (co // 3) % 3] *= s
            p[layer] = (w_,) + p[layer][1:]
    blk = SynthesisBlock(Cin, Cout, w_dim=512, resolution=2 * H, img_channels=3, is_last=False, conv_clamp=None).cuda()
    load_block(torch, blk, p)
    blk.precision = precision
    xo, io = blk(T(torch, x), T(torch, img), T(torch, ws), noise_mode="none")
    rx, ri = _block_fp64(torch, p, torch.from_numpy(x), torch.from_numpy(img), torch.from_numpy(ws), True, None)
    assert torch.isfinite(xo).all() and torch.isfinite(io).all()
    ex_abs, ei_abs = (xo.cpu().double() - rx).abs().numpy(), (io.cpu().double() - ri).abs().numpy()
    rx, ri = rx.abs().numpy(), ri.abs().numpy()
    ex, ei = ex_abs.max() / rx.max(), ei_abs.max() / ri.max()
    if where == "input":            # outputs no spike can reach: 2x up-sampling + FIR + two 3x3 convs = 6 output pixels; per sample
        fx = fi = 0.0
        for n, c in enumerate(centers):
            far = _far_mask((2 * H, 2 * W), [(2 * c[0], 2 * c[1])], 8)
            fx = max(fx, ex_abs[n][:, far].max() / rx[n][:, far].max())
            fi = max(fi, ei_abs[n][:, far].max() / ri[n][:, far].max())
    else:
        fx, fi = ex, ei
    print("SR block heavy tail [%s] %s 2^%d: x %.2e (far %.2e), img %.2e (far %.2e) of max|ref|" % (precision, where, k, ex, fx, ei, fi))
    far_tier = _TIER_FAR[(precision, k)] if where == "input" else _TIER[precision]
    assert max(ex, ei) <= _TIER[precision] and max(fx, fi) <= far_tier, (precision, where, k, ex, fx, ei, fi)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_sr_block_dense_heavy_tail(torch_cuda, precision):
    """The case the e5m2 records are weakest on: a DENSE heavy tail -- every element log-normal with sigma = 4 (the 16-channel group
    of every pixel spans > 2^10 in magnitude, the tensor > 2^30), random signs -- through a SynthesisBlock, vs float64.  No far field here: every output
    sees large and small operands alike."""
    torch = torch_cuda
    from real3dportrait_amd import synth
    from real3dportrait_amd.superresolution import SynthesisBlock
    N, Cin, Cout, H, W = 1, 32, 128, 24, 20
    p = {kk: tuple(np.array(a) for a in v) for kk, v in synth.synth_sr_block(95, Cin, Cout, 512, 710).items()}
    g = synth.hash_unitvar(96, (N, Cin, H, W), stream=1).astype(np.float64)
    sg = np.sign(synth.hash_unitvar(96, (N, Cin, H, W), stream=2)).astype(np.float64)
    x = (sg * np.exp(4.0 * g)).astype(np.float32)
    span = np.log2(np.abs(x).reshape(N, Cin // 16, 16, H * W).max(2) / np.abs(x).reshape(N, Cin // 16, 16, H * W).min(2))
    assert np.median(span) >= 10.0, np.median(span)
    img = synth.hash_unitvar(96, (N, 3, H, W), stream=3) * np.float32(0.5)
    ws = np.ones((N, 3, 512), np.float32)
    blk = SynthesisBlock(Cin, Cout, w_dim=512, resolution=2 * H, img_channels=3, is_last=False, conv_clamp=None).cuda()
    load_block(torch, blk, p)
    blk.precision = precision
    xo, io = blk(T(torch, x), T(torch, img), T(torch, ws), noise_mode="none")
    rx, ri = _block_fp64(torch, p, torch.from_numpy(x), torch.from_numpy(img), torch.from_numpy(ws), True, None)
    assert torch.isfinite(xo).all() and torch.isfinite(io).all()
    ex = float((xo.cpu().double() - rx).abs().max() / rx.abs().max()); ei = float((io.cpu().double() - ri).abs().max() / ri.abs().max())
    # the typical output (median |ref|) against the typical error: the figure a uniform bound / max comparison hides
    med = float((xo.cpu().double() - rx).abs().median() / rx.abs().median())
    print("SR block dense heavy tail [%s] (median group span 2^%.1f): x %.2e img %.2e of max|ref|; median error / median |ref| %.2e" % (precision, np.median(span), ex, ei, med))
    assert max(ex, ei) <= _TIER[precision], (precision, ex, ei)


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("k", SPIKES)
@pytest.mark.parametrize("where", ["input", "weight_rows"])
def test_conv_stack_heavy_tail(torch_cuda, k, where, precision):
    """Three chained plain convs (bg_encoder's plan, sr_with_ref.py:27-33; SPLIT hand-offs; under 'f16mx' the two 3x3 convs with a SPLIT_MX
    input run their cross products on the fp8 MFMA, the first one -- fp32 input, Cin = 3 -- stays f16x3) with a 2^k sigma spike in the input /
    in every weight row of every layer, vs float64.  A weight spike of a plain conv (no demodulation) makes a hot output pixel pattern in
    that channel: the next layer sees heavy-tailed activations.  In f16mx the second and third conv read operands one and two propagated
    bounds away from the measured input: the case DESIGN 4.2c states the tier for."""
    torch = torch_cuda
    from real3dportrait_amd import synth
    from real3dportrait_amd.superresolution import Conv2d, ConvStack
    plan = [(3, 64, 3, True), (64, 256, 3, True), (256, 256, 3, False)]
    s = np.float32(2.0 ** k)
    mods = []
    for (ci, co, ks, lrelu), (w, b) in zip(plan, synth.synth_conv_stack(93, plan, 300)):
        if where == "weight_rows":
            w = w.copy()
            co_i = np.arange(co)
            w[co_i, (co_i * 5) % ci, co_i % 3, (co_i // 3) % 3] *= s
        c = Conv2d(ci, co, ks, 1, padding=ks // 2)
        c.precision = precision
        with torch.no_grad():
            c.weight.copy_(torch.from_numpy(w)); c.bias.copy_(torch.from_numpy(b))
        mods.append(c)
        if lrelu:
            mods.append(torch.nn.LeakyReLU())
    st = ConvStack(*mods).cuda()
    x = synth.hash_unitvar(94, (1, 3, 40, 36), stream=1)
    if where == "input":
        x[0, 1, 17, 20] = s
    y = st(T(torch, x))
    r = torch.from_numpy(x).double()
    for m in st:
        r = torch.nn.functional.conv2d(r, m.weight.detach().double().cpu(), m.bias.detach().double().cpu(), padding=m.padding[0]) \
            if isinstance(m, Conv2d) else torch.nn.functional.leaky_relu(r, m.negative_slope)
    err = (y.cpu().double() - r).abs().numpy()[0]
    ref = r.abs().numpy()[0]
    e = err.max() / ref.max()
    far = _far_mask((40, 36), [(17, 20)], 4) if where == "input" else np.ones((40, 36), bool)
    f = err[:, far].max() / ref[:, far].max()
    print("conv stack heavy tail [%s] %s 2^%d: %.2e (far %.2e) of max|ref|" % (precision, where, k, e, f))
    assert torch.isfinite(y).all()
    if precision == "f16x3":
        assert e <= 2e-5 and f <= 1e-4, (where, k, e, f)
    else:
        # e5m2 activation records (round 5): the far-field no longer depends on the spike (round 4, e4m3 with one exponent per tensor: 2.3e-4 at a
        # spike of 2^10 sigma, 3.6e-4 at 2^14 -- the third conv's operand is two propagated bounds, ~10 binades, from the measured input on top of it)
        assert e <= 1e-4 and f <= (5e-5 if where == "input" else 1e-4), (where, k, e, f)


@pytest.mark.parametrize("k", SPIKES)
def test_decoder_heavy_tailed_planes(torch_cuda, oracle, k):
    """Planes with max >> rms: one texel of each plane at 2^k sigma (what a checkpoint's secc residual can do), so the range fold's bound is
    2^k above the typical feature.  Point queries vs float64 (near the spike the gathered feature itself is huge: the allowance follows
    |x| |w| like an fp32 evaluation's error does), and a small render vs the fp32 oracle."""
    torch = torch_cuda
    from real3dportrait_amd import ImportanceRenderer, OSGDecoder, synth
    planes = synth.synth_planes(95, N=2, H=32, W=32)
    s = np.float32(2.0 ** k)
    for p in range(3):
        planes[0, p, 5 + 3 * p, 9 + p, 20 - 2 * p] = s
        planes[1, p, 30 - p, 15, 3 + p] = -s
    dec = list(synth.synth_decoder(96, sigma_bias=2.0))
    coords = ((synth.synth_noise(97, (2, 900, 3)) - 0.5) * 1.2).astype(np.float32)
    decm = OSGDecoder().cuda()
    with torch.no_grad():
        decm.net[0].weight.copy_(torch.from_numpy(dec[0])); decm.net[0].bias.copy_(torch.from_numpy(dec[1]))
        decm.net[2].weight.copy_(torch.from_numpy(dec[2])); decm.net[2].bias.copy_(torch.from_numpy(dec[3]))
    out = ImportanceRenderer(hp={}).run_model(torch.from_numpy(planes).cuda(), decm, torch.from_numpy(coords).cuda(), None, {"box_warp": 1.0})
    rgb, sig = out["rgb"].cpu().numpy().astype(np.float64), out["sigma"].cpu().numpy().astype(np.float64)
    assert np.isfinite(rgb).all() and np.isfinite(sig).all()
    r_rgb, r_sig, t_rgb, t_sig = _run_model_fp64(planes, dec, coords)
    e_rgb, e_sig = np.abs(rgb - r_rgb), np.abs(sig - r_sig)
    # the typical point (no spike texel among its taps): plain fp32-class figures
    typical = np.abs(r_sig[..., 0]) < 50.0
    print("decoder, planes spike 2^%d: rgb err %.2e (typical points %.2e), sigma err %.2e rel %.2e (typical %.2e); worst / fp32 allowance: rgb %.1f sigma %.1f"
          % (k, e_rgb.max(), e_rgb[typical].max(), e_sig.max(), (e_sig / (1 + np.abs(r_sig))).max(), e_sig[typical].max(),
             (e_rgb / (2e-6 + t_rgb)).max(), (e_sig / (2e-6 + t_sig)).max()))
    assert e_rgb.max() <= 2e-5
    assert (e_sig / (1.0 + np.abs(r_sig))).max() <= 2e-5
    # ... and a render through the same planes vs the oracle (fp32)
    from test_gpu_parity import hip_render
    R, Nc, Nf = 20, 32, 32
    cams = synth.camera_sweep(2, -0.2, 0.25)
    o, d = oracle.raygen(cams[:, :16], cams[:, 16:], R)
    noise_c = synth.synth_noise(98, (2, R * R, Nc, 1)); u_f = synth.synth_noise(99, (2 * R * R, Nf))
    ref = oracle.render(planes, tuple(dec), o, d, Nc, Nf, noise_c, u_f)
    got = hip_render(torch, planes, tuple(dec), o, d, Nc, Nf, noise_c, u_f)
    e = (np.abs(got[0] - ref[0]).max(), np.abs(got[1] - ref[1]).max(), np.abs(got[2] - ref[2]).max())
    print("   render: rgb %.2e depth %.2e wsum %.2e" % e)
    assert e[0] <= RGB_TOL and e[2] <= RGB_TOL and e[1] <= DEPTH_TOL
