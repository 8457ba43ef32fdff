"""GPU parity tests, part 2 (`-m gpu`): the configurations and operand ranges the first round left untested.

  * BASELINE config 5 shapes: SR 256^2 -> 512^2 -> 1024^2 against the reference's blocks, the <6,6> (96+96 samples) ray kernel
  * torso / background fusion stacks and to_plane_cnn at their REFERENCE sizes (256^2, 128^2 -> 256^2)
  * dynamic range of the f16x3 path: inputs / weights / styles scaled by 2^k, k in [-20, 14], against torch fp64
    (the reference runs these layers in fp32 with conv_clamp=None: no range limit)
  * byte-exact uint8 frames vs the formula of inference/real3d_infer.py:472,518-522
  * mask_invalid_rays=True, the plane-cache identity hazard

Tolerances: as in test_gpu_parity.py (SR <= 2e-4 * max(1, max|ref|)); range sweeps <= 2e-5 * max|ref| (fp32-rounding class).
"""
import os

import numpy as np
import pytest

from conftest import load_golden
from test_gpu_parity import RGB_TOL, DEPTH_TOL, SR_TOL, T, hip_render, load_block, make_decoder, opts

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from real3dportrait_amd import _lib
    _lib.load()
    return torch


# ------------------------------------------------------------------------------------------------
# BASELINE config 5
# ------------------------------------------------------------------------------------------------
def test_sr_cfg5_golden(torch_cuda):
    """256^2 -> 1024^2 through the two SR blocks (SURVEY 8d cfg 5) vs the reference's SynthesisBlocks (tests/golden/sr_cfg5_a.npz)."""
    torch = torch_cuda
    from real3dportrait_amd import SynthesisBlock, synth
    g = load_golden("sr_cfg5_a")
    seed = int(g["seed"])
    params = synth.synth_sr_params(seed)
    b0 = SynthesisBlock(32, 256, w_dim=512, resolution=512, img_channels=3, is_last=False, conv_clamp=None).cuda()
    b1 = SynthesisBlock(256, 128, w_dim=512, resolution=1024, img_channels=3, is_last=True, conv_clamp=None).cuda()
    load_block(torch, b0, params[0]); load_block(torch, b1, params[1])
    x = T(torch, synth.hash_unitvar(seed, (1, 32, 256, 256), stream=1))
    ws = torch.ones(1, 3, 512, device="cuda")
    x0, r0 = b0(x, x[:, :3].contiguous(), ws, noise_mode="none")
    x1, out = b1(x0, r0, ws, noise_mode="none")
    out = out.cpu().numpy()
    tol = SR_TOL * max(1.0, np.abs(g["strided"]).max())
    assert out.shape == (1, 3, 1024, 1024)
    assert np.abs(out[:, :, ::8, ::8] - g["strided"]).max() <= tol
    assert np.abs(out[:, :, :64, :64] - g["corner"]).max() <= tol
    assert np.abs(out[:, :, -48:, -48:] - g["tail"]).max() <= tol
    assert np.abs(out[:, :, 480:544, 480:544] - g["mid"]).max() <= tol
    assert abs(float(np.abs(out).mean()) - float(g["absmean"])) <= 1e-4


def test_render_cfg5_shape_vs_oracle(torch_cuda, oracle):
    """The <6,6> instantiation (96 + 96 samples per ray, BASELINE config 5) on full-size planes, batch of 2 cameras."""
    from real3dportrait_amd import synth
    R, Nc, Nf, N = 96, 96, 96, 2
    planes = synth.synth_planes(501, N=1).repeat(N, axis=0)
    planes[1] *= np.float32(0.8)
    dec_np = synth.synth_decoder(502, sigma_bias=4.0)
    cams = synth.camera_sweep(N, -0.35, 0.3)
    o, d = oracle.raygen(cams[:, :16], cams[:, 16:], R)
    noise_c = synth.synth_noise(503, (N, R * R, Nc, 1), stream=7)
    u_f = synth.synth_noise(503, (N * R * R, Nf), stream=8)
    ref = oracle.render(planes, dec_np, o, d, Nc, Nf, noise_c, u_f)
    got = hip_render(torch_cuda, planes, dec_np, o, d, Nc, Nf, noise_c, u_f)
    assert np.array_equal(got[3], ref[3])
    assert np.abs(got[0] - ref[0]).max() <= RGB_TOL and np.abs(got[2] - ref[2]).max() <= RGB_TOL
    assert np.abs(got[1] - ref[1]).max() <= DEPTH_TOL


# ------------------------------------------------------------------------------------------------
# fusion stacks / plane producer tail at the reference sizes
# ------------------------------------------------------------------------------------------------
def test_fusion_stacks_full_size_golden(torch_cuda):
    """sr_with_ref.py:101-123 (fuse mode v2) at 256 x 256: multi-tile grids, Cin = 512 multi-stage K loops, image-edge tiles."""
    torch = torch_cuda
    from real3dportrait_amd import synth
    from real3dportrait_amd.superresolution import blend_cat
    from test_gpu_parity import _fusion_modules
    g = load_golden("fusion_full_a")
    seed, R = int(g["seed"]), int(g["R"])
    t = lambda shape, s, gg=1.0: T(torch, synth.hash_unitvar(seed, shape, stream=s) * np.float32(gg))
    x_head, hid, bg, rgb, rgb_torso = t((1, 256, R, R), 1), t((1, 64, R, R), 2), t((1, 3, R, R), 3, 0.5), t((1, 3, R, R), 4, 0.5), t((1, 3, R, R), 5, 0.5)
    alpha = T(torch, synth.synth_noise(seed, (1, 1, R, R), stream=6))
    occ = T(torch, synth.synth_noise(seed, (1, 1, R, R), stream=8))
    ws = T(torch, np.ones((1, 3, 512), np.float32) + synth.hash_unitvar(seed, (1, 3, 512), stream=9) * np.float32(0.1))
    stacks, blk = _fusion_modules(torch, seed, R)
    x_torso = stacks["torso_encoder"](hid)
    x_bg = stacks["bg_encoder"](bg)
    rgb1 = rgb * alpha + rgb_torso * (1 - alpha)
    x1 = stacks["fuse_head_torso_convs"](blend_cat(x_head, x_torso, alpha, stacks["fuse_head_torso_convs"]))
    x2, rgb2 = blk(x1, rgb1, ws, noise_mode="none")
    x3 = stacks["fuse_fg_bg_convs"](blend_cat(x2, x_bg, occ, stacks["fuse_fg_bg_convs"]))
    sl = lambda v: v.cpu().numpy()[:, ::16, ::8, ::8]
    cr = lambda v: v.cpu().numpy()[:, ::32, 120:152, 232:]
    for got, key in ((sl(x_torso), "x_torso"), (sl(x_bg), "x_bg"), (sl(x1), "x1"), (sl(x2), "x2"), (sl(x3), "x3"),
                     (rgb2.cpu().numpy()[:, :, ::4, ::4], "rgb2"), (cr(x1), "x1_crop"), (cr(x3), "x3_crop")):
        ref = g[key]
        assert got.shape == ref.shape, key
        print("fusion stacks at 256^2 [%s] %-8s err %.2e of max(1, max|ref|)" % (os.environ.get("R3D_SR_PRECISION", "default"), key, np.abs(got - ref).max() / max(1.0, np.abs(ref).max())))
        assert np.abs(got - ref).max() <= SR_TOL * max(1.0, np.abs(ref).max()), key


def test_to_plane_cnn_full_size_golden(torch_cuda):
    """to_plane_cnn at 128^2 -> 256^2 (segformer.py:691-700) + flips + cano add + channel-last layout vs tests/golden/toplane_full_a.npz."""
    torch = torch_cuda
    from real3dportrait_amd import ImportanceRenderer, synth
    from real3dportrait_amd.superresolution import Conv2d, ConvStack
    g = load_golden("toplane_full_a")
    seed, r = int(g["seed"]), int(g["r"])
    mods = []
    for i, ((ci, co, k, lrelu), (w, b)) in enumerate(zip(synth.TO_PLANE_CNN, synth.synth_conv_stack(seed, synth.TO_PLANE_CNN, 500))):
        if i == synth.TO_PLANE_CNN_UP_BEFORE:
            mods.append(torch.nn.UpsamplingBilinear2d(scale_factor=2.))
        c = Conv2d(ci, co, k, 1, padding=1)
        with torch.no_grad():
            c.weight.copy_(torch.from_numpy(w)); c.bias.copy_(torch.from_numpy(b))
        mods.append(c)
        if lrelu:
            mods.append(torch.nn.LeakyReLU(0.01))
    cnn = ConvStack(*mods).cuda()
    raw = cnn(T(torch, synth.hash_unitvar(seed, (1, 256, r, r), stream=1)))
    ren = ImportanceRenderer(hp={})
    cano = T(torch, synth.hash_unitvar(seed, (1, 3, 32, 2 * r, 2 * r), stream=2))
    nhwc = ren.prepare_planes(cano, add=raw, add_flip=ren.SECC_PLANE_FLIPS).cpu().numpy()      # [1,3,H,W,32]
    out = np.transpose(nhwc, (0, 1, 4, 2, 3))
    for got, key in ((out[:, :, :, ::8, ::8], "strided"), (out[:, :, ::4, :40, :40], "corner"), (out[:, :, ::4, -40:, -40:], "tail")):
        ref = g[key]
        assert got.shape == ref.shape, key
        assert np.abs(got - ref).max() <= SR_TOL * max(1.0, np.abs(ref).max()), key


# ------------------------------------------------------------------------------------------------
# dynamic range of the f16x3 path
# ------------------------------------------------------------------------------------------------
SWEEP = [-20, -12, -6, 0, 7, 14]


@pytest.mark.parametrize("kx", SWEEP)
@pytest.mark.parametrize("kw", [-20, 0, 14])
def test_conv2d_range_sweep(torch_cuda, kx, kw):
    """nn.Conv2d on the f16x3 kernel with the input scaled by 2^kx and the weights by 2^kw vs torch fp64: the power-of-two folding
    (weight rows at prepack, activations from the measured bound) must keep fp32-class accuracy over the whole sweep."""
    torch = torch_cuda
    from real3dportrait_amd import synth
    from real3dportrait_amd.superresolution import Conv2d
    N, Cin, Cout, H, W = 2, 48, 128, 21, 19
    x = T(torch, synth.hash_unitvar(61, (N, Cin, H, W), stream=1)) * float(2.0 ** kx)
    x[1] *= 2.0 ** -7                                    # per-sample bounds differ
    c = Conv2d(Cin, Cout, 3, 1, padding=1).cuda()
    with torch.no_grad():
        w = T(torch, synth.hash_unitvar(61, (Cout, Cin, 3, 3), stream=2) / np.float32(np.sqrt(Cin * 9.0))) * float(2.0 ** kw)
        w[3] *= 2.0 ** -9; w[7] *= 2.0 ** 6           # rows of very different magnitude
        c.weight.copy_(w)
        c.bias.copy_(T(torch, synth.hash_unitvar(61, (Cout,), stream=3)) * float(2.0 ** (kx + kw)))
    y = c(x, negative_slope=0.2)
    ref = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(x.double().cpu(), c.weight.detach().double().cpu(),
                                                                    c.bias.detach().double().cpu(), padding=1), 0.2)
    for n in range(N):
        err = (y[n].cpu().double() - ref[n]).abs().max().item()
        assert err <= 2e-6 * ref[n].abs().max().item(), (n, err, ref[n].abs().max().item())


@pytest.mark.parametrize("k", SWEEP)
def test_conv_stack_range_sweep(torch_cuda, k):
    """Three chained convs with SPLIT (fp16 hi/lo) hand-offs, input scaled by 2^k, vs torch fp64."""
    torch = torch_cuda
    from real3dportrait_amd import synth
    from real3dportrait_amd.superresolution import Conv2d, ConvStack
    plan = [(3, 64, 3, True), (64, 256, 3, True), (256, 256, 3, False)]        # bg_encoder (sr_with_ref.py:27-33)
    mods, ref_mods = [], []
    for (ci, co, ks, lrelu), (w, b) in zip(plan, synth.synth_conv_stack(63, plan, 300)):
        c = Conv2d(ci, co, ks, 1, padding=ks // 2)
        with torch.no_grad():
            c.weight.copy_(torch.from_numpy(w)); c.bias.copy_(torch.from_numpy(b) * float(2.0 ** k))
        mods.append(c)
        if lrelu:
            mods.append(torch.nn.LeakyReLU())
    st = ConvStack(*mods).cuda()
    x = T(torch, synth.hash_unitvar(63, (1, 3, 40, 36), stream=1)) * float(2.0 ** k)
    y = st(x)
    r = x.double().cpu()
    for m in st:
        r = torch.nn.functional.conv2d(r, m.weight.detach().double().cpu(), m.bias.detach().double().cpu(), padding=m.padding[0]) \
            if isinstance(m, Conv2d) else torch.nn.functional.leaky_relu(r, m.negative_slope)
    err = (y.cpu().double() - r).abs().max().item()
    tier = 2e-5 if st[0].precision == "f16x3" else 1e-4        # library default 'f16mx': measured 2.2e-5 .. 2.6e-5 (tests/test_gpu_f16x3.py re-runs this on f16x3)
    print("conv stack range sweep [%s] 2^%d: %.2e of max|ref|" % (st[0].precision, k, err / r.abs().max().item()))
    assert err <= tier * r.abs().max().item(), (err, r.abs().max().item())


def _block_fp64(torch, p, x, img, ws3, up, clamp):
    """torch fp64 restatement of SynthesisBlock / SynthesisBlockNoUp (networks_stylegan2.py:37-94,322-342,365-370,429-473) for one
    batch; same structure as oracle/r3d_oracle.c (which is fp32)."""
    Fn = torch.nn.functional
    D = lambda a: torch.from_numpy(np.asarray(a, np.float64))
    x, img, ws3 = x.double(), img.double(), ws3.double()
    f1 = torch.tensor([1.0, 3.0, 3.0, 1.0], dtype=torch.float64)
    f2 = (torch.outer(f1, f1) / 64.0 * 4.0)[None, None]

    def styles(layer, w):
        aw, ab = D(p[layer][2]), D(p[layer][3])
        return w @ (aw / np.sqrt(aw.shape[1])).t() + ab

    def modconv(xx, layer, w, upsample):
        W_ = D(p[layer][0]); b = D(p[layer][1])
        s = styles(layer, w)                                            # [N, Ci]
        outs = []
        for n in range(xx.shape[0]):
            wm = W_ * s[n][None, :, None, None]
            wm = wm * torch.rsqrt((wm ** 2).sum(dim=(1, 2, 3), keepdim=True) + 1e-8)
            if upsample:
                t = Fn.conv_transpose2d(xx[n:n + 1], wm.transpose(0, 1), stride=2)
                Co = t.shape[1]
                t = Fn.conv2d(Fn.pad(t, (1, 1, 1, 1)), f2.repeat(Co, 1, 1, 1), groups=Co)
            else:
                t = Fn.conv2d(xx[n:n + 1], wm, padding=1)
            t = t + b[None, :, None, None]
            t = Fn.leaky_relu(t, 0.2) * np.sqrt(2.0)
            if clamp is not None:
                t = t.clamp(-clamp, clamp)
            outs.append(t)
        return torch.cat(outs)
    y = modconv(x, "conv0", ws3[:, 0], up)
    y = modconv(y, "conv1", ws3[:, 1], False)
    Wr, br = D(p["torgb"][0]), D(p["torgb"][1])
    s = styles("torgb", ws3[:, 2]) / np.sqrt(Wr.shape[1])
    rgb = torch.cat([Fn.conv2d(y[n:n + 1], Wr * s[n][None, :, None, None]) for n in range(y.shape[0])]) + br[None, :, None, None]
    if clamp is not None:
        rgb = rgb.clamp(-clamp, clamp)
    if up:
        N_, C_, H_, W_2 = img.shape
        u = torch.zeros(N_, C_, 2 * H_, 2 * W_2, dtype=torch.float64)
        u[:, :, ::2, ::2] = img
        img = Fn.conv2d(Fn.pad(u, (2, 1, 2, 1)), f2.repeat(3, 1, 1, 1), groups=3)
    return y, img + rgb


@pytest.mark.parametrize("what", ["input", "weights", "styles", "bias"])
@pytest.mark.parametrize("k", SWEEP)
@pytest.mark.parametrize("up", [True, False])
def test_sr_block_range_sweep(torch_cuda, what, k, up):
    """SynthesisBlock / SynthesisBlockNoUp with one operand class scaled by 2^k (block input; conv weights; the affine layers that
    produce the styles; the biases) vs torch fp64.  The reference computes these layers in fp32 without clamps, so every case must
    keep fp32-class accuracy -- none may saturate at the fp16 limits."""
    torch = torch_cuda
    from real3dportrait_amd import synth
    from real3dportrait_amd.superresolution import SynthesisBlock, SynthesisBlockNoUp
    N, Cin, Cout, H, W = 2, 32, 128, 18, 14
    sc = np.float32(2.0 ** k)
    p = {kk: tuple(np.array(a) for a in v) for kk, v in synth.synth_sr_block(71, Cin, Cout, 512, 700).items()}
    if what == "weights":
        for layer in ("conv0", "conv1"):
            p[layer] = (p[layer][0] * sc,) + p[layer][1:]
    elif what == "styles":
        for layer in ("conv0", "conv1", "torgb"):
            w_, b_, aw, ab = p[layer]
            p[layer] = (w_, b_, aw * sc, ab * sc)
    elif what == "bias":
        for layer in ("conv0", "conv1"):
            w_, b_, aw, ab = p[layer]
            p[layer] = (w_, b_ * sc, aw, ab)
    blk = (SynthesisBlock if up else SynthesisBlockNoUp)(Cin, Cout, w_dim=512, resolution=2 * H if up else H, img_channels=3,
                                                         is_last=False, conv_clamp=None).cuda()
    load_block(torch, blk, p)
    blk.precision = "f16x3"           # this test states the fp32-class tier; the default 'f16mx' has its own sweep (tests/test_gpu_mx.py)
    x = synth.hash_unitvar(72, (N, Cin, H, W), stream=1) * (sc if what == "input" else np.float32(1.0))
    img = synth.hash_unitvar(72, (N, 3, H, W), stream=2) * np.float32(0.5)
    ws = np.ones((N, 3, 512), np.float32) + synth.hash_unitvar(72, (N, 3, 512), stream=3) * np.float32(0.2)
    xo, io = blk(T(torch, x), T(torch, img), T(torch, ws), noise_mode="none")
    rx, ri = _block_fp64(torch, p, torch.from_numpy(x), torch.from_numpy(img), torch.from_numpy(ws), up, None)
    ex = (xo.cpu().double() - rx).abs().max().item() / max(rx.abs().max().item(), 1e-300)
    ei = (io.cpu().double() - ri).abs().max().item() / max(ri.abs().max().item(), 1e-300)
    assert torch.isfinite(xo).all() and torch.isfinite(io).all()
    assert ex <= 4e-6 and ei <= 4e-6, (what, k, up, ex, ei)      # measured worst of the whole sweep: 1.3e-6 (profiles/r03/sr_sweep_errors.txt)


def test_bounds_are_upper_bounds_and_stored_operands_fit_fp16(torch_cuda):
    """The folded multipliers keep every stored fp16 operand below 2^15 and the propagated bound really bounds the activations."""
    torch = torch_cuda
    from real3dportrait_amd import synth
    from real3dportrait_amd.superresolution import SynthesisBlock
    p = synth.synth_sr_block(81, 32, 128, 512, 700)
    blk = SynthesisBlock(32, 128, w_dim=512, resolution=32, img_channels=3, is_last=False, conv_clamp=None).cuda()
    load_block(torch, blk, p)
    nxt = SynthesisBlock(128, 128, w_dim=512, resolution=64, img_channels=3, is_last=True, conv_clamp=None).cuda()
    load_block(torch, nxt, synth.synth_sr_block(82, 128, 128, 512, 700))
    from real3dportrait_amd.superresolution import chain_fold, _BoundMeter
    x = T(torch, synth.hash_unitvar(83, (1, 32, 16, 16), stream=1)) * 300.0
    ws = torch.ones(1, 3, 512, device="cuda")
    blk.prepare(ws, x.device); nxt.prepare(ws, x.device)
    bound = _BoundMeter()(x)
    assert abs(float(bound[0]) - float(x.abs().max())) == 0.0
    chain_fold([blk.chain_op(-1), nxt.chain_op(0)], 1, [bound])
    blk.out_format = "split"
    xs, _ = blk(x, x[:, :3].contiguous(), ws, noise_mode="none", _next=nxt, _folded=True)     # SPLIT, scaled for nxt
    hi = xs[:, 0].float().abs().max().item()
    assert hi < 2.0 ** 15 and hi > 2.0 ** 4, hi                                              # inside the window, not saturated
    blk.out_format = "nchw"
    xo, _ = blk(x, x[:, :3].contiguous(), ws, noise_mode="none")
    assert float(blk.bound_out(1)[0]) >= float(xo.abs().max())


# ------------------------------------------------------------------------------------------------
# output side
# ------------------------------------------------------------------------------------------------
def _u8_reference(torch, imgs):
    """inference/real3d_infer.py:517-519: imgs.clamp(-1,1); ((imgs.permute(0,2,3,1) + 1)/2 * 255).int().cpu().numpy().astype(np.uint8)"""
    imgs = imgs.cpu().clamp(-1, 1)
    return ((imgs.permute(0, 2, 3, 1) + 1) / 2 * 255).int().numpy().astype(np.uint8)


def test_frames_to_u8_is_byte_exact(torch_cuda):
    torch = torch_cuda
    from real3dportrait_amd import _lib
    lib = _lib.load()
    torch.manual_seed(5)
    img = (torch.rand(2, 3, 64, 48, device="cuda") * 2.6 - 1.3)                        # beyond [-1, 1] on both sides
    special = torch.tensor([-1.0, 1.0, 0.0, -0.0, 1.0 - 2 ** -24, -1.0 + 2 ** -24, 0.999999, -0.999999, 2 ** -30, 7.0, -7.0,
                            float("inf"), float("-inf")], device="cuda")
    img.view(-1)[:special.numel()] = special
    k = torch.arange(256, device="cuda", dtype=torch.float32)
    edges = torch.cat([k / 127.5 - 1.0, torch.nextafter(k / 127.5 - 1.0, torch.tensor(9.0, device="cuda")),
                       torch.nextafter(k / 127.5 - 1.0, torch.tensor(-9.0, device="cuda"))])   # every uint8 bucket boundary +- 1 ulp
    img.view(-1)[100:100 + edges.numel()] = edges
    out = torch.empty(2, 64, 48, 3, dtype=torch.uint8, device="cuda")
    _lib.check(lib.r3d_frames_to_u8(_lib.ptr(img), 2, 64, 48, _lib.ptr(out), _lib.stream_ptr()), "frames_to_u8")
    assert np.array_equal(out.cpu().numpy(), _u8_reference(torch, img))
    nan = torch.full((1, 3, 4, 4), float("nan"), device="cuda")
    o2 = torch.empty(1, 4, 4, 3, dtype=torch.uint8, device="cuda")
    _lib.check(lib.r3d_frames_to_u8(_lib.ptr(nan), 1, 4, 4, _lib.ptr(o2), _lib.stream_ptr()), "frames_to_u8")
    assert int(o2.max()) == 0                                                          # NaN -> 0 (int(NaN) then uint8 wrap gives 0 upstream too)


def test_fused_u8_epilogue_equals_reference_formula(torch_cuda):
    """The uint8 frame written by the last SR block's toRGB kernel == the reference's conversion of the fp32 image, byte for byte."""
    torch = torch_cuda
    from real3dportrait_amd import SuperresolutionHybrid8XDC, synth
    sr = SuperresolutionHybrid8XDC(channels=32, img_resolution=512, sr_num_fp16_res=0, sr_antialias=True).cuda()
    params = synth.synth_sr_params(9)
    load_block(torch, sr.block0, params[0]); load_block(torch, sr.block1, params[1])
    x = T(torch, synth.hash_unitvar(9, (1, 32, 128, 128), stream=1))
    ws = torch.ones(1, 14, 512, device="cuda")
    u8 = torch.empty(1, 512, 512, 3, dtype=torch.uint8, device="cuda")
    img = sr(x[:, :3].contiguous(), x, ws, noise_mode="none", _u8_out=u8)
    assert np.array_equal(u8.cpu().numpy(), _u8_reference(torch, img))
    u8b = torch.empty_like(u8)
    assert sr(x[:, :3].contiguous(), x, ws, noise_mode="none", _u8_out=u8b, _need_img=False) is None
    assert torch.equal(u8, u8b)
    frac_sat = float(((u8 == 0) | (u8 == 255)).float().mean())
    assert 0.01 < frac_sat < 0.99                                                      # the clamp really acts on part of the image


# ------------------------------------------------------------------------------------------------
# generator shell
# ------------------------------------------------------------------------------------------------
def _generator(torch, seed, hp=None):
    from real3dportrait_amd import TriPlaneGenerator, synth
    G = TriPlaneGenerator(hp=hp).cuda()
    dec_np = synth.synth_decoder(seed, sigma_bias=4.0)
    with torch.no_grad():
        G.decoder.net[0].weight.copy_(T(torch, dec_np[0])); G.decoder.net[0].bias.copy_(T(torch, dec_np[1]))
        G.decoder.net[2].weight.copy_(T(torch, dec_np[2])); G.decoder.net[2].bias.copy_(T(torch, dec_np[3]))
    params = synth.synth_sr_params(seed)
    load_block(torch, G.superresolution.block0, params[0]); load_block(torch, G.superresolution.block1, params[1])
    G._last_planes = T(torch, synth.synth_planes(seed, N=1)).view(1, 96, 256, 256)
    return G


def test_synthesis_mask_invalid_rays_golden(torch_cuda):
    """hparams['mask_invalid_rays'] = True (triplane.py:123-126) with 35 % of the rays missing the box, against the reference."""
    torch = torch_cuda
    from real3dportrait_amd import synth
    g = load_golden("synthesis_mask_a")
    seed, R, Nc, Nf = int(g["seed"]), int(g["R"]), int(g["Nc"]), int(g["Nf"])
    G = _generator(torch, seed, hp={"mask_invalid_rays": True})
    G.renderer.noise_override = (T(torch, synth.synth_noise(seed, (1, R * R, Nc, 1), stream=7)),
                                 T(torch, synth.synth_noise(seed, (R * R, Nf), stream=8)))
    out = G.synthesis(torch.ones(1, 14, 512, device="cuda"), T(torch, g["cam"]), use_cached_backbone=True, noise_mode="none")
    raw = out["image_raw"].cpu().numpy()
    assert abs(float((raw == -1).mean()) - float(g["masked_frac"])) < 1e-6 and float(g["masked_frac"]) > 0.2
    assert np.abs(raw - g["image_raw"]).max() <= RGB_TOL
    assert np.abs(out["image_depth"].cpu().numpy() - g["image_depth"]).max() <= DEPTH_TOL
    assert np.abs(out["image_feature"].cpu().numpy()[:, ::4] - g["image_feature_strided"]).max() <= RGB_TOL
    assert np.abs(out["image"].cpu().numpy()[:, :, ::4, ::4] - g["image_strided"]).max() <= SR_TOL * max(1.0, np.abs(g["image_strided"]).max())
    assert out["weights_img"].shape == (1, 1, R, R)


def test_plane_cache_is_keyed_on_the_tensor_not_its_address(torch_cuda):
    """A freed planes tensor's address is normally handed to the next allocation of the same size (the drop-in path builds
    `cano + secc` afresh every frame, always at version 0): the renderer must not serve the previous frame's layout."""
    torch = torch_cuda
    from real3dportrait_amd import ImportanceRenderer, RaySampler, synth
    dec = make_decoder(torch, synth.synth_decoder(6, sigma_bias=3.0))
    cam = T(torch, synth.look_at_camera(0.1, 0.0)[None])
    o, d = RaySampler()(cam[:, :16].view(-1, 4, 4), cam[:, 16:].view(-1, 3, 3), 24)
    ren = ImportanceRenderer(hp={})
    ren.noise_mode, ren.seed = "hash", 3
    base = T(torch, synth.synth_planes(5, N=1, H=64, W=64))
    p1 = base + 0.0
    a = ren(p1, dec, o, d, opts(16, 16))[0].clone()
    addr = p1.data_ptr()
    del p1
    p2 = base * 0.5
    same_address = p2.data_ptr() == addr
    b = ren(p2, dec, o, d, opts(16, 16))[0].clone()
    fresh = ImportanceRenderer(hp={})
    fresh.noise_mode, fresh.seed = "hash", 3
    want = fresh(p2, dec, o, d, opts(16, 16))[0]
    assert torch.equal(b, want) and not torch.equal(a, b), "stale plane layout served (address reuse: %s)" % same_address
    # a static tensor IS served from the cache (same object, same version) and an in-place update invalidates it
    c1 = ren._planes_nhwc(p2)
    assert ren._planes_nhwc(p2) is c1
    p2.mul_(2.0)
    assert ren._planes_nhwc(p2) is not c1


def test_rccl_gather_frames_single_rank(torch_cuda):
    """r3d_comm_* / r3d_gather_frames (grouped ncclSend / ncclRecv) on a 1-rank communicator: the root receives its own ring.  (The
    multi-rank path cannot run on a 1-GPU box; the sharding arithmetic is covered by the gloo tests.)"""
    import ctypes
    torch = torch_cuda
    from real3dportrait_amd import _lib
    lib = _lib.load()
    uid = ctypes.create_string_buffer(128)
    _lib.check(lib.r3d_comm_unique_id(uid), "comm_unique_id")
    comm = ctypes.c_void_p()
    _lib.check(lib.r3d_comm_init(uid, 0, 1, ctypes.byref(comm)), "comm_init")
    local = (torch.arange(3 * 64 * 64 * 3, device="cuda") % 251).to(torch.uint8)
    root = torch.zeros_like(local)
    _lib.check(lib.r3d_gather_frames(comm, _lib.ptr(local), local.numel(), _lib.ptr(root), 0, _lib.stream_ptr()), "gather_frames")
    torch.cuda.synchronize()
    assert torch.equal(root, local)
    assert lib.r3d_gather_frames(comm, _lib.ptr(local), local.numel(), None, 0, _lib.stream_ptr()) == -1     # root without a buffer
    _lib.check(lib.r3d_comm_destroy(comm), "comm_destroy")


# ---- the renderer's decoder on the f16 matrix pipe: exact power-of-two range fold (csrc/r3d_render.hip decoder_fold_kernel) ----------------
def _render_case(torch, planes, dec_np, Nc=24, Nf=24, R=16, tagged=True):
    from real3dportrait_amd import ImportanceRenderer, OSGDecoder, RaySampler, synth
    cam = torch.from_numpy(synth.look_at_camera(0.1, -0.05)[None]).cuda()
    o, d = RaySampler()(cam[:, :16].view(-1, 4, 4), cam[:, 16:].view(-1, 3, 3), R)
    dec = OSGDecoder().cuda()
    with torch.no_grad():
        dec.net[0].weight.copy_(torch.from_numpy(dec_np[0])); dec.net[0].bias.copy_(torch.from_numpy(dec_np[1]))
        dec.net[2].weight.copy_(torch.from_numpy(dec_np[2])); dec.net[2].bias.copy_(torch.from_numpy(dec_np[3]))
    ren = ImportanceRenderer(hp={})
    noise_c = synth.synth_noise(31, (1, R * R, Nc, 1)); u_f = synth.synth_noise(32, (R * R, Nf))
    ren.noise_override = (torch.from_numpy(noise_c).cuda(), torch.from_numpy(u_f).cuda())
    options = {"ray_start": "auto", "ray_end": "auto", "box_warp": 1.0, "depth_resolution": Nc, "depth_resolution_importance": Nf,
               "disparity_space_sampling": False, "clamp_mode": "softplus", "white_back": False}
    p = torch.from_numpy(planes).cuda()
    if not tagged:          # a caller that did its own layout: no |max| partials, the bound is measured inside r3d_render_forward
        nhwc = ren.prepare_planes(p).clone()
        nhwc._r3d_nhwc = True
        p = nhwc
    rgb, depth, wsum, valid = ren(p, dec, o, d, options)
    torch.cuda.synchronize()
    return (rgb.cpu().numpy(), depth.cpu().numpy(), wsum.cpu().numpy()), (o.cpu().numpy(), d.cpu().numpy(), noise_c, u_f)


@pytest.mark.parametrize("k", [-16, -12, -8, -4, 0, 4, 8, 12])
def test_render_range_sweep_planes_vs_first_layer(torch_cuda, oracle, k):
    """planes * 2^k with decoder.net.0.weight * 2^-k is the same function (exact power-of-two factors: the fp32 reference / oracle gives
    bit-identical pre-activations for every k).  Without the range fold the fp16 hi/lo split of the gathered features (k = -16: the lo
    terms fall under the fp16 subnormal step) or of the weights loses up to 2e-3; with it every k must match the oracle like k = 0."""
    torch = torch_cuda
    from real3dportrait_amd import synth
    planes = synth.synth_planes(41, N=1, H=64, W=64)
    dec = list(synth.synth_decoder(42, sigma_bias=3.0))
    s = np.float32(2.0 ** k)
    dec_k = [dec[0] / s, dec[1], dec[2], dec[3]]
    (rgb, depth, wsum), (o, d, noise_c, u_f) = _render_case(torch, planes * s, dec_k, tagged=(k % 8 != 4))
    ref = oracle.render(planes, tuple(dec), o, d, 24, 24, noise_c, u_f)
    e_rgb, e_d, e_w = np.abs(rgb - ref[0]).max(), np.abs(depth - ref[1]).max(), np.abs(wsum - ref[2]).max()
    print("k=%+d: rgb %.2e depth %.2e wsum %.2e" % (k, e_rgb, e_d, e_w))
    assert e_rgb <= 2e-5 and e_w <= 2e-5 and e_d <= 2e-5


def _run_model_fp64(planes, dec, coords, box_warp=1.0):
    """Point queries in float64 numpy (sample_from_planes + OSGDecoder, renderer.py:49-75, models/triplane.py:177-189), plus a bound on the
    error allowed: 64 fp32 ulp of the absolute accumulations, propagated through both layers (an fp32 evaluation carries ~8; the 3-term
    fp16 split represents each operand to 2^-22 and drops the lo*lo products, i.e. ~2^-21 relative per dot product)."""
    P = planes.astype(np.float64)
    N, _, C, H, W = P.shape
    w1, b1, w2, b2 = (a.astype(np.float64) for a in dec)
    q = coords.astype(np.float64) * (2.0 / box_warp)
    uv = [(q[..., 0], q[..., 1]), (q[..., 0], q[..., 2]), (q[..., 2], q[..., 0])]
    feat = np.zeros(q.shape[:2] + (C,))
    for p, (u, v) in enumerate(uv):
        ix, iy = ((u + 1) * W - 1) / 2, ((v + 1) * H - 1) / 2
        x0, y0 = np.floor(ix).astype(np.int64), np.floor(iy).astype(np.int64)
        for dy in (0, 1):
            for dx in (0, 1):
                xx, yy = x0 + dx, y0 + dy
                wgt = (1 - np.abs(ix - xx)) * (1 - np.abs(iy - yy))
                ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
                for n in range(N):
                    t = P[n, p][:, np.clip(yy[n], 0, H - 1), np.clip(xx[n], 0, W - 1)].T          # [npts, C]
                    feat[n] += np.where(ok[n][:, None], t * wgt[n][:, None], 0.0)
    x = feat / 3.0
    pre = x @ (w1.T / np.sqrt(32.0)) + b1
    h = np.logaddexp(0.0, pre)
    y = h @ (w2.T / 8.0) + b2
    eps = 64 * 2.0 ** -24
    dpre = eps * (np.abs(x) @ np.abs(w1.T / np.sqrt(32.0)) + np.abs(b1))
    dy = eps * (np.abs(h) @ np.abs(w2.T / 8.0) + np.abs(b2)) + dpre @ np.abs(w2.T / 8.0)
    rgb = 1.002 / (1.0 + np.exp(-y[..., 1:])) - 0.001
    return rgb, y[..., :1], 0.2505 * dy[..., 1:], dy[..., :1]


@pytest.mark.parametrize("which,k", [("w1", 10), ("w1", 15), ("w1", -14), ("w2", -16), ("w2", 9), ("w2", 16), ("b1", 14), ("planes", 18), ("planes", -30),
                                     ("none", 0)])
def test_decoder_range_sweep_magnitudes_fp64(torch_cuda, which, k):
    """One operand scaled on its own (a different function each time): point queries against a float64 evaluation, with the tolerance an
    fp32 evaluation of the same formulas is entitled to (the reference computes in fp32).  Exercises the optional factors of the fold:
    hidden values above 2^15 (2^c), colour rows above 2^15 (2^d), plane / weight magnitudes far from 1 -- no clamping, no overflow."""
    torch = torch_cuda
    from real3dportrait_amd import ImportanceRenderer, OSGDecoder, synth
    planes = synth.synth_planes(43, N=2, H=32, W=32)
    dec = list(synth.synth_decoder(44, sigma_bias=2.0))
    coords = ((synth.synth_noise(48, (2, 700, 3)) - 0.5) * 1.2).astype(np.float32)
    s = np.float32(2.0 ** k)
    if which == "planes":
        planes = planes * s
    elif which != "none":
        i = {"w1": 0, "b1": 1, "w2": 2}[which]
        dec[i] = dec[i] * s
    decm = OSGDecoder().cuda()
    with torch.no_grad():
        decm.net[0].weight.copy_(torch.from_numpy(dec[0])); decm.net[0].bias.copy_(torch.from_numpy(dec[1]))
        decm.net[2].weight.copy_(torch.from_numpy(dec[2])); decm.net[2].bias.copy_(torch.from_numpy(dec[3]))
    out = ImportanceRenderer(hp={}).run_model(torch.from_numpy(planes).cuda(), decm, torch.from_numpy(coords).cuda(), None, {"box_warp": 1.0})
    rgb, sig = out["rgb"].cpu().numpy().astype(np.float64), out["sigma"].cpu().numpy().astype(np.float64)
    assert np.isfinite(rgb).all() and np.isfinite(sig).all()
    r_rgb, r_sig, t_rgb, t_sig = _run_model_fp64(planes, dec, coords)
    e_rgb, e_sig = np.abs(rgb - r_rgb), np.abs(sig - r_sig)
    worst_rgb, worst_sig = (e_rgb / (2e-6 + t_rgb)).max(), (e_sig / (2e-6 + t_sig)).max()
    print("%s * 2^%+d: rgb err %.2e (%.2f of the allowance), sigma err %.2e (%.2f)" % (which, k, e_rgb.max(), worst_rgb, e_sig.max(), worst_sig))
    assert worst_rgb <= 1.0 and worst_sig <= 1.0


@pytest.mark.parametrize("k", [-16, 0, 12])
def test_run_model_range_sweep(torch_cuda, oracle, k):
    torch = torch_cuda
    from real3dportrait_amd import ImportanceRenderer, OSGDecoder, synth
    planes = synth.synth_planes(45, N=2, H=32, W=32)
    dec = list(synth.synth_decoder(46))
    coords = ((synth.synth_noise(47, (2, 500, 3)) - 0.5) * 1.2).astype(np.float32)
    s = np.float32(2.0 ** k)
    decm = OSGDecoder().cuda()
    with torch.no_grad():
        decm.net[0].weight.copy_(torch.from_numpy(dec[0] / s)); decm.net[0].bias.copy_(torch.from_numpy(dec[1]))
        decm.net[2].weight.copy_(torch.from_numpy(dec[2])); decm.net[2].bias.copy_(torch.from_numpy(dec[3]))
    out = ImportanceRenderer(hp={}).run_model(torch.from_numpy(planes * s).cuda(), decm, torch.from_numpy(coords).cuda(), None, {"box_warp": 1.0})
    ref = oracle.run_model(planes, tuple(dec), coords)
    e_rgb, e_s = np.abs(out["rgb"].cpu().numpy() - ref[0]).max(), np.abs(out["sigma"].cpu().numpy().reshape(ref[1].shape) - ref[1]).max()
    print("run_model k=%+d: rgb %.2e sigma %.2e" % (k, e_rgb, e_s))
    assert e_rgb <= 2e-5 and e_s <= 2e-4
