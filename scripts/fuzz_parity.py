"""Randomised parity sweep (GPU): ray kernel vs the C oracle, plain conv layers vs torch fp64, SR blocks vs the oracle, over
random shapes / options.  Not part of pytest (minutes of CPU oracle time); prints one line per case and a summary."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import Oracle
from real3dportrait_amd import ImportanceRenderer, OSGDecoder, synth
from real3dportrait_amd.superresolution import Conv2d, SynthesisBlock, SynthesisBlockNoUp
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 12
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
orc = Oracle(); bad = 0


def opts(Nc, Nf, bw, wb):
    return {"ray_start": "auto", "ray_end": "auto", "box_warp": bw, "depth_resolution": Nc, "depth_resolution_importance": Nf,
            "disparity_space_sampling": False, "clamp_mode": "softplus", "white_back": wb}


for c in range(ncase):
    # ---- ray kernel ---------------------------------------------------------------------------------------------
    R = int(rng.integers(5, 28)); Nc = int(rng.integers(4, 97)); Nf = int(rng.choice([0, rng.integers(1, 97)]))
    N = int(rng.integers(1, 4)); HW = int(rng.choice([8, 17, 32, 48])); D = int(rng.choice([1, 1, 3, 2]))
    bw = float(rng.choice([1.0, 0.8, 1.3])); wb = bool(rng.integers(0, 2)); seed = int(rng.integers(1, 1 << 20))
    planes = synth.hash_unitvar(seed, (N, 3, 32 * D, HW, HW), stream=1)
    dec = synth.synth_decoder(seed + 1, sigma_bias=float(rng.choice([0.0, 3.0])))
    cams = synth.camera_sweep(N, -0.3, 0.35, pitch=float(rng.uniform(-0.2, 0.2)))
    if rng.integers(0, 3) == 0:
        cams[:, 3] += 0.4                 # push the camera sideways: some rays miss the box
    o, d = orc.raygen(cams[:, :16], cams[:, 16:], R)
    noise_c = synth.synth_noise(seed + 2, (N, R * R, Nc, 1)); u_f = synth.synth_noise(seed + 3, (N * R * R, max(Nf, 1)))[:, :Nf]
    ref = orc.render(planes, dec, o, d, Nc, Nf, noise_c, u_f, bw, wb, triplane_depth=D)
    dm = OSGDecoder().cuda()
    with torch.no_grad():
        dm.net[0].weight.copy_(T(dec[0])); dm.net[0].bias.copy_(T(dec[1])); dm.net[2].weight.copy_(T(dec[2])); dm.net[2].bias.copy_(T(dec[3]))
    ren = ImportanceRenderer(hp={} if D == 1 else {"triplane_feature_type": "trigrid", "triplane_depth": D})
    ren.noise_override = (T(noise_c), T(u_f) if Nf > 0 else None)
    got = [t.cpu().numpy() for t in ren(T(planes), dm, T(o), T(d), opts(Nc, Nf, bw, wb))]
    e = (np.abs(got[0] - ref[0]).max(), np.abs(got[1] - ref[1]).max(), np.abs(got[2] - ref[2]).max(), not np.array_equal(got[3], ref[3]))
    ok = e[0] <= 2e-4 and e[1] <= 1e-4 and e[2] <= 2e-4 and not e[3]; bad += not ok
    print("render R=%d %d+%d N=%d HW=%d D=%d bw=%.1f wb=%d: rgb %.1e depth %.1e wsum %.1e valid_mismatch %s %s" % (R, Nc, Nf, N, HW, D, bw, wb, *e, "ok" if ok else "FAIL"), flush=True)
    # ---- plain conv ---------------------------------------------------------------------------------------------
    Ci = int(rng.choice([1, 3, 7, 16, 24, 64, 130])); Co = int(rng.choice([4, 12, 32, 96, 128, 260])); k = int(rng.choice([1, 3]))
    H = int(rng.integers(1, 40)); W = int(rng.integers(1, 40)); Nb = int(rng.integers(1, 4)); slope = rng.choice([None, 0.01, 0.2])
    cv = Conv2d(Ci, Co, k, 1, padding=k // 2).cuda()
    x = T(synth.hash_unitvar(seed + 5, (Nb, Ci, H, W), stream=1))
    with torch.no_grad():
        cv.weight.copy_(T(synth.hash_unitvar(seed + 6, (Co, Ci, k, k), stream=2) / np.float32(np.sqrt(Ci * k * k)))); cv.bias.copy_(T(synth.hash_unitvar(seed + 7, (Co,), stream=3)))
    y = cv(x, negative_slope=slope)
    r = torch.nn.functional.conv2d(x.double().cpu(), cv.weight.detach().double().cpu(), cv.bias.detach().double().cpu(), padding=k // 2)
    if slope is not None: r = torch.nn.functional.leaky_relu(r, float(slope))
    err = (y.cpu().double() - r).abs().max().item(); ok = err <= 2e-5 * max(1.0, r.abs().max().item()); bad += not ok
    print("conv2d N=%d %d->%d k=%d %dx%d slope=%s: err %.1e %s" % (Nb, Ci, Co, k, H, W, slope, err, "ok" if ok else "FAIL"), flush=True)
    # ---- SR block -----------------------------------------------------------------------------------------------
    Ci = int(rng.choice([16, 32, 48, 64])); Co = int(rng.choice([128, 256])); H = int(rng.integers(3, 30)); W = int(rng.integers(3, 30))
    up = bool(rng.integers(0, 2)); clamp = rng.choice([None, 1.5, 256.0]); Nb = int(rng.integers(1, 3))
    params = synth.synth_sr_block(seed + 8, Ci, Co, 512, 700)
    blk = (SynthesisBlock if up else SynthesisBlockNoUp)(Ci, Co, w_dim=512, resolution=8, img_channels=3, is_last=False, conv_clamp=clamp).cuda()
    with torch.no_grad():
        for name in ("conv0", "conv1", "torgb"):
            l = getattr(blk, name); w, b, aw, ab = params[name]
            l.weight.copy_(T(w)); l.bias.copy_(T(b)); l.affine.weight.copy_(T(aw)); l.affine.bias.copy_(T(ab))
    x = synth.hash_unitvar(seed + 9, (Nb, Ci, H, W), stream=1); img = synth.hash_unitvar(seed + 9, (Nb, 3, H, W), stream=2) * np.float32(0.5)
    ws = np.ones((Nb, 3, 512), np.float32) + synth.hash_unitvar(seed + 9, (Nb, 3, 512), stream=3) * np.float32(0.2)
    xo, io = blk(T(x), T(img), T(ws), noise_mode="none"); xo, io = xo.cpu().numpy(), io.cpu().numpy()
    err = 0.0
    for n in range(Nb):
        rx, ri = orc.sr_block(x[n], img[n], params, ws[n], clamp=None if clamp is None else float(clamp), up=up)
        err = max(err, np.abs(xo[n] - rx).max() / max(1.0, np.abs(rx).max()), np.abs(io[n] - ri).max() / max(1.0, np.abs(ri).max()))
    ok = err <= 2e-4; bad += not ok
    print("sr_block up=%d N=%d %d->%d %dx%d clamp=%s: rel err %.1e %s" % (up, Nb, Ci, Co, H, W, clamp, err, "ok" if ok else "FAIL"), flush=True)
    # ---- round 6: the fused blend + 1x1 conv (r3d_conv_forward_blend) vs fp64 and vs the two-step form, and a plain 3x3 conv on a Winograd-eligible shape -------------
    from real3dportrait_amd.superresolution import _fold_single, blend_cat
    Cs = 64 * int(rng.integers(1, 7)); Ca = 8 * int(rng.integers(1, Cs // 8)); Cb = Cs - Ca; Co = int(rng.choice([4, 12, 64, 128, 200])); H = int(rng.integers(1, 40)); W = int(rng.integers(1, 40))
    Nb = int(rng.integers(1, 4)); slope = rng.choice([None, 0.01, 0.2]); kk = int(rng.integers(-12, 13))
    a = synth.hash_unitvar(seed + 11, (Nb, Ca, H, W), stream=1) * np.float32(2.0 ** kk); b = synth.hash_unitvar(seed + 11, (Nb, Cb, H, W), stream=2) * np.float32(2.0 ** kk)
    m = np.abs(synth.hash_unitvar(seed + 11, (Nb, 1, H, W), stream=3)).clip(0, 1).astype(np.float32)

    def cb8(x):
        t = T(x); n_, c_, h_, w_ = t.shape
        y8 = t.view(n_, c_ // 8, 8, h_, w_).permute(0, 1, 3, 4, 2).contiguous(); y8._r3d_fmt = "cb8"
        y8._r3d_bound, y8._r3d_depth = t.abs().amax(dim=(1, 2, 3)).contiguous(), 0
        return y8
    cv = Conv2d(Cs, Co, 1, 1, padding=0).cuda(); cv.precision = "f16x3"
    with torch.no_grad():
        cv.weight.copy_(T(synth.hash_unitvar(seed + 12, (Co, Cs, 1, 1), stream=2) / np.float32(np.sqrt(Cs)))); cv.bias.copy_(T(synth.hash_unitvar(seed + 12, (Co,), stream=3)))
    a8, b8, mt = cb8(a), cb8(b), T(m)
    _fold_single(cv, Nb, a8.device, [a8._r3d_bound, b8._r3d_bound], negative_slope=None if slope is None else float(slope))
    y2 = cv(blend_cat(a8, b8, mt, cv, _folded_head=cv), negative_slope=None if slope is None else float(slope))
    y1 = cv(None, negative_slope=None if slope is None else float(slope), _folded=True, _blend=(a8, b8, mt))
    xd = torch.cat([torch.from_numpy(a).double() * torch.from_numpy(m).double(), torch.from_numpy(b).double() * (1 - torch.from_numpy(m).double())], dim=1)
    r = torch.nn.functional.conv2d(xd, cv.weight.detach().double().cpu(), cv.bias.detach().double().cpu())
    if slope is not None: r = torch.nn.functional.leaky_relu(r, float(slope))
    err = (y1.cpu().double() - r).abs().max().item() / max(2.0 ** kk, r.abs().max().item()); same = torch.equal(y1, y2); ok = err <= 2e-6 and same; bad += not ok
    print("blend_conv N=%d %d+%d->%d %dx%d slope=%s 2^%d: err %.1e bit-identical to two steps %s %s" % (Nb, Ca, Cb, Co, H, W, slope, kk, err, same, "ok" if ok else "FAIL"), flush=True)
    Ci = 16 * int(rng.integers(1, 9)); Co = int(rng.choice([32, 128, 136, 256])); H = 16 * int(rng.integers(1, 5)); W = 16 * int(rng.integers(1, 5)); Nb = int(rng.integers(1, 3))
    cv = Conv2d(Ci, Co, 3, 1, padding=1).cuda(); cv.precision = "f16x3"
    x = T(synth.hash_unitvar(seed + 13, (Nb, Ci, H, W), stream=1) * np.float32(2.0 ** kk))
    with torch.no_grad():
        cv.weight.copy_(T(synth.hash_unitvar(seed + 14, (Co, Ci, 3, 3), stream=2) / np.float32(np.sqrt(Ci * 9)))); cv.bias.copy_(T(synth.hash_unitvar(seed + 14, (Co,), stream=3) * np.float32(2.0 ** kk)))
    y = cv(x)
    r = torch.nn.functional.conv2d(x.double().cpu(), cv.weight.detach().double().cpu(), cv.bias.detach().double().cpu(), padding=1)
    err = (y.cpu().double() - r).abs().max().item() / r.abs().max().item(); ok = err <= 2e-6; bad += not ok
    print("conv3x3 (Winograd-eligible) N=%d %d->%d %dx%d 2^%d: err %.1e %s" % (Nb, Ci, Co, H, W, kk, err, "ok" if ok else "FAIL"), flush=True)
print("FUZZ: %d failures in %d cases" % (bad, 5 * ncase))
