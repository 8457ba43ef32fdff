"""What one build of the library computes -- the head frame (both SR precisions), ray-kernel shapes, the torso frame -- so that two builds can be
compared.  `R3D_LIB` selects the build (default: the shipped library).  Used by tests/test_gpu_coresidency.py::test_build_without_packed_f32_agrees
with `make NOPK=1` (no packed-f32 instruction in the code objects) against the product build (packed-f32 forms rewritten by
csrc/tools/pk_opsel_fix.py).  The two compilations contract different mul + add pairs into fmas (fp-contract=fast around the SLP vectoriser), so the
comparison is a tight tolerance, not a digest: fp32 ray-kernel outputs within 1e-5, uint8 frames within one count in < 0.2 % of the bytes, the torso frame's fp32 image within 3e-4 of its maximum -- an op_sel mix-up
puts the wrong operand into a quarter of the lanes and is off by the operand's magnitude.
usage: gpu_build_ab.py OUT.npz         compute with the build R3D_LIB names, save the arrays
       gpu_build_ab.py A.so B.so       run both as children and compare"""
import hashlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2:
    import tempfile
    import numpy as np
    got = []
    for lib in sys.argv[1:3]:
        env = dict(os.environ)
        if lib == "default": env.pop("R3D_LIB", None)
        else: env["R3D_LIB"] = os.path.join(ROOT, lib)
        f = tempfile.NamedTemporaryFile(suffix=".npz", delete=False); f.close()
        r = subprocess.run([sys.executable, os.path.abspath(__file__), f.name], env=env, capture_output=True, text=True)
        print("== %s\n%s" % (lib, "\n".join(l for l in r.stdout.splitlines() if l.startswith(("digest ", "time "))) or r.stderr[-600:]))
        got.append(dict(np.load(f.name))); os.unlink(f.name)
    ok = len(got[0]) > 0 and set(got[0]) == set(got[1])
    for k in sorted(got[0]):
        a, b = got[0][k], got[1].get(k)
        if b is None or a.shape != b.shape: ok = False; continue
        if a.dtype == np.uint8:
            d = np.abs(a.astype(np.int32) - b.astype(np.int32)); frac = float((d > 0).mean())
            good = int(d.max()) <= 1 and frac < 2e-3
            print("compare %-28s uint8: %d of %d bytes differ (%.2e), max %d  %s" % (k, int((d > 0).sum()), d.size, frac, int(d.max()), "ok" if good else "BAD"))
        else:
            fin = np.isfinite(a) & np.isfinite(b)
            e = float(np.abs(a[fin] - b[fin]).max()) if fin.any() else 0.0
            # ray-kernel outputs: fp32 arithmetic, 1e-5 absolute; the torso frame's fp32 image went through nine f16mx convolutions whose fp8 roundings
            # flip with the last bit of their input: twice the tier's own error, 3e-4 of the largest value
            tol = 1e-5 if k.startswith("render_") else 3e-4 * float(np.abs(a[fin]).max())
            good = e <= tol and bool((np.isfinite(a) == np.isfinite(b)).all())
            print("compare %-28s fp32 max |diff| %.2e (max |value| %.3g)  %s" % (k, e, float(np.abs(a[fin]).max()), "ok" if good else "BAD"))
        ok = ok and good
    print("AGREE" if ok else "DISAGREE")
    sys.exit(0 if ok else 1)
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from real3dportrait_amd import ImportanceRenderer, OSGDecoder, TriPlaneGenerator, synth, _lib
dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
dig = lambda *ts: hashlib.sha1(b"".join(t.contiguous().cpu().numpy().tobytes() for t in ts)).hexdigest()[:16]
print("lib", _lib.LIB_PATH)
SAVE = {}
# 1. the head frame, 4 frames of the clip, both SR precisions
for prec in ("f16mx", "f16x3"):
    G, clip, dec, scene = bench.build_scene(torch, dev, n_frames=8, precision=prec)
    fr = [clip.render_u8(t).clone() for t in range(4)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(40): clip.render_u8(t % 8)
    torch.cuda.synchronize()
    print("digest head_frame %s %s" % (prec, dig(*fr)))
    SAVE["head_frame_" + prec] = torch.stack(fr).cpu().numpy()
    print("time head_frame %s one stream %.4f ms" % (prec, (time.perf_counter() - t0) / 40 * 1e3))
# 2. ray-kernel shapes (fp32 outputs)
planes = T(synth.synth_planes(7, N=1) + synth.synth_planes(8, N=1, scale=0.1)); dn = synth.synth_decoder(7, sigma_bias=4.0)
dec = OSGDecoder().cuda()
with torch.no_grad():
    dec.net[0].weight.copy_(T(dn[0])); dec.net[0].bias.copy_(T(dn[1])); dec.net[2].weight.copy_(T(dn[2])); dec.net[2].bias.copy_(T(dn[3]))
for (R, Nc, Nf, N) in ((128, 48, 48, 1), (128, 48, 0, 1), (64, 96, 96, 2), (96, 32, 16, 1)):
    cam = T(synth.camera_sweep(64, -0.4, 0.4)[5:5 + N])
    ren = ImportanceRenderer(hp={}); ren.noise_mode = "hash"; ren.seed = 77
    opts = {"ray_start": "auto", "ray_end": "auto", "box_warp": 1.0, "depth_resolution": Nc, "depth_resolution_importance": Nf,
            "disparity_space_sampling": False, "clamp_mode": "softplus", "white_back": False}
    nhwc = ren.prepare_planes(planes.expand(N, -1, -1, -1, -1).contiguous())
    out = ren.forward_camera(nhwc, dec, cam[:, :16].view(-1, 4, 4), cam[:, 16:].view(-1, 3, 3), R, opts)
    torch.cuda.synchronize()
    print("digest render R=%d %d+%d N=%d %s" % (R, Nc, Nf, N, dig(*out)))
    for nm, t in zip(("rgb", "depth", "wsum"), out[:3]): SAVE["render_R%d_%d+%d_%s" % (R, Nc, Nf, nm)] = t.float().cpu().numpy()
# 3. the torso frame (fused SuperresolutionHybrid8XDC_Warp forward + to_plane_cnn), f16mx
torch.manual_seed(1234)                          # the generator shell and to_plane_cnn are torch-initialised: same values in both children
G = TriPlaneGenerator().to(dev).eval()
frame, _ = bench.build_torso_frame(torch, dev, G, precision="f16mx")
fr = [frame(t).clone() for t in range(2)]
torch.cuda.synchronize()
print("digest torso_frame f16mx %s" % dig(*fr))
SAVE["torso_frame_f16mx"] = torch.stack(fr).cpu().numpy()
if len(sys.argv) > 1: np.savez(sys.argv[1], **SAVE)
