"""A/B of ray-kernel builds on one box: for each library given (paths relative to the repo root; 'default' = the shipped one) a child process
renders the REF frame (R=128, 48+48, camera mode, hash noise, SPLIT output as ClipRenderer asks for it) and config-5-like shapes, prints the
kernel time (HIP events around 20 calls) and a digest of the outputs: equal digests = bit-identical renders."""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] != "--child":
    for lib in sys.argv[1:]:
        env = dict(os.environ)
        if lib != "default":
            env["R3D_LIB"] = os.path.join(ROOT, lib)
        else:
            env.pop("R3D_LIB", None)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
        print("%-52s %s" % (lib, out.stdout.strip().replace("\n", "\n" + " " * 53) or out.stderr[-400:]))
    sys.exit(0)
sys.path.insert(0, ROOT)
import ctypes
import numpy as np, torch
from real3dportrait_amd import ImportanceRenderer, OSGDecoder, synth, _lib
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
planes = T(synth.synth_planes(7, N=1) + synth.synth_planes(8, N=1, scale=0.1)); dn = synth.synth_decoder(7, sigma_bias=4.0)
dec = OSGDecoder().cuda()
with torch.no_grad():
    dec.net[0].weight.copy_(T(dn[0])); dec.net[0].bias.copy_(T(dn[1])); dec.net[2].weight.copy_(T(dn[2])); dec.net[2].bias.copy_(T(dn[3]))
lib = _lib.load()
res = []
for (R, Nc, Nf, N) in ((128, 48, 48, 1), (128, 48, 0, 1), (64, 96, 96, 2), (96, 32, 16, 1)):
    cam = T(synth.camera_sweep(64, -0.4, 0.4)[5:5 + N])
    ren = ImportanceRenderer(hp={}); ren.noise_mode = "hash"; ren.seed = 77
    opts = {"ray_start": "auto", "ray_end": "auto", "box_warp": 1.0, "depth_resolution": Nc, "depth_resolution_importance": Nf,
            "disparity_space_sampling": False, "clamp_mode": "softplus", "white_back": False}
    nhwc = ren.prepare_planes(planes.expand(N, -1, -1, -1, -1).contiguous())
    call = lambda: ren.forward_camera(nhwc, dec, cam[:, :16].view(-1, 4, 4), cam[:, 16:].view(-1, 3, 3), R, opts)
    for _ in range(3): out = call()
    torch.cuda.synchronize()
    h = hashlib.sha1()
    for t in out: h.update(t.contiguous().cpu().numpy().tobytes())
    lib.r3d_profile_configure(1); lib.r3d_profile_reset()
    for _ in range(20): call()
    torch.cuda.synchronize()
    ms, cnt = ctypes.c_double(0), ctypes.c_int(0); lib.r3d_profile_read(0, ctypes.byref(ms), ctypes.byref(cnt)); lib.r3d_profile_configure(0)
    res.append("R=%d %d+%d N=%d: %.4f ms %s" % (R, Nc, Nf, N, ms.value / max(1, cnt.value), h.hexdigest()[:10]))
print("\n".join(res))
