"""Per-phase cycle split of the fused ray kernel from an instrumented build (make EXTRA=-DR3D_STAMPS OUT=../lib/libr3d_hip_stamps.so ...;
run with R3D_LIB=.../libr3d_hip_stamps.so).  Prints the mean s_memtime cycles per ray and phase (wave time; two waves share a SIMD)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from real3dportrait_amd import ImportanceRenderer, OSGDecoder, synth, _lib
R, Nc, Nf = (int(sys.argv[i]) if len(sys.argv) > i else v for i, v in ((1, 128), (2, 48), (3, 48)))
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
planes = T(synth.synth_planes(7, N=1) + synth.synth_planes(8, N=1, scale=0.1)); dn = synth.synth_decoder(7, sigma_bias=4.0)
dec = OSGDecoder().cuda()
with torch.no_grad():
    dec.net[0].weight.copy_(T(dn[0])); dec.net[0].bias.copy_(T(dn[1])); dec.net[2].weight.copy_(T(dn[2])); dec.net[2].bias.copy_(T(dn[3]))
cam = T(synth.camera_sweep(64, -0.4, 0.4)[5:6])
ren = ImportanceRenderer(hp={}); ren.noise_mode = "hash"; ren.need_depth = False
opts = {"ray_start": "auto", "ray_end": "auto", "box_warp": 1.0, "depth_resolution": Nc, "depth_resolution_importance": Nf,
        "disparity_space_sampling": False, "clamp_mode": "softplus", "white_back": False}
nhwc = ren.prepare_planes(planes)
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 32)()
for _ in range(3): ren.forward_camera(nhwc, dec, cam[:, :16].view(-1, 4, 4), cam[:, 16:].view(-1, 3, 3), R, opts)
torch.cuda.synchronize()
assert lib.r3d_debug_stamps(buf) == 0
reps = 10
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(reps): ren.forward_camera(nhwc, dec, cam[:, :16].view(-1, 4, 4), cam[:, 16:].view(-1, 3, 3), R, opts)
ev1.record(); torch.cuda.synchronize()
assert lib.r3d_debug_stamps(buf) == 0
rays = buf[31]
names = ["setup+depths", "coarse decode", "march+importance", "fine decode", "merge", "final march+omega", "composite+store", "between rays",
         "coarse gather", "fine gather"]
tot = sum(buf[i] for i in range(10))
print("R=%d %d+%d: %.3f ms per call (instrumented), %d rays, %.0f cycles per ray" % (R, Nc, Nf, ev0.elapsed_time(ev1) / reps, rays // reps, tot / rays))
for i, n in enumerate(names):
    print("  %-22s %8.0f cycles  %5.1f %%" % (n, buf[i] / rays, 100.0 * buf[i] / tot))
if buf[29]:
    print("  the waves ran at %.3f GHz (s_memtime cycles / s_memrealtime ticks x 100 MHz)" % (buf[28] / buf[29] * 0.1))
