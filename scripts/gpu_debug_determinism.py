"""Is the ray kernel's output independent of what else is resident on the CUs?  Sequential frames (reference) vs the same frames issued
on three streams with SR kernels of the other streams co-resident; counts rays that differ.  (R3D_LIB selects the library.)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from real3dportrait_amd import TriPlaneGenerator, synth
from real3dportrait_amd.frames import ClipRenderer, clone_generator_shell
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
with_sr = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
G = TriPlaneGenerator().cuda().eval()
dec = synth.synth_decoder(3, sigma_bias=4.0)
with torch.no_grad():
    G.decoder.net[0].weight.copy_(T(dec[0])); G.decoder.net[0].bias.copy_(T(dec[1])); G.decoder.net[2].weight.copy_(T(dec[2])); G.decoder.net[2].bias.copy_(T(dec[3]))
cano = T(synth.synth_planes(3, N=1)); res = [T(synth.synth_planes(4 + i, N=1, scale=0.1)) for i in range(2)]
cams = T(synth.camera_sweep(6, -0.3, 0.3)); ws = torch.ones(1, 14, 512, device="cuda")
single = ClipRenderer(G, cano, res, cams, ws, base_seed=11)
a = [single._features(t).clone() for t in range(6)]
b = [single._features(t).clone() for t in range(6)]
print("same shell twice, sequential: max diff", max(float((x - y).abs().max()) for x, y in zip(a, b)))
shells = [single] + [ClipRenderer(clone_generator_shell(G), cano, res, cams, ws, base_seed=11) for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(3)]
sr_in = torch.randn(1, 32, 128, 128, device="cuda")
bad_frames = bad_rays = 0
for rep in range(reps):
    out = [None] * 6
    for t in range(6):
        with torch.cuda.stream(streams[t % 3]):
            out[t] = shells[t % 3]._features(t).clone()
            if with_sr: shells[t % 3].G.superresolution(sr_in[:, :3], sr_in, ws, noise_mode="none")
    torch.cuda.synchronize()
    for t in range(6):
        dmap = (a[t] - out[t]).abs().amax(dim=1)[0]
        n = int((dmap > 0).sum())
        if n:
            bad_frames += 1; bad_rays += n
            if bad_frames <= 6:
                rc = [(int(r), int(c)) for r, c in torch.nonzero(dmap > 0).tolist()]
                print("  rep %d frame %d: %d rays differ %s max %.2e" % (rep, t, n, rc[:12], float(dmap.max())))
print("lib %s  with_sr %s: %d of %d frames differ, %d rays in total" % (os.environ.get("R3D_LIB", "default")[-24:], with_sr, bad_frames, 6 * reps, bad_rays))
if "bis256" in os.environ.get("R3D_LIB", "") or "bis2048" in os.environ.get("R3D_LIB", ""):
    import ctypes
    from real3dportrait_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH); buf = (ctypes.c_int * (16 + 16 * 32))(); lib.r3d_debug_read(buf)
    print("tap mismatches recorded:", buf[0])
    import struct
    f = lambda i: struct.unpack("f", struct.pack("i", i))[0]
    for sl in range(min(buf[0], 30)):
        r = buf[16 + 16 * sl: 32 + 16 * sl]
        print("  plane %d tap %d tid %3d (q %d) block %4d: idx dpp %9d own %9d | w dpp %.6f own %.6f | exec %08x%08x | src-lane value idx %9d w %.6f | pos %.4f %.4f %.4f" % (
            r[0], r[1], r[2], r[12], r[11], r[3], r[4], f(r[5]), f(r[6]), r[8] & 0xffffffff, r[7] & 0xffffffff, r[9], f(r[10]), f(r[13]), f(r[14]), f(r[15])))
