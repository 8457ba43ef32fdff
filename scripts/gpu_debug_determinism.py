"""Is the ray kernel's output independent of what else is resident on the CUs?  Sequential frames (reference) vs the same frames issued
on three streams with SR kernels of the other streams co-resident; counts rays that differ.  (R3D_LIB selects the library.)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from real3dportrait_amd import TriPlaneGenerator, synth
from real3dportrait_amd.frames import ClipRenderer, clone_generator_shell
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
with_sr = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
other = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] not in ("0", "1") else None      # co-resident load instead of the SR: gemm | copy | f32sr | agg:<mode>[:iters[:grid]]
agg = None
if other and other.startswith("agg:"):      # synthetic load assembled from feature bits (scripts/probes/aggressor.hip)
    import ctypes
    f = other.split(":")
    agg_mode, agg_iters, agg_grid = int(f[1]), int(f[2]) if len(f) > 2 else 150, int(f[3]) if len(f) > 3 else 1024
    agg = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes", "bin", "libaggressor.so"))
    agg.agg_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
G = TriPlaneGenerator().cuda().eval()
dec = synth.synth_decoder(3, sigma_bias=4.0)
with torch.no_grad():
    G.decoder.net[0].weight.copy_(T(dec[0])); G.decoder.net[0].bias.copy_(T(dec[1])); G.decoder.net[2].weight.copy_(T(dec[2])); G.decoder.net[2].bias.copy_(T(dec[3]))
cano = T(synth.synth_planes(3, N=1)); res = [T(synth.synth_planes(4 + i, N=1, scale=0.1)) for i in range(2)]
cams = T(synth.camera_sweep(6, -0.3, 0.3)); ws = torch.ones(1, 14, 512, device="cuda")
def watch(c):                       # keep the weight sums of the last render (experiment builds flag rays in them)
    c.G.renderer.register_forward_hook(lambda m, inp, out: setattr(c, "_last_wsum", out[2]))
    return c
single = watch(ClipRenderer(G, cano, res, cams, ws, base_seed=11))
a = [single._features(t).clone() for t in range(6)]
aw = []
for t in range(6):
    single._features(t); aw.append(single._last_wsum.clone())
b = [single._features(t).clone() for t in range(6)]
print("same shell twice, sequential: max diff", max(float((x - y).abs().max()) for x, y in zip(a, b)))
shells = [single] + [watch(ClipRenderer(clone_generator_shell(G), cano, res, cams, ws, base_seed=11)) for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(3)]
sr_in = torch.randn(1, 32, 128, 128, device="cuda")
gm_a = torch.randn(4096, 4096, device="cuda", dtype=torch.float16); gm_b = torch.randn(4096, 4096, device="cuda", dtype=torch.float16); gm_c = [None] * 3
cp_a = torch.randn(64 << 20, device="cuda"); cp_b = [torch.empty_like(cp_a) for _ in range(3)]
if agg is not None:
    agg_src = torch.randn(1 << 18, 8, device="cuda").mul_(0.01).to(torch.float16).contiguous()                               # 2^18 uint4
    agg_out = [torch.empty(agg_grid * 512, device="cuda") for _ in range(3)]
if other == "f32sr":
    for sh in shells:
        sh.G.superresolution.block0.precision = "f32"; sh.G.superresolution.block1.precision = "f32"
bad_frames = bad_rays = flag_rays = wsum_bad = 0
for rep in range(reps):
    out = [None] * 6; flagged = [None] * 6
    for t in range(6):
        with torch.cuda.stream(streams[t % 3]):
            out[t] = shells[t % 3]._features(t).clone()
            flagged[t] = shells[t % 3]._last_wsum
            if other == "gemm":
                for _ in range(3): gm_c[t % 3] = gm_a @ gm_b                       # rocBLAS / hipBLASLt MFMA kernels
            elif other == "copy":
                for _ in range(6): cp_b[t % 3].copy_(cp_a)                         # plain HBM traffic
            elif other == "f32sr":
                shells[t % 3].G.superresolution(sr_in[:, :3], sr_in, ws, noise_mode="none")
            elif agg is not None:
                for _ in range(3):
                    rc = agg.agg_launch(agg_mode, agg_src.data_ptr(), 1 << 18, agg_out[t % 3].data_ptr(), agg_grid, agg_iters, torch.cuda.current_stream().cuda_stream)
                    assert rc == 0, rc
            elif with_sr: shells[t % 3].G.superresolution(sr_in[:, :3], sr_in, ws, noise_mode="none")
    torch.cuda.synchronize()
    for t in range(6):
        dmap = (a[t] - out[t]).abs().amax(dim=1)[0]
        n = int((dmap > 0).sum())
        nflag = int((flagged[t] > 500).sum()); flag_rays += nflag
        if nflag and flag_rays <= 40:
            fr = torch.nonzero(flagged[t].flatten() > 500).flatten().tolist()
            print("  rep %d frame %d: in-kernel tap mismatch flagged on rays %s (lanes flagged: %s)" % (rep, t, [(r // 128, r % 128) for r in fr[:8]], [int(flagged[t].flatten()[r] // 1000) for r in fr[:8]]))
        if n:
            dw = (aw[t] - flagged[t]).abs().flatten()
            wsum_bad += int((dw > 0).sum())
            if bad_frames < 3:
                rr = torch.nonzero(dmap.flatten() > 0).flatten().tolist()[:4]
                for r in rr:
                    dch = (a[t] - out[t]).abs()[0, :, r // 128, r % 128]
                    print("    ray (%d,%d): wsum seq %.6f conc %.6f | per-channel diff: %s" % (r // 128, r % 128, float(aw[t].flatten()[r]), float(flagged[t].flatten()[r]), " ".join("%.0e" % float(x) for x in dch.tolist())))
            bad_frames += 1; bad_rays += n
            if bad_frames <= 6:
                rc = [(int(r), int(c)) for r, c in torch.nonzero(dmap > 0).tolist()]
                print("  rep %d frame %d: %d rays differ %s max %.2e" % (rep, t, n, rc[:12], float(dmap.max())))
print("rays with an in-kernel shared-vs-own tap mismatch flag: %d; differing rays whose weight sum also differs: %d" % (flag_rays, wsum_bad))
print("lib %s  co-resident load %s: %d of %d frames differ, %d rays in total" % (os.environ.get("R3D_LIB", "default")[-24:], other or ("f16x3 SR" if with_sr else "none"), bad_frames, 6 * reps, bad_rays))
if "ablate12288" in os.environ.get("R3D_LIB", ""):
    import ctypes
    from real3dportrait_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH); buf = (ctypes.c_int * (64 + 64 * 24))(); lib.r3d_debug_read(buf)
    print("flagged rays recorded:", buf[0], " (per lane: plane bits 0-2, bit 3 = position differs inside the quad, bits 4-6 coarse tile, 7-9 fine tile)")
    for sl in range(min(buf[0], 24)):
        w = [buf[64 + 64 * sl + l] for l in range(64)]
        ray = max(x >> 8 for x in w)
        print("  ray (%d,%d): " % (ray // 128, ray % 128) + " ".join("%d:%03x" % (l, w[l] & 0xff | ((w[l] & 0x300))) for l in range(64) if w[l] & 0x3ff))
