"""Per-phase cycle split of the fused up-sampling conv from an instrumented build (make EXTRA=-DR3D_STAMPS OUT=../lib/libr3d_hip_stamps.so
OBJDIR=../lib/obj_stamps; run with R3D_LIB=.../libr3d_hip_stamps.so): block0.conv0 (32 -> 256, 128^2 -> 256^2) and block1.conv0 (256 -> 128,
256^2 -> 512^2) separately.  Mean s_memtime cycles per WAVE: prologue (first DMAs issued), main loop, FIR epilogue, store drain."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from real3dportrait_amd import SynthesisBlock, synth, _lib
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 32)()
for (cin, cout, res, seed) in ((32, 256, 128, 100), (256, 128, 256, 200)):
    blk = SynthesisBlock(cin, cout, w_dim=512, resolution=2 * res, img_channels=3, is_last=False, conv_clamp=None).cuda()
    blk.precision = prec
    p = synth.synth_sr_block(7, cin, cout, 512, seed)
    with torch.no_grad():
        for name in ("conv0", "conv1", "torgb"):
            l = getattr(blk, name); w, b, aw, ab = p[name]
            l.weight.copy_(T(w)); l.bias.copy_(T(b)); l.affine.weight.copy_(T(aw)); l.affine.bias.copy_(T(ab))
    x = T(synth.hash_unitvar(7, (1, cin, res, res), stream=1)); img = x[:, :3].contiguous(); ws = torch.ones(1, 3, 512, device="cuda")
    for _ in range(2): blk(x, img, ws, noise_mode="none")
    torch.cuda.synchronize(); assert lib.r3d_debug_stamps_sr(buf) == 0
    reps = 5
    for _ in range(reps): blk(x, img, ws, noise_mode="none")
    torch.cuda.synchronize(); assert lib.r3d_debug_stamps_sr(buf) == 0
    waves = buf[31]
    names = ["prologue", "main loop: compute", "FIR epilogue", "store drain", "main loop: own DMA wait", "main loop: barrier"]
    tot = sum(buf[i] for i in range(6))
    print("up-conv %d -> %d at %d^2 (%s): %d waves per launch, %.0f cycles per wave" % (cin, cout, res, prec, waves // reps, tot / max(1, waves)))
    for i, n in enumerate(names):
        print("  %-24s %8.0f cycles  %5.1f %%" % (n, buf[i] / max(1, waves), 100.0 * buf[i] / max(1, tot)))
    cw = buf[30]; ctot = sum(buf[12 + i] for i in range(4))
    print("  conv1 %d -> %d at %d^2: %d waves per launch, %.0f cycles per wave" % (cout, cout, 2 * res, cw // reps, ctot / max(1, cw)))
    for i, n in enumerate(["prologue", "main loop", "epilogue", "store drain"]):
        print("    %-14s %8.0f cycles  %5.1f %%" % (n, buf[12 + i] / max(1, cw), 100.0 * buf[12 + i] / max(1, ctot)))
    # the shader clock the waves ran at: s_memtime cycles / s_memrealtime ticks x the wall-clock rate (100 MHz on gfx950)
    import ctypes as _c
    hip = _c.CDLL("libamdhip64.so"); wr = _c.c_int(0); hip.hipDeviceGetAttribute(_c.byref(wr), 10017, 0)   # hipDeviceAttributeWallClockRate (kHz)
    wr_khz = wr.value if wr.value > 0 else 100000
    for nm, sl in (("up-conv", 28), ("conv1", 26)):
        if buf[sl + 1]: print("  %s waves ran at %.3f GHz (s_memtime / s_memrealtime, wall clock %d kHz)" % (nm, buf[sl] / buf[sl + 1] * wr_khz / 1e6, wr_khz))
