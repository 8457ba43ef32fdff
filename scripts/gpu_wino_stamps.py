"""Per-phase cycle split of the Winograd F(2,3) conv from an instrumented build (make EXTRA=-DR3D_STAMPS OUT=../lib/libr3d_hip_stamps.so
OBJDIR=../lib/obj_stamps; run with R3D_LIB=.../libr3d_hip_stamps.so R3D_CONV_WINO=1): mean s_memtime cycles per WAVE of block0.conv1 / block1.conv1."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from real3dportrait_amd import SynthesisBlock, synth, _lib
prec = sys.argv[1] if len(sys.argv) > 1 else "f16mx"
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
    cw = buf[30]; names = ["prologue", "matrix phase", "transform phase", "drain (vmcnt/lgkmcnt) before barrier", "barrier", "exchange", "epilogue", "store drain"]
    ctot = sum(buf[12 + i] for i in range(8))
    print("conv1 %d -> %d at %d^2 (%s, wino=%s): %d waves per launch, %.0f cycles per wave" % (cout, cout, 2 * res, prec, os.environ.get("R3D_CONV_WINO"), cw // reps, ctot / max(1, cw)))
    for i, n in enumerate(names):
        print("    %-38s %8.0f cycles  %5.1f %%" % (n, buf[12 + i] / max(1, cw), 100.0 * buf[12 + i] / max(1, ctot)))
    hip = ctypes.CDLL("libamdhip64.so"); wr = ctypes.c_int(0); hip.hipDeviceGetAttribute(ctypes.byref(wr), 10017, 0)
    wr_khz = wr.value if wr.value > 0 else 100000
    if buf[27]: print("    waves ran at %.3f GHz" % (buf[26] / buf[27] * wr_khz / 1e6))
