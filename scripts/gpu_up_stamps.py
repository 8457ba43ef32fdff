"""Per-block s_memtime stamps of upconv_fir_f16x3_kernel (experiment build -DR3D_ABLATE=512, selected with R3D_LIB):
where a block's time goes (first DMA wait, DMA waits + barriers of the main loop, main loop, epilogue) and how the blocks of
the launch spread over time.  The stamps of the LAST up-sampling launch of a forward survive (block1.conv0)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from real3dportrait_amd import SuperresolutionHybrid8XDC, synth, _lib
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
sr = SuperresolutionHybrid8XDC(channels=32, img_resolution=512, sr_num_fp16_res=0, sr_antialias=True).cuda()
with torch.no_grad():
    for blk, p in zip((sr.block0, sr.block1), synth.synth_sr_params(7)):
        for name in ("conv0", "conv1", "torgb"):
            l = getattr(blk, name); w, b, aw, ab = p[name]
            l.weight.copy_(T(w)); l.bias.copy_(T(b)); l.affine.weight.copy_(T(aw)); l.affine.bias.copy_(T(ab))
x = T(synth.hash_unitvar(7, (1, 32, 128, 128), stream=1)); rgb = x[:, :3].contiguous(); ws = torch.ones(1, 14, 512, device="cuda")
for _ in range(3): sr(rgb, x, ws, noise_mode="none")
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
nb = 1472
buf = (ctypes.c_ulonglong * (8 * nb))()
assert lib.r3d_debug_up_stamps(buf, nb) == 0
s = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 8).astype(np.int64)
live = s[:, 4] > 2000                       # padded blocks return at once (and never write: stale zeros or old stamps)
s = s[live]
t0 = s[:, 0] - s[:, 0].min()
print("blocks with stamps: %d" % len(s))
print("cycles (s_memtime ticks), median [p10 .. p90]:")
for name, col in (("first DMA wait", 1), ("all DMA waits + barriers of the main loop", 2), ("main loop", 3), ("whole block", 4)):
    v = s[:, col]; print("  %-44s %8.0f [%8.0f .. %8.0f]" % (name, np.median(v), np.percentile(v, 10), np.percentile(v, 90)))
epi = s[:, 4] - s[:, 3]
print("  %-44s %8.0f [%8.0f .. %8.0f]" % ("epilogue", np.median(epi), np.percentile(epi, 10), np.percentile(epi, 90)))
span = (s[:, 0] + s[:, 4]).max() - s[:, 0].min()
print("launch span %d ticks; block start times (ticks after the first), histogram over 10 bins of the span:" % span)
print("  ", np.histogram(t0, bins=10, range=(0, span))[0].tolist())
wall = s[:, 6]; print("wall clock span of the block ends: %.1f us (100 MHz counter)" % ((wall.max() - wall.min()) / 100.0))
hw = s[:, 5]; cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
print("distinct (se, sh, cu) seen: %d" % len(set(zip(se.tolist(), sh.tolist(), cu.tolist()))))
