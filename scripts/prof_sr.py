"""Run only SuperresolutionHybrid8XDC (128^2 -> 512^2) a few times -- target for rocprofv3 passes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from real3dportrait_amd import SuperresolutionHybrid8XDC, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
sr = SuperresolutionHybrid8XDC(channels=32, img_resolution=512, sr_num_fp16_res=0, sr_antialias=True).cuda()
with torch.no_grad():
    for blk, p in zip((sr.block0, sr.block1), synth.synth_sr_params(7)):
        for name in ("conv0", "conv1", "torgb"):
            l = getattr(blk, name); w, b, aw, ab = p[name]
            l.weight.copy_(T(w)); l.bias.copy_(T(b)); l.affine.weight.copy_(T(aw)); l.affine.bias.copy_(T(ab))
x = T(synth.hash_unitvar(7, (1, 32, 128, 128), stream=1)); rgb = x[:, :3].contiguous(); ws = torch.ones(1, 14, 512, device="cuda")
for _ in range(2): out = sr(rgb, x, ws, noise_mode="none")
import hashlib
print("digest", hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:16])
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(reps): sr(rgb, x, ws, noise_mode="none")
torch.cuda.synchronize(); total = (time.perf_counter() - t) / reps * 1e3
import ctypes
from real3dportrait_amd import _lib
lib = _lib.load(); lib.r3d_profile_configure(2 | 4); lib.r3d_profile_reset()
for _ in range(reps): sr(rgb, x, ws, noise_mode="none")
torch.cuda.synchronize(); ms, cnt = ctypes.c_double(0), ctypes.c_int(0); lib.r3d_profile_read(1, ctypes.byref(ms), ctypes.byref(cnt))
ums, ucnt = ctypes.c_double(0), ctypes.c_int(0); lib.r3d_profile_read(2, ctypes.byref(ums), ctypes.byref(ucnt))
print("SR 128->512: %.3f ms   conv kernels: %.3f ms/frame (%d launches)  upconv kernels: %.3f ms/frame  dbg=%s" % (total, ms.value / reps, cnt.value, ums.value / reps, os.environ.get("R3D_DBG", "0")))
