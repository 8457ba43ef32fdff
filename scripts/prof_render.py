"""Run only the fused ray kernel (REF: R=128, 48+48) a few times -- target for rocprofv3 --pmc passes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from real3dportrait_amd import ImportanceRenderer, OSGDecoder, RaySampler, synth
R = int(sys.argv[1]) if len(sys.argv) > 1 else 128
Nc = int(sys.argv[2]) if len(sys.argv) > 2 else 48
Nf = int(sys.argv[3]) if len(sys.argv) > 3 else 48
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
planes = T(synth.synth_planes(7, N=1)); dn = synth.synth_decoder(7, sigma_bias=4.0)
dec = OSGDecoder().cuda()
with torch.no_grad():
    dec.net[0].weight.copy_(T(dn[0])); dec.net[0].bias.copy_(T(dn[1])); dec.net[2].weight.copy_(T(dn[2])); dec.net[2].bias.copy_(T(dn[3]))
cam = T(synth.look_at_camera(0.1, 0.0)[None])
o, d = RaySampler()(cam[:, :16].view(-1, 4, 4), cam[:, 16:].view(-1, 3, 3), R)
ren = ImportanceRenderer(hp={}); ren.noise_mode = "hash"
opts = {"ray_start": "auto", "ray_end": "auto", "box_warp": 1.0, "depth_resolution": Nc, "depth_resolution_importance": Nf,
        "disparity_space_sampling": False, "clamp_mode": "softplus", "white_back": False}
for cm in ((True, False, True, False) if os.environ.get("R3D_PROF_LAYOUTS") else (True,)):
    ren.rgb_channel_major = cm
    for _ in range(2): ren(planes, dec, o, d, opts)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): ren(planes, dec, o, d, opts)
    torch.cuda.synchronize(); print("render R=%d %d+%d%s: %.3f ms" % (R, Nc, Nf, "" if cm else " (colours [N,M,32])", (time.perf_counter() - t) / reps * 1e3))
