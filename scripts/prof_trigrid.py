"""Time the ray kernel on tri-grids (triplane_depth 3, R=128, 48+48) next to the tri-plane configuration."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from real3dportrait_amd import ImportanceRenderer, OSGDecoder, RaySampler, synth
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
dn = synth.synth_decoder(7, sigma_bias=4.0)
dec = OSGDecoder().cuda()
with torch.no_grad():
    dec.net[0].weight.copy_(T(dn[0])); dec.net[0].bias.copy_(T(dn[1])); dec.net[2].weight.copy_(T(dn[2])); dec.net[2].bias.copy_(T(dn[3]))
cam = T(synth.look_at_camera(0.1, 0.0)[None])
o, d = RaySampler()(cam[:, :16].view(-1, 4, 4), cam[:, 16:].view(-1, 3, 3), 128)
opts = {"ray_start": "auto", "ray_end": "auto", "box_warp": 1.0, "depth_resolution": 48, "depth_resolution_importance": 48,
        "disparity_space_sampling": False, "clamp_mode": "softplus", "white_back": False}
for D in (1, 3):
    planes = torch.randn(1, 3, 32 * D, 256, 256, device="cuda")
    ren = ImportanceRenderer(hp={} if D == 1 else {"triplane_feature_type": "trigrid_v2", "triplane_depth": D}); ren.noise_mode = "hash"
    nhwc = ren.prepare_planes(planes)
    for _ in range(3): ren(nhwc, dec, o, d, opts)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): ren(nhwc, dec, o, d, opts)
    torch.cuda.synchronize(); print("depth %d: render R=128 48+48: %.3f ms" % (D, (time.perf_counter() - t) / 20 * 1e3))
