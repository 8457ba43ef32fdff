"""How far apart are the uint8 frames of the three SR precisions on the bench scene?  (what switching the default to
'f16mx' does to the bytes that leave the pipeline.)  Output kept in profiles/r03/precision_frame_diff.txt."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from real3dportrait_amd.frames import ClipRenderer
dev = torch.device("cuda:0")
G, clip, dec, scene = bench.build_scene(torch, dev, n_frames=8)
frames = {}
for prec in ("f32", "f16x3", "f16mx"):
    G.superresolution.block0.precision = G.superresolution.block1.precision = prec
    frames[prec] = torch.stack([clip.render_u8(t).clone() for t in range(8)]).int()
    img = torch.cat([clip.render_image(t) for t in range(2)])
    frames[prec + "_img"] = img.clone()
n = frames["f32"].numel()
for a, b in (("f16x3", "f32"), ("f16mx", "f32"), ("f16mx", "f16x3")):
    d = (frames[a] - frames[b]).abs()
    e = (frames[a + "_img"] - frames[b + "_img"]).abs().max().item() / frames[b + "_img"].abs().max().item()
    print("%-6s vs %-6s: %.3f %% of %d bytes differ, max |diff| %d count(s); fp32 image: max err %.2e of max|ref|" % (a, b, 100.0 * float((d > 0).sum()) / n, n, int(d.max()), e))
