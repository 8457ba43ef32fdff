import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_golden
from test_gpu_parity import _fusion_modules, T
from test_oracle_golden import _fusion_inputs
from real3dportrait_amd import synth
g = load_golden("fusion_a"); seed, R = int(g["seed"]), int(g["R"])
i = {k: T(torch, v) for k, v in _fusion_inputs(seed, R).items()}
stacks, blk = _fusion_modules(torch, seed, R)
def err(got, key):
    ref = g[key]; print(key, "err", float(np.abs(got.cpu().numpy() - ref).max()), "max", float(np.abs(ref).max()))
x_torso = stacks["torso_encoder"](i["hid"]); err(x_torso[:, ::4], "x_torso")
x_bg = stacks["bg_encoder"](i["bg"]); err(x_bg[:, ::4], "x_bg")
# layer by layer against torch
st = stacks["bg_encoder"]; r = i["bg"].double().cpu()
convs = [m for m in st if hasattr(m, "weight")]
for k, m in enumerate(convs):
    r = torch.nn.functional.conv2d(r, m.weight.detach().double().cpu(), m.bias.detach().double().cpu(), padding=m.padding[0])
    if k < 2: r = torch.nn.functional.leaky_relu(r, 0.01)
    sc = m._scales.view(torch.float32)
    print("layer", k, "ref max", float(r.abs().max()), "scales meta", sc[-8:].tolist(), "in_vec0", float(sc[0]))
y0 = convs[0](i["bg"], negative_slope=0.01)
r0 = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(i["bg"].double().cpu(), convs[0].weight.detach().double().cpu(), convs[0].bias.detach().double().cpu(), padding=1), 0.01)
print("conv0 alone err", float((y0.cpu().double() - r0).abs().max()))
from real3dportrait_amd.superresolution import blend_cat
a, occ = i["alpha"], i["occ"]
rgb1 = i["rgb"] * a + i["rgb_torso"] * (1 - a)
cat1 = torch.cat([i["x_head"] * a, x_torso * (1 - a)], dim=1)
x1 = stacks["fuse_head_torso_convs"](cat1); err(x1[:, ::4], "x1")
st = stacks["fuse_head_torso_convs"]; convs = [m for m in st if hasattr(m, "weight")]
for k, m in enumerate(convs):
    sc = m._scales.view(torch.float32); print(" fuse_ht layer", k, "meta", sc[-4:].tolist(), "in_vec0", float(sc[0]), "cat max", float(cat1.abs().max()))
r = torch.nn.functional.conv2d(cat1.double().cpu(), convs[0].weight.detach().double().cpu(), convs[0].bias.detach().double().cpu(), padding=1)
y0 = convs[0](cat1, negative_slope=None); print("fuse_ht conv0 alone err", float((y0.cpu().double() - r).abs().max()), "refmax", float(r.abs().max()))
y0h = convs[0](cat1[:, :256].contiguous() if False else cat1, negative_slope=0.01)
x2, rgb2 = blk(x1, rgb1, i["ws"], noise_mode="none"); err(x2[:, ::4], "x2"); err(rgb2, "rgb2")
x3 = stacks["fuse_fg_bg_convs"](torch.cat([x2 * occ, x_bg * (1 - occ)], dim=1)); err(x3[:, ::4], "x3")
y1 = stacks["fuse_head_torso_convs"](blend_cat(i["x_head"], x_torso, a, stacks["fuse_head_torso_convs"]))
print("blend path y1 vs x1", float((y1 - x1).abs().max()))
