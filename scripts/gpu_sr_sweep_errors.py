"""Error table behind tests/test_gpu_range_and_sizes.py::test_sr_block_range_sweep: max relative error of SynthesisBlock (up) and
SynthesisBlockNoUp against torch fp64 with one operand class scaled by 2^k.  Output kept in profiles/r03/sr_sweep_errors.txt."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import test_gpu_range_and_sizes as tr
from real3dportrait_amd import synth
from real3dportrait_amd.superresolution import SynthesisBlock, SynthesisBlockNoUp


def run(what, k, up):
    N, Cin, Cout, H, W = 2, 32, 128, 18, 14
    sc = np.float32(2.0 ** k)
    p = {kk: tuple(np.array(a) for a in v) for kk, v in synth.synth_sr_block(71, Cin, Cout, 512, 700).items()}
    if what == "weights":
        for layer in ("conv0", "conv1"):
            p[layer] = (p[layer][0] * sc,) + p[layer][1:]
    elif what == "styles":
        for layer in ("conv0", "conv1", "torgb"):
            w_, b_, aw, ab = p[layer]; p[layer] = (w_, b_, aw * sc, ab * sc)
    elif what == "bias":
        for layer in ("conv0", "conv1"):
            w_, b_, aw, ab = p[layer]; p[layer] = (w_, b_ * sc, aw, ab)
    blk = (SynthesisBlock if up else SynthesisBlockNoUp)(Cin, Cout, w_dim=512, resolution=2 * H if up else H, img_channels=3, is_last=False, conv_clamp=None).cuda()
    tr.load_block(torch, blk, p)
    x = synth.hash_unitvar(72, (N, Cin, H, W), stream=1) * (sc if what == "input" else np.float32(1.0))
    img = synth.hash_unitvar(72, (N, 3, H, W), stream=2) * np.float32(0.5)
    ws = np.ones((N, 3, 512), np.float32) + synth.hash_unitvar(72, (N, 3, 512), stream=3) * np.float32(0.2)
    xo, io = blk(tr.T(torch, x), tr.T(torch, img), tr.T(torch, ws), noise_mode="none")
    rx, ri = tr._block_fp64(torch, p, torch.from_numpy(x), torch.from_numpy(img), torch.from_numpy(ws), up, None)
    ex = (xo.cpu().double() - rx).abs().max().item() / max(rx.abs().max().item(), 1e-300)
    ei = (io.cpu().double() - ri).abs().max().item() / max(ri.abs().max().item(), 1e-300)
    return ex, ei


worst = 0.0
print("%-8s %4s | %-22s | %-22s" % ("operand", "k", "SynthesisBlockNoUp x / img", "SynthesisBlock x / img"))
for what in ("input", "weights", "styles", "bias"):
    for k in tr.SWEEP:
        a, b = run(what, k, False), run(what, k, True)
        worst = max(worst, *a, *b)
        print("%-8s %4d | %.2e  %.2e     | %.2e  %.2e" % (what, k, a[0], a[1], b[0], b[1]))
print("worst %.2e (test tolerance: see tests/test_gpu_range_and_sizes.py)" % worst)
