"""Stress: run the SR module on several streams concurrently (each with its own module shell sharing parameters) and compare
with the single-stream result; reports which intermediate first differs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from real3dportrait_amd import SuperresolutionHybrid8XDC, synth
torch.manual_seed(0)
base = SuperresolutionHybrid8XDC(32, 512, 0, True).cuda()
def shell():
    sr = SuperresolutionHybrid8XDC(32, 512, 0, True).cuda()
    for name in ("block0", "block1"):
        src, dst = getattr(base, name), getattr(sr, name)
        dst.conv0, dst.conv1, dst.torgb = src.conv0, src.conv1, src.torgb
    return sr
x = torch.randn(1, 32, 128, 128, device="cuda"); rgb = torch.randn(1, 3, 128, 128, device="cuda") * 0.3; ws = torch.ones(1, 14, 512, device="cuda")
ref = base(rgb, x, ws, noise_mode="none").clone(); torch.cuda.synchronize()
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
shells = [shell() for _ in range(NS)]; streams = [torch.cuda.Stream() for _ in range(NS)]
for s in shells: s(rgb, x, ws, noise_mode="none")
torch.cuda.synchronize()
bad = 0; worst = 0.0; npx = 0
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for it in range(ITERS):
    outs = []
    for s, st in zip(shells, streams):
        with torch.cuda.stream(st):
            outs.append(s(rgb, x, ws, noise_mode="none"))
    torch.cuda.synchronize()
    for o in outs:
        if not torch.equal(o, ref):
            bad += 1; d = (o - ref).abs(); worst = max(worst, float(d.max())); npx += int((d > 0).sum())
            if bad <= 3:
                idx = torch.nonzero(d[0] > 0)
                import collections
                tiles = collections.Counter((int(c), int(y) // 16, int(xx) // 16) for c, y, xx in idx.tolist())
                print("  diff #%d: %d values; (channel, tile_y, tile_x) -> count:" % (bad, len(idx)), sorted(tiles.items())[:12])
                rows = collections.Counter((int(c), int(y)) for c, y, xx in idx.tolist()); print("   rows per (c,y):", sorted(rows.items())[:10])
print("SR concurrent on %d streams: %d / %d outputs differ, worst abs diff %.4g, differing values %d" % (NS, bad, ITERS * NS, worst, npx))
