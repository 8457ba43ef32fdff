"""GPU: the C ABI driven WITHOUT the package -- the ctypes stub printed in INTEGRATION.md section 2 is extracted from the document and
executed verbatim (so the document cannot drift from the library), then checked against the reference goldens / the oracle.
The other GPU tests reach the same entry points through real3dportrait_amd's operators."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def stub():
    import torch
    assert torch.cuda.is_available()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"```python\n(# modules/eg3ds/torch_utils/r3d_hip\.py.*?)```", doc, re.S)
    assert m, "INTEGRATION.md lost its ctypes stub"
    ns = {"R3D_LIB_PATH": os.path.join(ROOT, "real3dportrait_amd", "lib", "libr3d_hip.so")}
    exec(compile(m.group(1), "INTEGRATION.md:stub", "exec"), ns)
    return ns


class _Dec:     # anything with .net[0] / .net[2] .weight / .bias, like the reference's OSGDecoder
    def __init__(self, torch, g):
        L = lambda w, b: type("L", (), {"weight": torch.from_numpy(w).cuda(), "bias": torch.from_numpy(b).cuda()})()
        self.net = [L(g["dec_w1"], g["dec_b1"]), None, L(g["dec_w2"], g["dec_b2"])]


def test_raw_stub_render_matches_reference_golden(stub):
    import torch
    g = load_golden("render_b_n2_r16_48p48")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    cams = T(g["cams"])
    o, d = stub["raygen"](cams[:, :16].reshape(-1, 4, 4).contiguous(), cams[:, 16:].reshape(-1, 3, 3).contiguous(), int(g["R"]))
    assert np.abs(o.cpu().numpy() - g["origins"]).max() <= 1e-6 and np.abs(d.cpu().numpy() - g["dirs"]).max() <= 2e-6
    nhwc, part = stub["planes_to_nhwc"](T(g["planes"]))
    assert abs(float(part.max()) - float(np.abs(g["planes"]).max())) == 0.0
    N, M, Nc, Nf = g["cams"].shape[0], int(g["R"]) ** 2, int(g["Nc"]), int(g["Nf"])
    for pm in (part, None):                                       # with the layout pass's partials, and measured inside the call
        rgb, depth, wsum, valid = stub["render"](nhwc, pm, _Dec(torch, g), o, d, Nc, Nf, 1.0, T(g["noise_c"]).reshape(N, M, Nc), T(g["u_f"]))
        torch.cuda.synchronize()
        assert np.array_equal(valid.cpu().numpy(), g["valid"])
        assert np.abs(rgb.cpu().numpy() - g["rgb"]).max() <= 2e-4 and np.abs(wsum.cpu().numpy() - g["wsum"]).max() <= 2e-4
        assert np.abs(depth.cpu().numpy() - g["depth"]).max() <= 1e-4


def test_raw_stub_sr_block_matches_reference_golden(stub):
    """sr_small_a: the reference's SynthesisBlock pair (32 -> 256 @ 32^2, 256 -> 128 @ 64^2) with non-trivial ws, both blocks through the
    raw stub (x0 / x1 are stored for every 4th / 8th channel)."""
    import torch
    from real3dportrait_amd import synth
    g = load_golden("sr_small_a")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    params = synth.synth_sr_params(int(g["seed"]))
    L = lambda w, b, aw, ab: type("L", (), {"weight": T(w), "bias": T(b), "affine": type("A", (), {"weight": T(aw), "bias": T(ab)})()})()
    blocks = [type("B", (), {"conv0": L(*p["conv0"]), "conv1": L(*p["conv1"]), "torgb": L(*p["torgb"])})() for p in params]
    ws = T(g["ws"])
    x0, rgb0 = stub["sr_block"](blocks[0], T(g["x"]), T(g["rgb"]), ws)
    x1, rgb1 = stub["sr_block"](blocks[1], x0, rgb0, ws)
    torch.cuda.synchronize()
    for got, ref, name in ((x0[:, ::4], g["x0"], "x0"), (rgb0, g["rgb0"], "rgb0"), (x1[:, ::8], g["x1"], "x1"), (rgb1, g["rgb1"], "rgb1")):
        e, tol = np.abs(got.cpu().numpy() - ref).max(), 2e-4 * max(1.0, float(np.abs(ref).max()))
        print("raw stub %s: max err %.2e (tol %.1e)" % (name, e, tol))
        assert e <= tol
