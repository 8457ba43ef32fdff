"""GPU: results must not depend on what else is resident on the CUs.

Round 3 found why round 2's experiment kernels sometimes did: on gfx950 a packed-f32 instruction whose src1 / src2 op_sel bit is set
returns a wrong low half in lanes 48-63 while another wave of the SIMD executes MFMAs.  The build rewrites those forms
(csrc/tools/pk_opsel_fix.py; tests/test_abi.py checks the shipped code objects).  These tests keep the hardware facts the rewrite relies
on, and the end-to-end guarantees, honest:
  * the stand-alone probe (scripts/probes/coexec_probe.hip): every pattern is exact when it runs alone, and the forms the rewriter EMITS
    stay exact next to an MFMA load;
  * the ray kernel (which now shares bilinear taps inside lane quads -- the variant that exposed the erratum) next to a synthetic MFMA
    load that made the unfixed variant differ in more than half of its frames;
  * the soak of scripts/gpu_soak_pipeline.py reduced to test size, for both SR precisions that run MFMAs next to packed-f32 epilogues.
"""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import ROOT
from test_gpu_parity import T, load_block

pytestmark = pytest.mark.gpu
BUILD = os.path.join(ROOT, "tests", "_build")


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch


def _helpers():
    import __graft_entry__ as g
    g.build_test_helpers()


def _probe(select):
    _helpers()
    out = subprocess.run([os.path.join(BUILD, "coexec_probe"), "20000", "29", str(select)], capture_output=True, text=True, timeout=300).stdout
    res, phase = {}, None
    for line in out.splitlines():
        if line.startswith("== victims"):
            phase = "load" if "next to" in line else "alone"
        m = re.match(r"\s+\[(\d+)\] (.*?)\s+mismatches\s+(\d+)", line)
        if m and phase:
            res.setdefault(int(m.group(1)), {"name": m.group(2)})[phase] = int(m.group(3))
    return res


def test_probe_rewritten_forms_are_exact_next_to_mfma_load():
    res = {**_probe(94), **_probe(93)}
    assert set(range(39, 51)) <= set(res), sorted(res)
    for k, r in sorted(res.items()):
        print("  pattern %d %-62s alone %d, next to MFMA load %d" % (k, r["name"], r["alone"], r["load"]))
    for k in (39, 40, 41, 42, 43, 44, 45, 46, 48, 49, 50):
        assert res[k]["alone"] == 0, (k, res[k])                    # every form is exact when nothing else runs
    for k in (43, 44, 45, 49, 50):                                   # src0-crossed / broadcast / swapped-SGPR forms: what pk_opsel_fix.py emits or leaves
        assert res[k]["load"] == 0, "form relied on by the build is not exact next to MFMAs: %r" % (res[k],)
    if res[39]["load"] == 0 and res[48]["load"] == 0:
        print("  NOTE: the erratum (patterns 39, 48) did not show on this device / firmware")


def test_stand_alone_erratum_reproducer():
    """scripts/probes/pk_opsel_erratum_repro.hip (84 lines, no library; expected output profiles/r04/pk_opsel_erratum_repro.txt): the form
    the build emits (src0 crossed) is exact alone and next to the conv-shaped MFMA load -- its exit code; the hazardous form (src1
    crossed) is exact alone.  That the hazardous form DOES miscompute next to the load is printed, not asserted: a fixed part would be
    good news, not a test failure."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "pk_opsel_erratum_repro")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    rows = [l for l in out.stdout.splitlines() if l.startswith("form ")]
    assert len(rows) == 4 and all(int(r.split(":")[2].split("mismatches")[0]) == 0 for r in (rows[0], rows[2], rows[3])), rows


def test_probe_other_patterns_are_exact():
    """The instruction patterns that were suspected and cleared on the way (DPP after VALU, SGPR mask RAW / WAR / WAW, DPP scans, VMEM address
    reuse) stay exact next to the MFMA load."""
    res = _probe(-1)
    for k in (0, 1, 2, 4, 5, 6, 8, 9, 10, 11, 22, 23, 24, 26, 27, 28, 29):
        assert res[k]["alone"] == 0 and res[k]["load"] == 0, (k, res[k])


def _ray_scene(torch):
    from real3dportrait_amd import TriPlaneGenerator, synth
    from real3dportrait_amd.frames import ClipRenderer, clone_generator_shell
    G = TriPlaneGenerator().cuda().eval()
    dec = synth.synth_decoder(3, sigma_bias=4.0)
    with torch.no_grad():
        G.decoder.net[0].weight.copy_(T(torch, dec[0])); G.decoder.net[0].bias.copy_(T(torch, dec[1]))
        G.decoder.net[2].weight.copy_(T(torch, dec[2])); G.decoder.net[2].bias.copy_(T(torch, dec[3]))
    cano = T(torch, synth.synth_planes(3, N=1)); res = [T(torch, synth.synth_planes(4 + i, N=1, scale=0.1)) for i in range(2)]
    cams = T(torch, synth.camera_sweep(6, -0.3, 0.3)); ws = torch.ones(1, 14, 512, device="cuda")
    shells = [ClipRenderer(G, cano, res, cams, ws, base_seed=11)]
    shells += [ClipRenderer(clone_generator_shell(G), cano, res, cams, ws, base_seed=11) for _ in range(2)]
    return shells


@pytest.mark.parametrize("mode", [29, 31])
def test_ray_kernel_next_to_synthetic_mfma_load(torch_cuda, mode):
    """120 frames on three streams, each followed by three launches of the synthetic load (f16 MFMAs + LDS reads + barriers [+ LDS-DMA],
    76 KB LDS, 512 threads: the SR conv's shape).  The round-2 experiment build of this very gather differed in 320 of 600 frames."""
    torch = torch_cuda
    _helpers()
    agg = ctypes.CDLL(os.path.join(BUILD, "libaggressor.so"))
    agg.agg_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    shells = _ray_scene(torch)
    ref = [shells[0]._features(t).clone() for t in range(6)]
    streams = [torch.cuda.Stream() for _ in range(3)]
    src = torch.randn(1 << 18, 8, device="cuda").mul_(0.01).to(torch.float16).contiguous()
    outb = [torch.empty(1024 * 512, device="cuda") for _ in range(3)]
    torch.cuda.synchronize()
    bad = 0
    for _ in range(20):
        out = [None] * 6
        for t in range(6):
            with torch.cuda.stream(streams[t % 3]):
                out[t] = shells[t % 3]._features(t).clone()
                for _k in range(3):
                    assert agg.agg_launch(mode, src.data_ptr(), 1 << 18, outb[t % 3].data_ptr(), 1024, 150, torch.cuda.current_stream().cuda_stream) == 0
        torch.cuda.synchronize()
        bad += sum(int(not torch.equal(a, b)) for a, b in zip(ref, out))
    assert bad == 0, "%d of 120 frames differ from the sequential render" % bad


@pytest.mark.parametrize("precision", ["f16x3", "f16mx"])
def test_pipelined_frames_equal_sequential_soak(torch_cuda, precision):
    """scripts/gpu_soak_pipeline.py at test size: 48 frames through the 3-stream pipeline, 16 passes (768 frames, ~0.6 s), every uint8
    frame identical to the sequential render.  Two causes it has caught: the packed-f32 op_sel erratum (f16mx 2 of 768) and
    the missing LDS-read wait in front of the raw barrier of the DMA-pipelined conv loop (f16mx 55 of 1 920, i.e. ~22 expected
    here)."""
    torch = torch_cuda
    from real3dportrait_amd import TriPlaneGenerator, synth
    from real3dportrait_amd.frames import ClipRenderer, PipelinedClipRenderer
    G = TriPlaneGenerator().cuda().eval()
    dec = synth.synth_decoder(7, sigma_bias=4.0)
    with torch.no_grad():
        G.decoder.net[0].weight.copy_(T(torch, dec[0])); G.decoder.net[0].bias.copy_(T(torch, dec[1]))
        G.decoder.net[2].weight.copy_(T(torch, dec[2])); G.decoder.net[2].bias.copy_(T(torch, dec[3]))
    params = synth.synth_sr_params(7)
    load_block(torch, G.superresolution.block0, params[0]); load_block(torch, G.superresolution.block1, params[1])
    for b in (G.superresolution.block0, G.superresolution.block1):
        b.precision = precision
    n = 48
    cano = T(torch, synth.synth_planes(7, N=1)); res = [T(torch, synth.synth_planes(8 + i, N=1, scale=0.1)) for i in range(4)]
    cams = T(torch, synth.camera_sweep(n, -0.4, 0.4)); ws = torch.ones(1, 14, 512, device="cuda")
    clip = ClipRenderer(G, cano, res, cams, ws, base_seed=7)
    ref = torch.stack([clip.render_u8(t).clone() for t in range(n)])
    torch.cuda.synchronize()
    pipe = PipelinedClipRenderer(G, cano, res, cams, ws, base_seed=7, n_streams=3)
    ring = torch.zeros(n, 512, 512, 3, dtype=torch.uint8, device="cuda")
    bad = 0
    passes = 16
    for _ in range(passes):
        ring.zero_()
        for t in range(n):
            pipe.render_u8(t, out=ring[t:t + 1])
        pipe.sync(); torch.cuda.synchronize()
        bad += int((ring != ref).flatten(1).any(dim=1).sum())
    assert bad == 0, "%d of %d pipelined frames differ from the sequential render" % (bad, passes * n)


def test_build_with_packed_f32_agrees():
    """The product is compiled without packed-f32 instructions (round 5: they buy nothing next to MFMAs and one of their forms is hazardous on gfx950,
    DESIGN 4.1a).  Its A/B partner is the same source WITH them, every translation unit through the op_sel rewriter (`make PK=1`, built by
    __graft_entry__.build_test_helpers(); the product build of rounds 3-4).  The two compilations contract different mul + add pairs
    (fp-contract=fast around the SLP vectoriser), so the comparison is a tight tolerance -- fp32 ray-kernel outputs within 1e-5, uint8 frames within
    one count in < 0.2 % of the bytes, the torso frame's fp32 image within 3e-4 of its maximum (nine f16mx convolutions deep) -- where a wrong
    operand in a quarter of the lanes is off by the operand's magnitude: head frames (both SR precisions), four ray-kernel shapes, the torso frame."""
    import subprocess
    import sys
    pk = os.path.join(ROOT, "tests", "_build", "libr3d_hip_pk.so")
    assert os.path.exists(pk), "tests/_build/libr3d_hip_pk.so is missing: __graft_entry__.build() builds it"
    listing = open(os.path.join(ROOT, "tests", "_build", "obj_pk", "r3d_render.fix.s")).read()
    assert "v_pk_fma_f32" in listing and "v_pk_mul_f32" in listing                      # it really is the build with them
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gpu_build_ab.py"), "default", "tests/_build/libr3d_hip_pk.so"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    print(r.stdout)
    assert r.returncode == 0 and "\nAGREE" in r.stdout, (r.stdout[-3000:], r.stderr[-1500:])
    assert r.stdout.count("compare ") >= 3 + 4 * 3 and " BAD" not in r.stdout
