#!/usr/bin/env python3
"""Benchmark of the per-frame hot path: TriPlaneGenerator.synthesis() on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step = one 512x512 frame per GPU: (cano + residual_t) tri-planes -> channel-last layout -> 128^2 rays ->
48 coarse + 48 importance samples/ray (the reference's "48 depth samples": num_samples_coarse/fine,
egs/egs_bases/eg3d/base.yaml:39-40) -> fused ray kernel -> SuperresolutionHybrid8XDC 128^2 -> 512^2 ->
clamp -> uint8 frame in a device ring.  Frames are independent units: rank r renders its own K frames
(weak scaling, no data-path collective) and ONE gather to rank 0 re-assembles the clip inside the timed region.
Synthetic tri-planes / decoder / SR weights / cameras (no checkpoints offline).

Prints ONE JSON line on rank 0 (see README of the task): metric, value (whole-job frames/s), roofline of the
dominant kernel (conv_mfma_kernel, MFMA-bound) from HIP events recorded on the launch stream inside the timed
region, and the CPU oracle timed on this box's host cores (a baseline, not the target).
"""
import argparse
import json
import math
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_F16_MFMA_TFLOPS = 2500.0         # dense f16/bf16 MFMA peak (same guide)
PEAK_HBM_TBPS = 8.0
PEAK_ENGINE_CLOCK_GHZ = 2.4           # "Max clock 2400 MHz" (same guide): the clock the MFMA peaks above are quoted at

# algorithmic conv FLOPs of one frame: 2*taps*Cin*Cout*pixels for the four conv launches
def conv_flops_per_frame(r=128):
    t0 = 2 * 9 * 32 * 256 * r * r              # block0.conv0 transposed conv (9 taps per INPUT pixel)
    c1 = 2 * 9 * 256 * 256 * (2 * r) ** 2      # block0.conv1
    t2 = 2 * 9 * 256 * 128 * (2 * r) ** 2      # block1.conv0 transposed conv
    c3 = 2 * 9 * 128 * 128 * (4 * r) ** 2      # block1.conv1
    return [t0, c1, t2, c3]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--device-warmup-ms", type=float, default=500.0,
                    help="untimed frames rendered for this long before the W warm-up steps (brings the GPU's clocks to their sustained state; 0 = off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("R3D_BENCH_STREAMS", "3")),
                    help="HIP streams that consecutive frames are issued on (frames are independent)")
    ap.add_argument("--sr-precision", default=None, choices=["f16mx", "f16x3", "f32"],
                    help="SR precision of the timed frames; default = the LIBRARY default (superresolution.DEFAULT_SR_PRECISION = 'f16mx' since round 5, "
                         "or R3D_SR_PRECISION): the cross products of each SR conv on the block-scaled 8-bit MFMA, e5m2 activation records -- inside the "
                         "2e-4 of SURVEY 8(d) on every reference golden and heavy-tail sweep (tests/test_gpu_mx.py, tests/test_gpu_pinned_config.py). "
                         "The fp32-class 'f16x3' is reported as alt_f16x3 in the same line")
    ap.add_argument("--no-traffic", action="store_true", help="do not re-measure roofline.traffic with rocprofv3 --pmc child runs")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--clip", type=int, default=0,
                    help="strong-scaling mode (BASELINE config 3): render ONE clip of this many frames, frame-sharded over the ranks "
                         "(uneven tail), gathered to rank 0 inside the timed region; --steps is ignored")
    return ap.parse_args()


def build_scene(torch, dev, seed=7, n_frames=64, precision=None):
    import numpy as np
    from real3dportrait_amd import TriPlaneGenerator, synth
    from real3dportrait_amd.frames import ClipRenderer
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    G = TriPlaneGenerator().to(dev).eval()
    dec = synth.synth_decoder(seed, sigma_bias=4.0)
    with torch.no_grad():
        G.decoder.net[0].weight.copy_(T(dec[0])); G.decoder.net[0].bias.copy_(T(dec[1]))
        G.decoder.net[2].weight.copy_(T(dec[2])); G.decoder.net[2].bias.copy_(T(dec[3]))
        for blk, p in zip((G.superresolution.block0, G.superresolution.block1), synth.synth_sr_params(seed)):
            for name in ("conv0", "conv1", "torgb"):
                l = getattr(blk, name); w, b, aw, ab = p[name]
                l.weight.copy_(T(w)); l.bias.copy_(T(b)); l.affine.weight.copy_(T(aw)); l.affine.bias.copy_(T(ab))
    cano = T(synth.synth_planes(seed, N=1))
    residuals = [T(synth.synth_planes(seed + 1 + i, N=1, scale=0.1)) for i in range(4)]
    cams = T(synth.camera_sweep(n_frames, -0.4, 0.4))
    ws = torch.ones(1, 14, 512, device=dev)
    clip = ClipRenderer(G, cano, residuals, cams, ws, base_seed=seed, precision=precision)
    return G, clip, dec, (cano, residuals, cams)


def cpu_baseline(seed=7):
    """The C oracle (oracle/r3d_oracle.c, OpenMP) renders the same frame on the host cores."""
    import numpy as np
    from oracle import Oracle
    from real3dportrait_amd import synth
    orc = Oracle()
    planes = synth.synth_planes(seed, N=1) + synth.synth_planes(seed + 1, N=1, scale=0.1)
    dec = synth.synth_decoder(seed, sigma_bias=4.0)
    sr = synth.synth_sr_params(seed)
    cam = synth.camera_sweep(64, -0.4, 0.4)[:1]
    R, Nc, Nf = 128, 48, 48
    noise_c = synth.synth_noise(seed, (1, R * R, Nc, 1)); u_f = synth.synth_noise(seed + 1, (R * R, Nf))
    ws = np.ones((14, 512), np.float32)

    def frame():
        t0 = time.perf_counter()
        o, d = orc.raygen(cam[:, :16], cam[:, 16:], R)
        rgb, depth, wsum, valid = orc.render(planes, dec, o, d, Nc, Nf, noise_c, u_f)
        t1 = time.perf_counter()
        feat = np.ascontiguousarray(rgb[0].T.reshape(32, R, R))
        orc.superresolution(np.ascontiguousarray(feat[:3]), feat, sr, ws)
        return t1 - t0, time.perf_counter() - t1
    t_render, t_sr = frame()                      # first call: page faults of the oracle's scratch buffers, OpenMP thread start
    runs = 1
    if t_render + t_sr <= 12.0:                   # bounded sample: at most ~30 s of CPU work
        for _ in range(2):
            r, q = frame()
            runs += 1
            if r + q < t_render + t_sr:
                t_render, t_sr = r, q
    sample = "1 full frame, best of %d: render R=128 48+48 (%.2fs) + SR 128^2->512^2 (%.2fs)" % (runs, t_render, t_sr)
    return {"value": 1.0 / (t_render + t_sr), "unit": "frames/s", "cores": orc.num_threads, "kind": "port",
            "sample": sample,
            "note": "the C oracle = the parity CHECKER (scalar restatement of the reference, OpenMP over rays / output rows), not a tuned CPU "
                    "renderer: its SR does not scale with the core count; the reference's own PyTorch CPU path is `cpu_baseline_reference`"}


def cpu_baseline_reference():
    """The REFERENCE's own PyTorch CPU renderer + SR on the same frame (modules/eg3ds/volumetric_rendering/renderer.py:118-167 +
    models/superresolution.py:348-359), timed IN THIS RUN on this box's host cores whenever a reference checkout is reachable
    (R3D_REFERENCE or /root/reference; scripts/time_reference_cpu.py in a subprocess, so that its torch thread settings and module
    stand-ins stay out of this process).  The GPU box of the driver has no checkout: the committed measurement of the build container
    is then reported, and says so."""
    ref = os.environ.get("R3D_REFERENCE", "/root/reference")
    r, measured = None, None
    if os.path.isdir(os.path.join(ref, "modules", "eg3ds")):
        try:
            out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "time_reference_cpu.py")], capture_output=True, text=True,
                                 timeout=900, cwd="/tmp", env={**os.environ, "PYTHONDONTWRITEBYTECODE": "1", "R3D_REFERENCE": ref})
            r = json.loads(out.stdout)
            measured = "in this run, on this box's %d host cores (reference checkout %s)" % (r["cores"], ref)
        except Exception as e:      # noqa: BLE001
            r, measured = None, "in-run measurement failed (%s)" % type(e).__name__
    if r is None:
        for name in ("cpu_reference_r03.json", "cpu_reference_r02.json"):
            try:
                r = json.load(open(os.path.join(ROOT, "profiles", name)))
            except Exception:       # noqa: BLE001
                continue
            measured = "NOT in this run: no reference checkout on this box; committed measurement profiles/%s (%s, %s, %d cores)%s" % (
                name, r.get("where", "?"), r.get("when", "?"), r["cores"], "" if measured is None else "; " + measured)
            break
    if r is None:
        return None
    return {"value": round(r["value"], 4), "unit": r["unit"], "cores": r["cores"], "kind": "reference",
            "sample": "1 full frame: reference ImportanceRenderer.forward R=128 48+48 (%.2fs) + SuperresolutionHybrid8XDC.forward "
                      "128^2->512^2 (%.2fs), torch %s CPU, best of 3" % (r["render_s"], r["sr_s"], r["torch"]),
            "measured": measured}


VALU_INSTS = {}


def measure_traffic(prec):
    """HBM bytes per launch of the conv kernels, measured NOW: two `rocprofv3 --pmc` child runs of this script (FETCH_SIZE and WRITE_SIZE
    need separate passes, MI355X_MICROARCH.md "rocprofv3 PMC slots"; --pmc only, no trace domain), 6 frames on one stream each.
    bytes = 2 x FETCH_SIZE (gfx950's counter tallies 128-byte requests as 64 B, same guide, "HBM") + WRITE_SIZE, both reported in KiB.
    Returns {kernel name: bytes per launch} or None when rocprofv3 is not on PATH / a pass fails (the committed figure is then used)."""
    import collections
    import csv
    import glob
    import shutil
    import tempfile
    tool = shutil.which("rocprofv3")
    if tool is None:
        return None, "rocprofv3 not on PATH"
    tmp = tempfile.mkdtemp(prefix="r3d_traffic_", dir="/tmp")
    vals = collections.defaultdict(dict)
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
            cmd = [tool, "--pmc", ctr, "--output-format", "csv", "-d", os.path.join(tmp, ctr), "-o", "p", "--", sys.executable,
                   os.path.abspath(__file__), "--traffic-child", "--sr-precision", prec, "--steps", "6"]
            # (the child is a plain single-process run: no rendezvous variables of a torch.distributed launcher may leak into it)
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK",
                                                                      "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID") and not k.startswith("TORCHELASTIC_")}
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd="/tmp", env={**env, "TMPDIR": "/tmp"})
            files = glob.glob(os.path.join(tmp, ctr, "**", "p_counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                if ctr == "SQ_INSTS_VALU":      # the optional third pass (the ray kernel's issue-bound figure): the traffic figures stand without it
                    continue
                return None, "rocprofv3 --pmc %s pass failed (rc %d)" % (ctr, r.returncode)
            acc = collections.defaultdict(list)
            for f in files:
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] == ctr:
                        acc[row["Kernel_Name"].split("(")[0].replace("void r3d::", "").strip()].append(float(row["Counter_Value"]))
            for k, v in acc.items():
                vals[k][ctr] = sum(v) / len(v)
    except Exception as e:      # noqa: BLE001
        return None, "traffic measurement failed (%s)" % type(e).__name__
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = {k: int((2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024) for k, v in vals.items() if "FETCH_SIZE" in v and "WRITE_SIZE" in v}
    global VALU_INSTS
    VALU_INSTS = {k: v["SQ_INSTS_VALU"] for k, v in vals.items() if "SQ_INSTS_VALU" in v}      # wave-level VALU instructions per launch (third pass)
    return out, "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this run (6 frames, one stream): 2 x FETCH_SIZE + WRITE_SIZE per launch"


def build_torso_frame(torch, dev, G, seed=7, fused_input=True, precision=None):
    """BASELINE config 4 surrogate: everything real3d_infer.py:480-492 runs per frame with the shipped torso model that is on the
    hot path -- to_plane_cnn (segformer.py:691-700) -> flips + cano add + layout -> rays -> fused ray kernel -> fused
    SuperresolutionHybrid8XDC_Warp.forward (block0, torso/background fusion convs at 256^2, SynthesisBlockNoUp, block1) -> uint8.
    The cold encoders in front (MiT SegFormer) and the face-vid2vid warp network are out of scope: their per-frame OUTPUTS are
    synthetic tensors of the documented shapes (a stand-in torso_model returns them without computing)."""
    import numpy as np
    from real3dportrait_amd import synth
    from real3dportrait_amd.sr_with_ref import SuperresolutionHybrid8XDC_Warp
    from real3dportrait_amd.superresolution import Conv2d, ConvStack, const_bound
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    class StandInTorso(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.rgb_torso = T(synth.hash_unitvar(seed, (1, 3, 256, 256), stream=31) * np.float32(0.3))
            self.ret = {"deformed_torso_hid": T(synth.hash_unitvar(seed, (1, 64, 256, 256), stream=32)),
                        "occlusion_2": T(synth.synth_noise(seed, (1, 1, 64, 64), stream=33))}

        def forward(self, *a, **k):
            return self.rgb_torso, self.ret
    sr = SuperresolutionHybrid8XDC_Warp(32, 512, 0, True, torso_model=StandInTorso()).to(dev).eval()
    from real3dportrait_amd.superresolution import set_sr_precision
    set_sr_precision(sr, precision if precision is not None else G.superresolution.block0.precision)
    with torch.no_grad():
        for blk, p in ((sr.block0, synth.synth_sr_block(seed, 32, 256, 512, 100)), (sr.block1, synth.synth_sr_block(seed, 256, 128, 512, 200)),
                       (sr.head_torso_block, synth.synth_sr_block(seed, 256, 256, 512, 400))):
            for name in ("conv0", "conv1", "torgb"):
                l = getattr(blk, name); w, b, aw, ab = p[name]
                l.weight.copy_(T(w)); l.bias.copy_(T(b)); l.affine.weight.copy_(T(aw)); l.affine.bias.copy_(T(ab))
        for i, (name, plan) in enumerate(synth.FUSION_STACKS.items()):
            for m, (w, b) in zip([m for m in getattr(sr, name) if hasattr(m, "weight")], synth.synth_conv_stack(seed, plan, 300 + 20 * i)):
                m.weight.copy_(T(w)); m.bias.copy_(T(b))
        mods = []
        for i, ((ci, co, k, lrelu), (w, b)) in enumerate(zip(synth.TO_PLANE_CNN, synth.synth_conv_stack(seed, synth.TO_PLANE_CNN, 500))):
            if i == synth.TO_PLANE_CNN_UP_BEFORE:
                mods.append(torch.nn.UpsamplingBilinear2d(scale_factor=2.))
            c = Conv2d(ci, co, k, 1, padding=1)
            c.weight.copy_(torch.from_numpy(w)); c.bias.copy_(torch.from_numpy(b))
            mods.append(c)
            if lrelu:
                mods.append(torch.nn.LeakyReLU(0.01))
        cnn = ConvStack(*mods).to(dev)
    feat = T(synth.hash_unitvar(seed, (1, 256, 128, 128), stream=41))          # fused SegFormer feature map (output of the cold encoder)
    cano = T(synth.synth_planes(seed, N=1))
    ref_torso, ref_bg = T(synth.hash_unitvar(seed, (1, 3, 512, 512), stream=42) * np.float32(0.5)), T(synth.hash_unitvar(seed, (1, 3, 512, 512), stream=43) * np.float32(0.5))
    cams = T(synth.camera_sweep(8, -0.4, 0.4))
    ws = torch.ones(1, 14, 512, device=dev)
    G.renderer.noise_mode = "hash"

    def frame(t):
        raw = cnn(feat)                                                                            # [1,96,256,256], not flipped
        planes = G.renderer.prepare_planes(cano, add=raw, add_flip=G.renderer.SECC_PLANE_FLIPS)
        cam = cams[t % 8: t % 8 + 1]
        if not fused_input:            # the unfused reference sequence (tests: must give the same frame bit for bit)
            o, d = G.ray_sampler(cam[:, :16].view(-1, 4, 4), cam[:, 16:25].view(-1, 3, 3), 128)
            fe, depth, wsum, valid = G.renderer(planes, G.decoder, o, d, G.rendering_kwargs)
            fimg = fe.permute(0, 2, 1).reshape(1, 32, 128, 128).contiguous()
            fimg._r3d_bound = const_bound(1.01, 1, dev)
            wimg = wsum.permute(0, 2, 1).reshape(1, 1, 128, 128).contiguous()
            return sr(fimg[:, :3], fimg, ws, ref_torso, ref_bg, wimg, None, None, None, noise_mode="none")[0]
        # rays generated inside the render launches; the ray kernel also writes block0's SPLIT operand (no conversion launch); only the
        # image leaves the frame, so no depth image (real3d_infer.py:480-492 keeps `image` only)
        keep, G.renderer.need_depth = G.renderer.need_depth, False
        try:
            fe, depth, wsum, valid = G.renderer.forward_camera(planes, G.decoder, cam[:, :16].view(-1, 4, 4), cam[:, 16:25].view(-1, 3, 3), 128,
                                                               G.rendering_kwargs, _split_for=sr.split_input_spec(ws, 1, dev))
        finally:
            G.renderer.need_depth = keep
        fimg = fe.permute(0, 2, 1).reshape(1, 32, 128, 128).contiguous()
        fimg._r3d_bound = const_bound(1.01, 1, dev)
        wimg = wsum.permute(0, 2, 1).reshape(1, 1, 128, 128).contiguous()
        x = fe._r3d_split if getattr(fe, "_r3d_split", None) is not None else fimg
        img, _ = sr(fimg[:, :3], x, ws, ref_torso, ref_bg, wimg, None, None, None, noise_mode="none")
        return img
    # algorithmic conv FLOPs of this frame (2 * taps * Cin * Cout * pixels): to_plane_cnn + SR blocks + fusion stacks + NoUp block
    px = 256 * 256
    fl = 3 * 2 * 9 * 256 * 256 * 128 * 128 + 2 * 9 * 256 * 96 * px                                 # to_plane_cnn
    fl += sum(conv_flops_per_frame(128))                                                           # block0 + block1
    fl += 2 * 64 * 256 * px                                                                        # torso_encoder (1x1)
    fl += 2 * 9 * 512 * 256 * px + 2 * 9 * 256 * 256 * px                                          # fuse_head_torso_convs
    fl += 2 * 2 * 9 * 256 * 256 * px                                                               # head_torso_block conv0 + conv1
    fl += 2 * 512 * 64 * px + 2 * 9 * 64 * 256 * px + 2 * 9 * 256 * 256 * px                      # fuse_fg_bg_convs
    return frame, fl


def sustained_mix_probe(conv_tflops, seconds=1.5):
    """What the f16mx conv's MFMA mix (2 f16 + 1 block-scaled fp8 MFMA per 2 taps x 16 channels and tile) sustains on this board for `seconds` per
    variant when only the matrix pipe -- and then its LDS feed -- is busy: constant register operands, random register operands, random operands from
    LDS at the conv's read ratio.  No memory traffic, barrier or epilogue in any of them; the board clocks each to its power budget.  The dominant
    kernel's algorithmic rate is reported as a fraction of the LDS-fed variant beside `roofline.frac`."""
    import re, subprocess
    exe = os.path.join(ROOT, "scripts", "probes", "bin", "power_ceiling_probe")
    if not os.path.exists(exe):
        return {"skipped": "scripts/probes/bin/power_ceiling_probe not built (__graft_entry__.build())"}
    try:
        txt = subprocess.run([exe, str(seconds)], capture_output=True, text=True, timeout=120).stdout
    except Exception as e:      # noqa: BLE001
        return {"skipped": "probe failed: %r" % (e,)}
    res = {"what": "scripts/probes/power_ceiling_probe.hip, %.1f s per variant, 4 waves/SIMD x 64 accumulators (the conv's shape); algorithmic TFLOP/s of the "
                   "f16mx MFMA mix, the clock its waves ran at and the socket power meanwhile" % seconds}
    for line in txt.splitlines():
        m = re.match(r"(R0|R1|L1):.*?([0-9.]+) ms/launch\s+([0-9.]+) algorithmic TFLOP/s = ([0-9.]+) of 2500 .*?waves at ([0-9.]+) GHz\s+socket\s+([0-9.]+) W", line)
        if m:
            res[{"R0": "registers_constant", "R1": "registers_random", "L1": "lds_fed_random"}[m.group(1)]] = {
                "tflops": float(m.group(3)), "frac_of_f16_peak": float(m.group(4)), "wave_clock_ghz": float(m.group(5)), "socket_w": float(m.group(6))}
    if conv_tflops and "lds_fed_random" in res and res["lds_fed_random"]["tflops"] > 0:
        res["dominant_kernel_vs_lds_fed_mix"] = round(conv_tflops / res["lds_fed_random"]["tflops"], 3)
        res["dominant_kernel_vs_register_random_mix"] = round(conv_tflops / res["registers_random"]["tflops"], 3) if "registers_random" in res else None
    return res


def power_probe(torch, step, sync, K, seconds=3.0):
    """Socket power and sclk as rocm-smi reports them while the headline frame loop runs (a sampler thread polls the CLI; the GPU work is not
    touched).  The frame loop of this path runs AT THE BOARD'S POWER CAP: frames/s is joules per frame, and the MFMA peaks -- quoted at the
    2.4 GHz maximum engine clock -- are not reachable by any kernel that keeps the matrix cores, LDS and L2 busy together."""
    import re, shutil, subprocess, threading
    if not shutil.which("rocm-smi"):
        return {"skipped": "rocm-smi not on PATH"}
    samples, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            try:
                t = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True, timeout=10).stdout
            except Exception:       # noqa: BLE001
                return
            w = re.search(r"Power \(W\):\s*([0-9.]+)", t); c = re.search(r"sclk clock level:\s*\d+:\s*\((\d+)Mhz\)", t)
            j = re.search(r"Sensor junction\) \(C\):\s*([0-9.]+)", t)
            if w and c:
                samples.append((float(w.group(1)), float(c.group(1)), float(j.group(1)) if j else None))
    cap = None
    try:
        t = subprocess.run(["rocm-smi", "--showmaxpower"], capture_output=True, text=True, timeout=10).stdout
        m = re.search(r"Max Graphics Package Power \(W\):\s*([0-9.]+)", t)
        cap = float(m.group(1)) if m else None
    except Exception:               # noqa: BLE001
        pass
    th = threading.Thread(target=poll, daemon=True)
    t0 = time.perf_counter(); n = 0
    th.start()
    while time.perf_counter() - t0 < seconds:
        for i in range(K):
            step(i)
        sync(); n += K
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    stop.set(); th.join(timeout=15)
    s2 = samples[1:] if len(samples) > 2 else samples          # the first sample may predate the ramp
    if not s2:
        return {"skipped": "rocm-smi gave no samples"}
    mean = lambda v: round(sum(v) / len(v), 1)
    w = [a for a, _, _ in s2]
    return {"what": "rocm-smi --showpower --showclocks polled while the headline frame loop ran for %.1f s (%d frames, %.0f frames/s)" % (el, n, n / el),
            "socket_power_w_mean": mean(w), "socket_power_w_max": max(w), "max_package_power_w": cap,
            "sclk_mhz_mean": mean([b for _, b, _ in s2]), "junction_c_max": max([c for _, _, c in s2 if c is not None], default=None),
            "samples": len(s2), "joules_per_frame": round(mean(w) / (n / el), 4)}


def clip125(torch, dev, G, scene, clip, streams):
    """BASELINE configs[2] on ONE GPU: the 125 frames of a 5 s clip @ 25 fps through the stream pipeline into a device ring (the 8-GPU
    run shards the same clip: bench.py --gpus 8 --clip 125)."""
    from real3dportrait_amd.frames import PipelinedClipRenderer
    cano, residuals, cams = scene
    n = 125
    import numpy as np
    from real3dportrait_amd import synth
    cams125 = torch.from_numpy(np.ascontiguousarray(synth.camera_sweep(n, -0.4, 0.4))).to(dev)
    pipe = PipelinedClipRenderer(G, cano, residuals, cams125, clip.ws, base_seed=clip.base_seed, n_streams=max(1, streams))
    ring = torch.zeros(n, 512, 512, 3, dtype=torch.uint8, device=dev)
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for t in range(n):
            pipe.render_u8(t, out=ring[t:t + 1])
        pipe.sync(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t1)
    return {"what": "125-frame clip (5 s @ 25 fps) on one GPU, %d streams, uint8 ring; best of 3" % max(1, streams), "ms_per_clip": round(best * 1e3, 2),
            "fps": round(n / best, 1), "realtime_factor": round(n / best / 25.0, 1)}


def cfg5_stress(torch, dev, lib, precision="f16mx"):
    """BASELINE configs[4]: N = 8 novel-view cameras of one tri-plane per batch, R = 256, 96 + 96 samples, SR 256^2 -> 512^2 -> 1024^2 (the
    reference SR asserts a 512 output, superresolution.py:334: as SURVEY 8(d) defines it, the same two SynthesisBlocks at twice the size)."""
    import ctypes
    import numpy as np
    from real3dportrait_amd import ImportanceRenderer, OSGDecoder, RaySampler, SynthesisBlock, synth, _lib
    from real3dportrait_amd.superresolution import chain_fold, const_bound
    N, R, Nc, Nf = 8, 256, 96, 96
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    planes = T(synth.synth_planes(7, N=1)).repeat(N, 1, 1, 1, 1)
    dn = synth.synth_decoder(7, sigma_bias=4.0)
    dec = OSGDecoder().to(dev)
    with torch.no_grad():
        dec.net[0].weight.copy_(T(dn[0])); dec.net[0].bias.copy_(T(dn[1])); dec.net[2].weight.copy_(T(dn[2])); dec.net[2].bias.copy_(T(dn[3]))
    cams = T(synth.camera_sweep(N, -0.4, 0.4))
    b0 = SynthesisBlock(32, 256, w_dim=512, resolution=512, img_channels=3, is_last=False, conv_clamp=None).to(dev)
    b1 = SynthesisBlock(256, 128, w_dim=512, resolution=1024, img_channels=3, is_last=True, conv_clamp=None).to(dev)
    with torch.no_grad():
        for blk, p in zip((b0, b1), synth.synth_sr_params(7)):
            for name in ("conv0", "conv1", "torgb"):
                l = getattr(blk, name); w, b, aw, ab = p[name]
                l.weight.copy_(T(w)); l.bias.copy_(T(b)); l.affine.weight.copy_(T(aw)); l.affine.bias.copy_(T(ab))
    b0.precision = b1.precision = precision
    b0.out_format = "split_mx" if precision == "f16mx" else "split"; b1.return_x = False      # f16mx: fp8 records for block1's up-sampling conv
    ren = ImportanceRenderer(hp={}); ren.noise_mode = "hash"; ren.seed = 5
    opts = {"ray_start": "auto", "ray_end": "auto", "box_warp": 1.0, "depth_resolution": Nc, "depth_resolution_importance": Nf,
            "disparity_space_sampling": False, "clamp_mode": "softplus", "white_back": False}
    ws = torch.ones(N, 3, 512, device=dev)
    rs = RaySampler()
    nhwc = ren.prepare_planes(planes)

    mx_slot = torch.zeros(N, device=dev, dtype=torch.float32)

    def batch():
        o, d = rs(cams[:, :16].view(-1, 4, 4), cams[:, 16:].view(-1, 3, 3), R)
        feat, depth, wsum, valid = ren(nhwc, dec, o, d, opts)
        fimg = feat.permute(0, 2, 1).reshape(N, 32, R, R).contiguous()
        prep0, prep1 = b0.prepare(ws, dev), b1.prepare(ws, dev)
        b0._depth_in, b1._depth_in = 0, 2
        from real3dportrait_amd import superresolution as _srm
        mx = b0.precision == "f16mx" and _srm._MX_TAIL_FOLD      # as SuperresolutionHybrid8XDC.forward (R3D_MX_TAIL_FOLD=1: block0 measures max|x0|, block1's conv1 operand is re-folded from it)
        chain_fold([b0.chain_op(-1), b1.chain_op(0)], N, [const_bound(1.01, N, dev)], zero=[mx_slot] if mx else ())
        x, rgb = b0(fimg, fimg[:, :3].contiguous(), ws, noise_mode="none", _prepared=prep0, _next=b1, _folded=True, _x_absmax=mx_slot if mx else None)
        if mx:
            chain_fold([b1.chain_op(-1, tail=True)], N, [mx_slot])
        return b1(x, rgb, ws, noise_mode="none", _prepared=prep1, _folded=True)[1]
    for _ in range(2):
        img = batch()
    torch.cuda.synchronize()
    assert tuple(img.shape) == (N, 3, 1024, 1024) and bool(torch.isfinite(img).all())
    lib.r3d_profile_configure(0x7F); lib.r3d_profile_reset()
    reps = 4
    t1 = time.perf_counter()
    for _ in range(reps):
        batch()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t1) / reps
    ms, cnt = ctypes.c_double(0), ctypes.c_int(0)
    fam = {}
    for j, nme in enumerate(["render", "conv_mfma", "upconv_fir", "torgb", "sr_pack", "layout", "misc"]):
        _lib.check(lib.r3d_profile_read(j, ctypes.byref(ms), ctypes.byref(cnt)), "profile_read")
        fam[nme] = round(ms.value / reps, 4)
    lib.r3d_profile_configure(0)
    flops = N * sum(conv_flops_per_frame(256))
    conv_ms = fam["conv_mfma"] + fam["upconv_fir"]
    S = N * R * R * (Nc + Nf)
    return {"what": "N=8 cameras per batch, R=256, 96+96 samples, SR 256^2 -> 1024^2 (%s), one stream" % b0.precision,
            "ms_per_batch": round(dt * 1e3, 3), "fps": round(N / dt, 1), "breakdown_ms_per_batch": fam,
            "roofline": {"bound": "mfma", "achieved": round(flops / (conv_ms * 1e-3) / 1e12, 2), "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(flops / (conv_ms * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS, 4), "conv_gflop_per_batch": round(flops / 1e9, 1),
                         "note": "all four SR convolutions of the batch (algorithmic FLOPs / summed conv kernel time)"},
            "render_kernel": {"ms": fam["render"], "algorithmic_GBps": round(S * 1536 / (fam["render"] * 1e-3) / 1e9, 1),
                              "note": "gather-bound view (SURVEY 8d): S * 1536 B of taps, served by L1 / L2"}}


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:      # convenience: self-launch one process per GPU
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", "29511", os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import torch
    import torch.distributed as dist
    from real3dportrait_amd import _lib
    rank = int(os.environ.get("RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback exists)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or "RANK" in os.environ          # under torchrun even a 1-rank job exercises the RCCL path
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)
    lib = _lib.load()

    K, W = args.steps, args.warmup
    from real3dportrait_amd.frames import shard_frames
    clip_lo = 0
    if args.clip > 0:                                    # strong scaling: this rank's contiguous chunk of ONE clip
        clip_lo, clip_hi = shard_frames(args.clip, world, rank)
        K = (args.clip + world - 1) // world             # ring slots per rank (the gather moves equal-sized rings)
        K_mine = clip_hi - clip_lo
    else:
        K_mine = K
    from real3dportrait_amd.superresolution import DEFAULT_SR_PRECISION
    prec = args.sr_precision or os.environ.get("R3D_SR_PRECISION", DEFAULT_SR_PRECISION)      # the precision of `value` IS the product's default
    G, clip, dec, scene = build_scene(torch, dev, n_frames=max(64, K * world, args.clip), precision=prec)
    ring = torch.zeros(K, 512, 512, 3, dtype=torch.uint8, device=dev)
    if args.traffic_child:                               # child of measure_traffic(): a few frames on one stream under rocprofv3 --pmc
        for i in range(6):
            clip.render_u8(i, out=ring[i % K:i % K + 1])
        torch.cuda.synchronize()
        return

    pipe = None
    if args.streams > 1:
        from real3dportrait_amd.frames import PipelinedClipRenderer
        cano, residuals, cams = scene
        pipe = PipelinedClipRenderer(G, cano, residuals, cams, clip.ws, base_seed=clip.base_seed, n_streams=args.streams)

    def step(i):
        t = (clip_lo if args.clip > 0 else rank * K) + i
        (pipe or clip).render_u8(t, out=ring[i:i + 1])

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    lib.r3d_profile_configure(0x7F)       # creates the event pools (all families) NOW, not between the warm-up and the timed region
    lib.r3d_profile_configure(0)
    # Device warm-up (untimed, in front of the W warm-up steps): a step is 0.8 ms, so W = 5 steps are 4 ms of GPU work -- the chip's
    # power management needs ~20 ms under load to reach its sustained clocks and drops them again within 50 ms of idle
    # (scripts/gpu_warm_probe.py: 20-frame chunks render at 1 218 / 1 330 / 1 367 / 1 360 frames/s back to back, 1 181 again after a
    # 50 ms pause).  The same frames are rendered until --device-warmup-ms of wall time have passed; the timed region is unchanged.
    # Round 4: the frame loop runs at the board's power cap (1.36 kW of 1.4, `power` in the line), and the power management takes longer to
    # settle than 100 ms of 8-frame chunks with a host sync between them: five back-to-back 20-frame regions read 1 565 / 1 538 / 1 637 /
    # 1 655 / 1 678 frames/s after 100 ms, 1 658 / 1 641 / 1 665 / 1 657 / 1 663 after 500 ms (1 656 +- 20 after 1.5 s and 3 s): default 500 ms.
    warm_frames = 0
    if args.device_warmup_ms > 0:
        for i in range(2 * max(1, args.streams)):          # first touches: device allocations, per-stream workspaces (host-bound, not GPU load)
            step(i % K)
        warm_frames = 2 * max(1, args.streams)
        if pipe is not None:
            pipe.sync()
        torch.cuda.synchronize()
        t_w = time.perf_counter()
        while (time.perf_counter() - t_w) * 1e3 < args.device_warmup_ms:
            for i in range(8):
                step((warm_frames + i) % K)
            warm_frames += 8
            if pipe is not None:
                pipe.sync()
            torch.cuda.synchronize()
    for i in range(max(W, args.streams)):
        step(i % K)
    if pipe is not None:
        pipe.sync()
    from real3dportrait_amd.frames import ClipGatherer
    total_frames = args.clip if args.clip > 0 else K * world
    # clip assembly on rank 0: r3d_gather_frames behind the C ABI (grouped ncclSend / ncclRecv over RCCL) into ONE pre-allocated buffer
    gatherer = ClipGatherer(K, (512, 512), dev, backend="r3d" if use_dist else "torch")
    if use_dist:                                         # warm the gather path too
        clip_out = gatherer.gather(ring, total_frames)
    import ctypes
    lib.r3d_profile_configure(1 << 1); lib.r3d_profile_reset()      # event pairs around the dominant kernel, on its launch stream
    def timed_region():
        barrier()
        t0 = time.perf_counter()
        for i in range(K_mine):
            step(i)
        if pipe is not None:
            pipe.sync()
        if use_dist:
            clip_out = gatherer.gather(ring, total_frames)
            if rank == 0:
                assert clip_out.shape == (total_frames, 512, 512, 3)
        barrier()
        return time.perf_counter() - t0

    elapsed = timed_region()                             # THE measurement (`value`): W warm-up steps, then exactly K timed steps
    lib.r3d_profile_configure(0)
    tms, tcnt = ctypes.c_double(0), ctypes.c_int(0)
    _lib.check(lib.r3d_profile_read(1, ctypes.byref(tms), ctypes.byref(tcnt)), "profile_read")
    in_region_ms = tms.value / max(1, tcnt.value)        # includes whatever other streams' kernels shared the GPU
    el = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    # the same K-step region four more times, back to back (run-to-run spread of a 15 ms measurement; `value` stays the first one)
    rep_s = [elapsed]
    for _ in range(4):
        e = torch.tensor([timed_region()], device=dev, dtype=torch.float64)
        if use_dist:
            dist.all_reduce(e, op=dist.ReduceOp.MAX)
        rep_s.append(float(e.item()))

    # ---- roofline of the dominant kernel (conv_mfma_f16x3_kernel: the two plain 3x3 convs of a frame), MFMA-bound ----
    # HIP event pairs recorded on the launch stream around every conv launch.  With frames pipelined over several
    # streams a bracket would also contain other frames' kernels, so the kernel's own duration is measured over the
    # same K frames issued on ONE stream right after the timed region (what rocprofv3 --stats sees with --streams 1).
    # The duration measured INSIDE the timed region (other frames' kernels share the GPU) is reported next to it.
    lib.r3d_profile_configure((1 << 0) | (1 << 1) | (1 << 2)); lib.r3d_profile_reset()
    for i in range(K):
        clip.render_u8(rank * K + i, out=ring[i:i + 1])
    torch.cuda.synchronize()
    # the shader clock the kernels ran at in that loop: one wave per launch reads s_memtime / s_memrealtime (r3d_profile_clock)
    clock_ghz = {}
    for fam, pid in (("render_kernel", 0), ("conv", 1), ("upconv_fir", 2)):
        g = ctypes.c_double(0)
        _lib.check(lib.r3d_profile_clock(pid, ctypes.byref(g), None), "profile_clock")
        clock_ghz[fam] = round(g.value, 3) if g.value > 0 else None
    lib.r3d_profile_configure(0)
    ms, cnt = ctypes.c_double(0), ctypes.c_int(0)
    _lib.check(lib.r3d_profile_read(1, ctypes.byref(ms), ctypes.byref(cnt)), "profile_read")
    flops = conv_flops_per_frame(128)
    ums, ucnt = ctypes.c_double(0), ctypes.c_int(0)
    _lib.check(lib.r3d_profile_read(2, ctypes.byref(ums), ctypes.byref(ucnt)), "profile_read")
    if prec == "f32":       # the exact-f32 path runs all four convs on one kernel (per-phase transposed conv + FIR kernel)
        dom_flops, launches = sum(flops), 4
    else:                   # f16x3: plain convs (block0.conv1, block1.conv1) on the dominant kernel; the up-sampling convs
        dom_flops, launches = flops[1] + flops[3], 2     # (block0/1.conv0 + FIR + activation) are fused into upconv_fir_f16x3_kernel
    conv_ms_per_frame = ms.value / max(1, cnt.value) * launches
    achieved_tf = dom_flops / (conv_ms_per_frame * 1e-3) / 1e12 if cnt.value else 0.0
    up_ms_per_frame = ums.value / max(1, ucnt.value) * 2
    up_tf = (flops[0] + flops[2]) / (up_ms_per_frame * 1e-3) / 1e12 if (ucnt.value and prec != "f32") else None
    # HBM bytes per launch of the dominant kernel: re-measured in this run when rocprofv3 is available (rank 0 of a 1-GPU run), else the
    # committed figure of profiles/traffic.json
    traffic, traffic_source, traffic_all = None, None, None
    dom_pat = {"f32": lambda k: k.startswith("conv_mfma_kernel"), "f16mx": lambda k: "conv_mfma_f16x3_kernel" in k and "true>" in k,
               "f16x3": lambda k: "conv_mfma_f16x3_kernel" in k and "true>" not in k}[prec]
    if rank == 0 and world == 1 and not args.no_traffic:
        traffic_all, traffic_source = measure_traffic(prec)
        if traffic_all:
            hits = [v for k, v in traffic_all.items() if dom_pat(k)]
            traffic = hits[0] if hits else None
    if traffic is None:
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        try:
            traffic = json.load(open(tpath)).get("conv_bytes_per_launch_" + prec)
            traffic_source = "profiles/traffic.json (committed rocprofv3 --pmc passes; not re-measured in this run: %s)" % (traffic_source or "multi-GPU run or --no-traffic")
        except Exception:       # noqa: BLE001
            traffic = None
    if prec == "f32":       # exact fp32 on v_mfma_f32_32x32x2_f32
        kname, peak, products = "conv_mfma_kernel", PEAK_F32_MFMA_TFLOPS, 1
    elif prec == "f16mx":   # hi*hi on v_mfma_f32_32x32x16_f16 + both cross products of two taps in one v_mfma_scale_f32_32x32x64_f8f6f4 (twice
        # the f16 rate): 2 f16-rate matrix passes per algorithmic MAC; priced against the dense f16 peak (the fp8 peak is 2x that)
        kname, peak, products = "conv_mfma_f16x3_kernel<4, 2, 4, true> (f16mx instantiation)", PEAK_F16_MFMA_TFLOPS, 2
    else:                   # fp32-accurate 3-term fp16 split on v_mfma_f32_32x32x16_f16: 3 MFMA products per algorithmic MAC
        kname, peak, products = "conv_mfma_f16x3_kernel", PEAK_F16_MFMA_TFLOPS, 3
    roofline = {"kernel": kname, "bound": "mfma", "achieved": round(achieved_tf, 2),
                "peak": peak, "unit": "TFLOP/s", "frac": round(achieved_tf / peak, 4),
                "traffic": traffic, "traffic_source": traffic_source,
                # block0.conv1: 256 ch x 256^2 in + out (4 B per value in SPLIT) + weights; block1.conv1: 128 ch x 512^2 in + weights + 2 toRGB partial planes
                "algorithmic_bytes_per_launch": (2 * 256 * 256 * 256 * 4 + 9 * 256 * 256 * 4 + 128 * 512 * 512 * 4 + 9 * 128 * 128 * 4 + 2 * 3 * 512 * 512 * 4) // 2 if prec != "f32" else None,
                "launches_per_frame": launches,
                "avg_launch_ms": round(ms.value / max(1, cnt.value), 4),
                "avg_launch_ms_in_timed_region": round(in_region_ms, 4),
                "algorithmic_gflop_per_launch": round(dom_flops / launches / 1e9, 3),
                "mfma_products_per_mac": products,
                "executed_tflops": round(achieved_tf * products, 1), "pipe_frac": round(achieved_tf * products / peak, 4),
                # `peak` is the dense MFMA rate at the 2.4 GHz peak engine clock; the clock the power management held while THIS kernel ran is
                # measured (s_memtime / s_memrealtime of one wave per launch, same one-stream loop as avg_launch_ms), and the fraction of the
                # matrix rate at that clock is reported beside `frac`, never instead of it
                "shader_clock_ghz": clock_ghz.get("conv"), "peak_clock_ghz": PEAK_ENGINE_CLOCK_GHZ,
                "frac_at_measured_clock": round(achieved_tf / (peak * clock_ghz["conv"] / PEAK_ENGINE_CLOCK_GHZ), 4) if clock_ghz.get("conv") else None,
                "pipe_frac_at_measured_clock": round(achieved_tf * products / (peak * clock_ghz["conv"] / PEAK_ENGINE_CLOCK_GHZ), 4) if clock_ghz.get("conv") else None,
                "shader_clock_ghz_other_kernels": {k: v for k, v in clock_ghz.items() if k != "conv"},
                "precision": prec}
    if up_tf is not None:   # second kernel family, reported beside the dominant one (its time includes the fused FIR/activation)
        roofline["upconv_fir_f16x3_kernel"] = {"launches_per_frame": 2, "avg_launch_ms": round(ums.value / max(1, ucnt.value), 4),
                                               "achieved": round(up_tf, 2), "frac": round(up_tf / peak, 4),
                                               "mfma_products_per_mac": 3 if prec != "f16mx" else 2.1, "pipe_frac": round(up_tf * (3 if prec != "f16mx" else 2.1) / peak, 4),
                                               # (f16mx: block1.conv0, 94 % of the family's FLOPs, runs 9 f16 + 5 fp8 K=64 MFMAs per stage = 19 f16-rate passes for 9 taps)
                                               # (the two launches of a frame are two instantiations since block1's takes fp8 records: mean over the instantiations seen)
                                               "traffic": (lambda u: int(sum(u) / len(u)) if u else None)([v for k, v in (traffic_all or {}).items() if "upconv_fir" in k]),
                                               # block0.conv0: 32 ch x 128^2 in, 256 ch x 256^2 out; block1.conv0: 256 ch x 256^2 in, 128 ch x 512^2 out; + weights
                                               "algorithmic_bytes_per_launch": (32 * 128 * 128 * 4 + 256 * 256 * 256 * 4 + 9 * 32 * 256 * 4 + 256 * 256 * 256 * 4 + 128 * 512 * 512 * 4 + 9 * 256 * 128 * 4) // 2}
        if traffic_all:
            roofline["traffic_all_kernels"] = {k: v for k, v in sorted(traffic_all.items(), key=lambda kv: -kv[1])[:6]}

    # ---- the same K frames on ONE stream (no frame pipelining), so that the gain of the multi-stream issue is visible ----
    single_stream_fps = None
    if args.streams > 1 and args.clip == 0:
        best = 1e9
        for rep in range(2):      # the first pass after the multi-stream phase re-warms the default stream's allocator pool: best of 2
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(K):
                clip.render_u8(rank * K + i, out=ring[i:i + 1])
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t1)
        single_stream_fps = K / best

    out = None
    if rank == 0:
        fps = total_frames / elapsed
        out = {"metric": "rendered frames/sec @ 512x512, 48 depth samples", "value": round(fps, 2), "unit": "frames/s",
               "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(elapsed / K * 1e3, 4),
               "higher_is_better": True, "scaling": "strong" if args.clip > 0 else "weak", "vs_baseline": None,
               "dtype": {"f32": "f32",
                         "f16x3": "f32 (f16x3 split: fp32 operands as two fp16 terms, 3 MFMA products per MAC, fp32 accumulate)",
                         "f16mx": "f32 operands as fp16 hi + lo, fp32 accumulate (SR precision 'f16mx' = the library default, superresolution.DEFAULT_SR_PRECISION; the fp32-class "
                                  "f16x3 is `alt_f16x3`): in block0.conv1, block1.conv0 and block1.conv1 hi*hi runs on the f16 MFMA and the two cross products "
                                  "on the block-scaled 8-bit MFMA, e5m2 activation x e4m3 weight records (<= 5e-5 * max|ref| on every reference golden, 3.6e-5 on "
                                  "the benchmarked frame vs the oracle; heavy-tail sweeps, spikes of 2^6 / 2^10 / 2^14 sigma: <= 6.3e-5 near field, 2e-5 far field "
                                  "against the 2e-4 of SURVEY 8(d), tests/test_gpu_pinned_config.py); block0.conv0 and the renderer: f16x3"}[prec],
               "data": "synthetic",
               "config": {"workload": "ref_frame_512: TriPlaneGenerator.synthesis path, 1 frame/step/GPU: "
                                      "planes cano+residual [1,3,32,256,256] -> 128^2 rays x (48 coarse + 48 importance) "
                                      "-> SuperresolutionHybrid8XDC -> 512^2 uint8; clip gathered to rank 0",
                          "neural_rendering_resolution": 128, "depth_samples": "48+48", "final_resolution": 512,
                          "frames_total": total_frames, "streams_per_gpu": args.streams, "parallelism": "frame-sharded dp%d + gather" % world,
                          "device_warmup": "%d untimed frames (%.0f ms) rendered before the %d warm-up steps: the GPU reaches its sustained clocks only after "
                                           "~20 ms of load and drops them within 50 ms of idle (scripts/gpu_warm_probe.py, profiles/r02/warm_probe.txt); "
                                           "--device-warmup-ms 0 gives the cold-start rate" % (warm_frames, args.device_warmup_ms, W),
                          "clip_constants": "weight prepack and the SR style / demodulation vectors (functions of ws = ones and the parameters only, "
                                            "triplane.py:131-132) are computed once per clip; every per-frame input (planes = cano + residual_t, camera, "
                                            "sampling noise) is processed inside the timed region"},
               "roofline": roofline}
        rep_fps = sorted(total_frames / t for t in rep_s)
        out["repeats"] = {"n": len(rep_s), "what": "the timed region run 5 times back to back (the first one is `value`)",
                          "fps": [round(total_frames / t, 1) for t in rep_s], "median": round(rep_fps[len(rep_fps) // 2], 1),
                          "min": round(rep_fps[0], 1), "max": round(rep_fps[-1], 1)}
        if single_stream_fps is not None:
            out["value_single_stream"] = round(single_stream_fps, 2)
            out["stream_pipelining_gain"] = round(fps / world / single_stream_fps, 4)

    # ---- per-family breakdown + the literal 512^2 neural render (untimed extras, rank 0 of a 1-GPU run) -------
    if rank == 0 and world == 1 and not args.no_extras:
        lib.r3d_profile_configure(0x7F); lib.r3d_profile_reset()
        nb = 10
        for i in range(nb):
            clip.render_u8(i % K, out=ring[(i % K):(i % K) + 1])
        torch.cuda.synchronize()
        names = ["render", "conv_mfma", "upconv_fir", "torgb", "sr_pack", "layout", "misc"]
        bd = {}
        for j, nme in enumerate(names):
            _lib.check(lib.r3d_profile_read(j, ctypes.byref(ms), ctypes.byref(cnt)), "profile_read")
            bd[nme] = round(ms.value / nb, 4)
        lib.r3d_profile_configure(0)
        out["breakdown_ms_per_frame"] = bd
        # renderer-only roofline (gather-bound view): algorithmic touched bytes S*1536 + outputs
        S = 128 * 128 * 96
        out["render_kernel"] = {"ms": bd["render"], "algorithmic_GBps": round((S * 1536 + 128 * 128 * 137) / (bd["render"] * 1e-3) / 1e9, 1),
                                "mlp_TFLOPs": round(S * 8320 / (bd["render"] * 1e-3) / 1e12, 2)}
        # what the ray kernel is bound by: `algorithmic_GBps` counts every tap a sample touches and is served by L1 / L2 -- not a
        # roofline.  The two real ones: (i) the VALU issue port -- wave-level VALU instructions of one launch (SQ_INSTS_VALU, a --pmc child pass of
        # this run) / 1 024 SIMDs x 4 cycles per wave64 instruction at the clock the kernel ran at; (ii) HBM on the COMPULSORY bytes of a frame:
        # planes 3 x 256^2 x 32 x 4 B read once + the feature / depth / weight images and the SPLIT copy written once.
        rk = out["render_kernel"]
        compulsory = 3 * 256 * 256 * 32 * 4 + 128 * 128 * (32 * 4 + 4 + 4 + 1) + 128 * 128 * 32 * 4
        rk["compulsory_bytes"] = compulsory
        rk["frac_of_hbm_on_compulsory_bytes"] = round(compulsory / (bd["render"] * 1e-3) / 8e12, 4)
        vi = [v for k, v in VALU_INSTS.items() if k.startswith("render_kernel")]
        if vi and clock_ghz.get("render_kernel"):
            floor_ms = vi[0] / 1024.0 * 4.0 / (clock_ghz["render_kernel"] * 1e9) * 1e3
            rk.update({"valu_wave_insts_per_launch": int(vi[0]), "shader_clock_ghz": clock_ghz["render_kernel"],
                       "valu_issue_floor_ms": round(floor_ms, 4), "frac_of_valu_issue_bound": round(floor_ms / bd["render"], 4),
                       "bound": "VALU issue (2 waves per SIMD; the rest is gather latency and the transcendental / MFMA dependency chains of a tile)"})
        # literal "512x512 neural render, 48 depth samples": R=512 rays, no SR
        cano, residuals, cams = scene
        opts = dict(G.rendering_kwargs)
        o, d = G.ray_sampler(cams[:1, :16].view(-1, 4, 4), cams[:1, 16:25].view(-1, 3, 3), 512)
        planes = clip.planes_for(0)
        alt = {}
        for nf in (0, 48):
            opts["depth_resolution_importance"] = nf
            for _ in range(2):
                G.renderer(planes, G.decoder, o, d, opts)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                G.renderer(planes, G.decoder, o, d, opts)
            torch.cuda.synchronize()
            alt["R512_48+%d_fps" % nf] = round(5 / (time.perf_counter() - t1), 2)
        out["alt_neural_render_512"] = alt

        # ---- the same frames through the drop-in API: TriPlaneGenerator.synthesis(ws, camera, use_cached_backbone=True) on a generator
        # whose operators were installed by patch_model() -- what inference/real3d_infer.py:486-492 gets per frame: planes handed over
        # NCHW as `cano + secc` (secc_img2plane.py:76-77), depth image + global clamp, image_raw, image_feature, the fp32 image, the dict.
        from real3dportrait_amd import patch_model
        from real3dportrait_amd.frames import clone_generator_shell, frame_seed
        ws_api = torch.ones(1, 14, 512, device=dev)
        ref_u8 = clip.render_u8(3).clone()
        api = {}
        for api_prec in ([prec] + [p for p in ("f16mx", "f16x3") if p != prec]):      # the drop-in contract at BOTH shipped precisions
            G_api = patch_model(clone_generator_shell(G), precision=api_prec)
            G_api.renderer.noise_mode = "hash"

            def api_frame(t):
                G_api.renderer.seed = frame_seed(clip.base_seed, t)
                G_api._last_planes = (cano + residuals[t % len(residuals)]).view(1, 96, 256, 256)
                return G_api.synthesis(ws_api, cams[t:t + 1], use_cached_backbone=True, noise_mode="none")
            for i in range(12):                          # (first calls of a precision: weight prepack, style vectors, workspaces)
                ret = api_frame(i)
            torch.cuda.synchronize()
            assert tuple(ret["image"].shape) == (1, 3, 512, 512) and tuple(ret["image_raw"].shape) == (1, 3, 128, 128) \
                and tuple(ret["image_depth"].shape) == (1, 1, 128, 128) and tuple(ret["image_feature"].shape) == (1, 29, 128, 128)
            api_u8 = ((ret["image"][0].permute(1, 2, 0) + 1) / 2 * 255).int().clamp(0, 255).to(torch.uint8)
            api_equal = bool(torch.equal(api_u8, ref_u8)) if api_prec == prec else None
            nb, best_api = 40, 1e9
            for rep in range(2):                         # best of 2 passes of 40 frames
                t1 = time.perf_counter()
                for i in range(nb):
                    ret = api_frame(i)
                torch.cuda.synchronize()
                best_api = min(best_api, (time.perf_counter() - t1) / nb)
            api[api_prec] = (best_api, api_equal)
        t_api, api_equal = api[prec]
        out["value_synthesis_api"] = {
            "what": "TriPlaneGenerator.synthesis() per frame on ONE stream through patch_model()'d operators: `cano + secc` add (torch), layout, rays, "
                    "fused ray kernel WITH the depth image + clamp, SR with the fp32 image, clamps, the reference's output dict",
            "precision": prec,
            "value": round(1.0 / t_api, 2), "ms_per_frame": round(t_api * 1e3, 4), "vs_value_single_stream": round(1.0 / t_api / single_stream_fps, 4) if single_stream_fps else None,
            "frame_equals_uint8_ring_frame": api_equal,
            "other_precisions": {p: {"value": round(1.0 / t, 2), "ms_per_frame": round(t * 1e3, 4)} for p, (t, _) in api.items() if p != prec}}

        # the same W + K measurement from an idle (cold-clock) GPU, i.e. without the device warm-up: what the first K frames after a pause cost
        if pipe is not None and args.device_warmup_ms > 0:
            torch.cuda.synchronize(); time.sleep(0.25)
            for i in range(max(W, args.streams)):
                step(i % K)
            pipe.sync(); torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(K):
                step(i)
            pipe.sync(); torch.cuda.synchronize()
            out["value_cold_start"] = round(K / (time.perf_counter() - t1), 2)
            out["value_r01_protocol"] = {"value": out["value_cold_start"], "note": "no device warm-up in front of the W + K steps: the protocol of the round-1 "
                                         "number (1 052); compare rounds on this field, `value` carries the warm-up described in config.device_warmup"}

    # ---- BASELINE config 4 surrogate: the per-frame hot path of the shipped torso model (extra, rank 0 of a 1-GPU run) ----
    if rank == 0 and world == 1 and not args.no_extras:
        frame, tf_flops = build_torso_frame(torch, dev, G)
        for i in range(3):
            frame(i)
        torch.cuda.synchronize()
        nb = 10
        t1 = time.perf_counter()
        for i in range(nb):
            frame(i)
        torch.cuda.synchronize()
        t_frame = (time.perf_counter() - t1) / nb
        # per-kernel-family breakdown in a SEPARATE pass: the event pairs around every launch cost ~7 us each on this 45-launch frame
        # (3.02 ms with them, 2.70 ms without -- the figure rocprofv3 sees, profiles/r03/torso_kernel_stats.txt)
        lib.r3d_profile_configure(0x7F); lib.r3d_profile_reset()
        for i in range(nb):
            frame(i)
        torch.cuda.synchronize()
        bd2 = {}
        for j, nme in enumerate(["render", "conv_mfma", "upconv_fir", "torgb", "sr_pack", "layout", "misc"]):
            _lib.check(lib.r3d_profile_read(j, ctypes.byref(ms), ctypes.byref(cnt)), "profile_read")
            bd2[nme] = round(ms.value / nb, 4)
        lib.r3d_profile_configure(0)
        # the same frames round-robin on 3 streams (own module shells, as the head path's PipelinedClipRenderer): the under-filled launches of
        # the small layers (to_plane_cnn at 128^2 = 128 blocks for 512 block slots) and the kernel tails fill with other frames' work
        from real3dportrait_amd.frames import StreamPipeline, clone_generator_shell
        pipe3 = StreamPipeline([frame] + [build_torso_frame(torch, dev, clone_generator_shell(G))[0] for _ in range(2)])
        for i in range(6):
            pipe3.submit(i)
        pipe3.sync(); torch.cuda.synchronize()
        nb3 = 18
        t1 = time.perf_counter()
        for i in range(nb3):
            pipe3.submit(i)
        pipe3.sync(); torch.cuda.synchronize()
        t_frame3 = (time.perf_counter() - t1) / nb3
        del pipe3
        conv_ms = bd2["conv_mfma"] + bd2["upconv_fir"]
        out["torso_frame"] = {"what": "to_plane_cnn -> planes -> 128^2 rays x (48+48) -> fused SuperresolutionHybrid8XDC_Warp.forward (fuse mode v2) "
                                      "-> 512^2; cold encoders and the face-vid2vid warp net replaced by synthetic outputs",
                              "ms_per_frame": round(t_frame * 1e3, 4), "fps": round(1.0 / t_frame, 2),
                              "fps_3_streams": round(1.0 / t_frame3, 2),
                              "breakdown_ms_per_frame": bd2, "conv_gflop_per_frame": round(tf_flops / 1e9, 1),
                              "roofline": {"bound": "mfma", "achieved": round(tf_flops / (conv_ms * 1e-3) / 1e12, 2), "peak": PEAK_F16_MFMA_TFLOPS,
                                           "unit": "TFLOP/s", "frac": round(tf_flops / (conv_ms * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS, 4),
                                           "note": "all conv kernels of the frame (algorithmic FLOPs / summed conv time)"}}

    # ---- what the power management reports while the headline loop runs (extras; ~3 s of the same pipelined frames) ----------------------
    if rank == 0 and world == 1 and not args.no_extras:
        out["power"] = power_probe(torch, lambda i: step(i % K), (lambda: pipe.sync()) if pipe is not None else torch.cuda.synchronize, K)

    # ---- the sustained rate of the dominant kernel's MFMA mix on THIS board with nothing else busy (scripts/probes/power_ceiling_probe.hip) ----
    if rank == 0 and world == 1 and not args.no_extras:
        out["mfma_mix_sustained"] = sustained_mix_probe(achieved_tf if prec == "f16mx" else None)

    # ---- BASELINE configs[2] on one GPU and configs[4] (stress) -- extras, rank 0 of a 1-GPU run ----------------------------------------
    if rank == 0 and world == 1 and not args.no_extras:
        out["clip125_1gpu"] = clip125(torch, dev, G, scene, clip, args.streams)
        out["cfg5_stress"] = cfg5_stress(torch, dev, lib, prec)

    # ---- the other shipped SR precision on the same frames (default f16mx -> f16x3, the fp32-class tier; and vice versa) ----------------
    if rank == 0 and world == 1 and not args.no_extras and prec in ("f16x3", "f16mx"):
        from real3dportrait_amd.frames import PipelinedClipRenderer
        other = "f16x3" if prec == "f16mx" else "f16mx"
        cano, residuals, cams = scene
        for b in (G.superresolution.block0, G.superresolution.block1):
            b.precision = other
        pipe2 = PipelinedClipRenderer(G, cano, residuals, cams, clip.ws, base_seed=clip.base_seed, n_streams=max(1, args.streams))
        for i in range(2 * max(1, args.streams)):
            pipe2.render_u8(i % K, out=ring[(i % K):(i % K) + 1])
        pipe2.sync(); torch.cuda.synchronize()
        t_w = time.perf_counter()                          # same device warm-up as the headline measurement
        while (time.perf_counter() - t_w) * 1e3 < args.device_warmup_ms:
            for i in range(8):
                pipe2.render_u8(i % K, out=ring[(i % K):(i % K) + 1])
            pipe2.sync(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(K):
            pipe2.render_u8(i, out=ring[i:i + 1])
        pipe2.sync(); torch.cuda.synchronize()
        t_o = (time.perf_counter() - t1) / K
        lib.r3d_profile_configure(1 << 1); lib.r3d_profile_reset()
        for i in range(10):
            clip.render_u8(i % K, out=ring[(i % K):(i % K) + 1])
        torch.cuda.synchronize()
        _lib.check(lib.r3d_profile_read(1, ctypes.byref(ms), ctypes.byref(cnt)), "profile_read")
        lib.r3d_profile_configure(0)
        o_ms = ms.value / max(1, cnt.value)
        what = {"f16x3": "SR precision 'f16x3' (R3D_SR_PRECISION=f16x3): every product as 3 fp16 MFMA terms, fp32-class (<= 1.3e-6 over the operand sweeps, "
                         "tests/test_gpu_range_and_sizes.py; tests/test_gpu_f16x3.py re-runs the default-precision tests on it); selected by name",
                "f16mx": "SR precision 'f16mx' (the library default): cross products on the block-scaled 8-bit MFMA; parity tier 5e-5 * max|ref| (tests/test_gpu_mx.py)"}[other]
        out["alt_" + other] = {"what": what, "value": round(1.0 / t_o, 2), "ms_per_step": round(t_o * 1e3, 4), "conv_avg_launch_ms": round(o_ms, 4),
                               "conv_algorithmic_tflops": round((flops[1] + flops[3]) / 2 / (o_ms * 1e-3) / 1e12, 1),
                               "conv_frac_of_f16_peak": round((flops[1] + flops[3]) / 2 / (o_ms * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS, 4)}
        for b in (G.superresolution.block0, G.superresolution.block1):
            b.precision = prec

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        ref = cpu_baseline_reference()
        if ref is not None:
            out["cpu_baseline_reference"] = ref
    if rank == 0:
        # the driver keeps `roofline` and `config` verbatim and drops the other extras, so the per-kernel record of ITS box rides inside them
        rf = out["roofline"]
        for k in ("breakdown_ms_per_frame", "render_kernel", "value_single_stream", "stream_pipelining_gain", "value_cold_start"):
            if k in out:
                rf[k] = out[k]
        if "value_synthesis_api" in out:
            api_o = out["value_synthesis_api"]
            rf["value_synthesis_api"] = dict({api_o["precision"]: api_o["value"]}, **{p_: v_["value"] for p_, v_ in api_o.get("other_precisions", {}).items()})
        if "alt_f16x3" in out:
            rf["alt_f16x3"] = out["alt_f16x3"].get("value") if isinstance(out["alt_f16x3"], dict) else out["alt_f16x3"]
        if "repeats" in out:
            out["config"]["value_is"] = "the FIRST of five back-to-back timed regions (the driver's protocol); their median: %.1f frames/s (min %.1f, max %.1f)" % (
                out["repeats"]["median"], out["repeats"]["min"], out["repeats"]["max"])
            out["config"]["value_median_of_5"] = out["repeats"]["median"]
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
