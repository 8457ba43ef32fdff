#!/usr/bin/env python3
"""Time the REFERENCE's own PyTorch CPU renderer + super-resolution on this machine's host cores (north_star: "next to the
reference CPU renderer").

Runs only where /root/reference exists (the build container; the GPU box has no reference checkout), so its JSON output is
committed (profiles/cpu_reference_r02.json) and bench.py reports it as `cpu_baseline_reference`, stating where / when / on how many
cores it was measured.  Workload = one REF frame, the same one bench.py renders: ImportanceRenderer.forward at R=128 with 48 coarse +
48 importance samples on [1,3,32,256,256] planes (modules/eg3ds/volumetric_rendering/renderer.py:118-167) followed by
SuperresolutionHybrid8XDC.forward 128^2 -> 512^2 (modules/eg3ds/models/superresolution.py:348-359); sampling noise injected as in
tests/golden/make_golden.py; torch.set_num_threads(all cores); warm-up 1, best of 3.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/scripts/time_reference_cpu.py > /root/repo/profiles/cpu_reference_r02.json
"""
import json
import os
import platform
import sys
import time

REF = os.environ.get("R3D_REFERENCE", "/root/reference")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests", "golden"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import make_golden as mg  # noqa: E402  (reference imports + noise injection + weight loading helpers)
from real3dportrait_amd import synth  # noqa: E402


def best_of(fn, n=3):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts), ts


def main():
    cores = os.cpu_count()
    torch.set_num_threads(cores)
    seed, R, Nc, Nf = 7, 128, 48, 48
    planes = torch.from_numpy(synth.synth_planes(seed, N=1) + synth.synth_planes(seed + 1, N=1, scale=0.1))
    dec = mg.make_decoder(synth.synth_decoder(seed, sigma_bias=4.0))
    cam = torch.from_numpy(synth.camera_sweep(64, -0.4, 0.4)[:1])
    noise_c = synth.synth_noise(seed, (1, R * R, Nc, 1))
    u_f = synth.synth_noise(seed + 1, (R * R, Nf))
    ren = mg.ImportanceRenderer(hp={"enable_rescale_plane_regulation": False, "triplane_feature_type": "triplane"}).eval()
    sampler = mg.RaySampler()
    sr = mg.SuperresolutionHybrid8XDC(channels=32, img_resolution=512, sr_num_fp16_res=0, sr_antialias=True, channel_base=32768,
                                      channel_max=512, fused_modconv_default="inference_only").eval()
    for blk, p in zip((sr.block0, sr.block1), synth.synth_sr_params(seed)):
        mg.load_block(blk, p)
    ws = torch.ones(1, 14, 512)
    state = {}

    def render():
        with torch.no_grad(), mg.injected_noise(noise_c, u_f):
            o, d = sampler(cam[:, :16].view(-1, 4, 4), cam[:, 16:].view(-1, 3, 3), R)
            state["out"] = ren(planes, dec, o, d, mg.opts(Nc, Nf))

    def superres():
        feat = state["out"][0].permute(0, 2, 1).reshape(1, 32, R, R).contiguous()
        with torch.no_grad():
            state["img"] = sr(feat[:, :3], feat, ws, noise_mode="none")

    t_render, all_r = best_of(render)
    t_sr, all_s = best_of(superres)
    out = {
        "what": "reference PyTorch CPU path (modules/eg3ds ImportanceRenderer.forward + SuperresolutionHybrid8XDC.forward), 1 REF frame",
        "workload": "R=128, 48+48 samples, planes [1,3,32,256,256], SR 128^2 -> 512^2, fp32",
        "value": 1.0 / (t_render + t_sr), "unit": "frames/s", "cores": cores, "torch_threads": torch.get_num_threads(),
        "render_s": t_render, "sr_s": t_sr, "render_runs_s": all_r, "sr_runs_s": all_s,
        "torch": torch.__version__, "machine": platform.processor() or platform.machine(),
        "where": "host %s" % platform.node(), "when": time.strftime("%Y-%m-%d"),
        "kind": "reference",
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
