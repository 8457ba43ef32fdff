"""GPU: the commands the driver will run.  `bench.py` under torch.distributed.run with ONE rank exercises everything the N-rank scaling
run does -- env rendezvous on 127.0.0.1, the RCCL process group, the r3d_comm communicator (its id shipped over the process group),
r3d_gather_frames behind the C ABI into the pre-allocated clip buffer, the uneven-tail arithmetic of --clip, MAX-over-ranks timing --
before an 8-GPU node ever sees it."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(args, torchrun):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable]
    if torchrun:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29533"]
    cmd += [os.path.join(ROOT, "bench.py")] + args
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, (out.returncode, out.stdout[-2000:], out.stderr[-2000:])
    return json.loads(lines[0])


def test_bench_weak_scaling_line_under_torchrun():
    d = _run(["--gpus", "1", "--steps", "6", "--warmup", "2", "--no-extras", "--no-cpu-baseline"], torchrun=True)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "repeats"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["config"]["frames_total"] == 6 and "workload" in d["config"]
    assert abs(d["value"] - 6 / (d["ms_per_step"] * 6e-3)) / d["value"] < 1e-3
    assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1 and d["repeats"]["n"] == 5
    assert d["value"] > 100
    # roofline.traffic is re-measured in the run (two rocprofv3 --pmc child passes) whenever the tool is there; else the committed figure, labelled
    import shutil
    r = d["roofline"]
    assert r["traffic"] and r["traffic_source"] and r["algorithmic_bytes_per_launch"] > 1e8
    if shutil.which("rocprofv3"):
        assert "child passes of this run" in r["traffic_source"], r["traffic_source"]
        assert 0.9 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.6, (r["traffic"], r["algorithmic_bytes_per_launch"])
    assert "f16mx" in d["dtype"] and r["precision"] == "f16mx"          # the precision of `value` is selected by name and spelled out


def test_bench_clip_mode_uneven_tail_under_torchrun():
    d = _run(["--gpus", "1", "--clip", "7", "--warmup", "2", "--no-extras", "--no-cpu-baseline"], torchrun=True)
    assert d["scaling"] == "strong" and d["config"]["frames_total"] == 7 and d["steps"] == 7
    assert d["value"] > 100
