"""bench.py's modes as the driver launches them, on one GPU: the default weak-scaling step, the multi-GPU step (RCCL gather of
the packed records on a side stream, here with world_size 1 under torchrun) and the strong-scaling job of BASELINE config[4]
(--global-batch 256).  Each run must end with ONE JSON line whose frame-0 boxes equal the reference's golden ones."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--steps", "8", "--warmup", "4", "--no-cpu-baseline", "--no-kernel-roofline", "--no-extras"]


def run(cmd, port=None):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if port:
        env["MASTER_PORT"] = str(port)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def check_line(d, scaling, frames):
    assert d["unit"] == "frames/s" and d["value"] > 0 and d["n_gpus"] == 1 and d["steps"] == 8 and d["warmup"] == 4
    assert d["scaling"] == scaling and d["dtype"] == "f32" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["config"]["global_batch"] == frames
    assert d["config"]["boxes_match_reference_golden_frame0"] is True
    assert d["config"]["graph_captures_per_executor"] == 1
    assert abs(d["value"] - frames * 8 / (d["ms_per_step"] * 8e-3)) / d["value"] < 1e-3


@pytest.mark.gpu
def test_bench_default_step():
    check_line(run([sys.executable, "bench.py"] + COMMON), "weak", 64)


@pytest.mark.gpu
def test_bench_gather_step_under_torchrun():
    d = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
             "--master-port", "29541", "bench.py", "--gpus", "1", "--force-gather"] + COMMON)
    check_line(d, "weak", 64)
    assert "RCCL gather" in d["config"]["gather"]


@pytest.mark.gpu
def test_bench_strong_scaling_job():
    d = run([sys.executable, "bench.py", "--force-gather", "--global-batch", "256"] + COMMON, port=29543)
    check_line(d, "strong", 256)
    assert d["config"]["frames_per_gpu"] == 256
