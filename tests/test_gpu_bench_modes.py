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
NO_NODE = ["--no-node-line"]


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
    assert d["config"]["graph_captures_per_executor"] <= 2       # the executor's graph + the u8 form's; bench.py itself asserts that none is captured after the warm-up
    assert abs(d["value"] - frames * 8 / (d["ms_per_step"] * 8e-3)) / d["value"] < 1e-3


@pytest.mark.gpu
def test_bench_default_step():
    d = run([sys.executable, "bench.py"] + COMMON)
    check_line(d, "weak", 64)
    assert d["config"]["untimed_forwards_before_t0"] > 0
    # beside it: the same job through the C node API (a child `bench.py --node`), same metric, boxes checked there as well
    assert "strong_b256" not in d                                # (--no-extras; the driver's default N = 1 line carries it: test_bench_default_line_has_strong_b256)
    c = d["c_node_api"]
    assert "error" not in c, c
    assert c["unit"] == "frames/s" and c["value"] > 0 and c["n_gpus"] == 1 and "C node API" in c["host"]
    assert c["config"]["boxes_match_reference_golden_frame0"] is True


@pytest.mark.gpu
def test_bench_node_mode():
    """bench.py --node: one process, ffgpu_node_create / submit / wait (the C host path of north_star), same JSON contract"""
    d = run([sys.executable, "bench.py", "--node", "--steps", "8", "--warmup", "4", "--depth", "4"])
    assert d["unit"] == "frames/s" and d["value"] > 0 and d["n_gpus"] == 1 and d["steps"] == 8 and d["warmup"] == 4
    assert d["scaling"] == "weak" and d["dtype"] == "f32" and d["config"]["global_batch"] == 64
    assert d["config"]["boxes_match_reference_golden_frame0"] is True
    assert abs(d["value"] - 64 * 8 / (d["ms_per_step"] * 8e-3)) / d["value"] < 1e-3
    d = run([sys.executable, "bench.py", "--node", "--steps", "4", "--warmup", "2", "--global-batch", "256", "--depth", "2"])
    assert d["scaling"] == "strong" and d["config"]["global_batch"] == 256 and d["config"]["boxes_match_reference_golden_frame0"] is True


@pytest.mark.gpu
def test_bench_gather_step_under_torchrun():
    d = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
             "--master-port", "29541", "bench.py", "--gpus", "1", "--force-gather"] + COMMON + NO_NODE)
    check_line(d, "weak", 64)
    assert "RCCL gather" in d["config"]["gather"]
    # one command, both scaling answers: the weak-scaling line carries north_star's batch-256 job (merged and one step per launch) as an extra,
    # with the size of the communicator it ran on
    sb = d["strong_b256"]
    assert sb["global_batch"] == 256 and sb["frames_per_gpu"] == 256 and sb["n_gpus"] == 1 and sb["rccl_ranks"] == 1 and d["rccl_ranks"] == 1
    for key in ("merged", "unmerged"):
        assert sb[key]["value"] > 0 and sb[key]["steps_per_launch"] >= 1 and abs(sb[key]["value"] - 256 * 8 / (sb[key]["ms_per_step"] * 8e-3)) / sb[key]["value"] < 1e-3
    assert sb["unmerged"]["steps_per_launch"] == 1 and sb["merged"]["frames_per_launch"] == 256


@pytest.mark.gpu
def test_bench_strong_scaling_job():
    d = run([sys.executable, "bench.py", "--force-gather", "--global-batch", "256"] + COMMON + NO_NODE, port=29543)
    check_line(d, "strong", 256)
    assert d["config"]["frames_per_gpu"] == 256 and "strong_b256" not in d      # (the extra belongs to the weak-scaling line)


@pytest.mark.gpu
def test_bench_refuses_more_gpus_than_visible():
    """`bench.py --gpus N` uses N GPUs however it is launched, or fails loudly: on a box with fewer devices the plain command, the
    --node form and a torchrun job whose world size differs from --gpus all exit non-zero and print no JSON line."""
    import torch
    n = torch.cuda.device_count() + 1
    for cmd in ([sys.executable, "bench.py", "--gpus", str(n)] + COMMON,
                [sys.executable, "bench.py", "--node", "--gpus", str(n), "--steps", "4", "--warmup", "2"]):
        r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=300)
        assert r.returncode != 0, r.stdout[-500:]
        assert "HIP device" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], r.stderr[-500:]
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "8"] + COMMON, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr, r.stderr[-500:]


@pytest.mark.gpu
def test_bench_strong_scaling_reports_both_schedules():
    """strong scaling with a shard of 32 frames (the per-GPU load of 256 frames over 8 GPUs): the headline merges 4 steps per launch,
    the one-step-per-launch figure stands beside it; the line names the size of the RCCL communicator it created"""
    d = run([sys.executable, "bench.py", "--force-gather", "--global-batch", "32"] + COMMON + NO_NODE, port=29549)
    check_line(d, "strong", 32)
    assert d["config"]["steps_per_launch"] == 4 and d["config"]["frames_per_launch"] == 128
    u = d["strong_unmerged"]
    assert u["steps_per_launch"] == 1 and u["frames_per_launch"] == 32 and u["value"] > 0
    assert d["rccl_ranks"] == 1                                  # --force-gather: a one-rank communicator really exists


@pytest.mark.gpu
def test_bench_default_line_has_strong_b256():
    """the N = 1 line without --no-extras: the batch-256 job (the denominator of north_star's 8-vs-1 sentence) sits beside the value, as it does at N > 1"""
    d = run([sys.executable, "bench.py", "--steps", "8", "--warmup", "4", "--no-cpu-baseline", "--no-kernel-roofline", "--no-node-line"])
    check_line(d, "weak", 64)
    sb = d["strong_b256"]
    assert sb["global_batch"] == 256 and sb["frames_per_gpu"] == 256 and sb["n_gpus"] == 1 and sb["merged"]["value"] > 0 and sb["merged"]["frames_per_launch"] == 256
