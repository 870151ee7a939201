"""The suite's own safety net (tests/conftest.py): a test marked `isolated` runs in a child interpreter, so a runtime abort() inside
it is ONE red test with the child's output attached and the session goes on; and the collection order puts the BASELINE-config and
golden-fixture parity tests in front of everything else that needs the GPU, stress tests last."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.isolated
def test_child_target():
    if os.environ.get("FFCNN_TEST_ABORT") == "1":
        os.abort()


def test_after_child_target():
    pass


def test_abort_in_child_is_one_red_test():
    env = dict(os.environ, FFCNN_TEST_ABORT="1")
    env.pop("FFCNN_TEST_IN_CHILD", None)
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_isolation.py", "-q", "-p", "no:cacheprovider",
                        "-k", "child_target"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    assert r.returncode == 1, out[-2000:]
    assert "1 failed, 1 passed" in out and "signal 6" in out, out[-2000:]


def test_gpu_collection_order():
    r = subprocess.run([sys.executable, "-m", "pytest", "tests", "--collect-only", "-q", "-m", "gpu", "-p", "no:cacheprovider"],
                       cwd=ROOT, capture_output=True, text=True, timeout=300)
    ids = [ln for ln in r.stdout.splitlines() if "::" in ln]
    assert len(ids) > 300
    assert "test_dw_config1_full_batch_sampled" in ids[0] and "test_pw_config2_against_oracle" in ids[1]
    first80 = " ".join(ids[:80])
    for name in ("test_net_api_single_frame", "test_batch64_plans", "test_groupconv_dropin_golden", "test_golden_layer_samples",
                 "test_every_layer_keep_all", "test_big_batch_plans_activations"):
        assert name in first80, name
    assert "test_no_device_or_host_memory_leak" in ids[-1]
    stress = [k for k, i in enumerate(ids) if "lifecycle_walk" in i or "node_walk" in i or "memory_leak" in i]
    assert min(stress) >= len(ids) - 9, "stress tests must be the tail of the collection"
