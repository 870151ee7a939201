"""bench.py's launch decision (`launch_plan`): `--gpus N` runs on N GPUs however it is started, or is refused -- never a 1-GPU run
labelled N, never an N-GPU label on one device.  Pure host logic: no GPU, no torch import."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_launch_plan():
    lp = _bench().launch_plan
    assert lp(1, {}, 1) == ("inline", None)
    assert lp(1, {}, 8) == ("inline", None)
    assert lp(1, {}, 0)[0] == "refuse"
    # the driver's N = 1 command shape with N > 1 and no launcher: become the launcher (or the C node path on request)
    assert lp(8, {}, 8) == ("torchrun", None)
    assert lp(2, {}, 8) == ("torchrun", None)
    assert lp(8, {"FFCNN_BENCH_MULTI": "node"}, 8) == ("node", None)
    how, why = lp(8, {}, 1)
    assert how == "refuse" and "8" in why and "1 HIP device" in why
    assert lp(2, {}, 1)[0] == "refuse" and lp(8, {}, 4)[0] == "refuse" and lp(0, {}, 8)[0] == "refuse"
    # under torchrun: the world size is the GPU count of the line, anything else is refused
    assert lp(8, {"WORLD_SIZE": "8", "LOCAL_WORLD_SIZE": "8"}, 8) == ("inline", None)
    assert lp(1, {"WORLD_SIZE": "1"}, 1) == ("inline", None)
    assert lp(8, {"WORLD_SIZE": "1"}, 8)[0] == "refuse"
    assert lp(1, {"WORLD_SIZE": "8"}, 8)[0] == "refuse"
    assert lp(4, {"WORLD_SIZE": "8"}, 8)[0] == "refuse"
    assert lp(8, {"WORLD_SIZE": "8"}, 4)[0] == "refuse"
    assert lp(8, {"WORLD_SIZE": "x"}, 8)[0] == "refuse"
    # --node: one process for all devices
    assert lp(8, {}, 8, node=True) == ("node", None)
    assert lp(1, {}, 1, node=True) == ("node", None)
    assert lp(8, {}, 2, node=True)[0] == "refuse"
    assert lp(2, {"WORLD_SIZE": "2"}, 2, node=True)[0] == "refuse"


def test_node_budget_counts_the_executors():
    b = _bench()
    # 8 devices x 4 slots = 32 executors must fit with margin; the one-device child (8 slots) stays a short limit
    assert b.node_budget_s(8, 4) >= 3 * (0.35 * 32 + 8) and b.node_budget_s(8, 4) <= 300
    assert b.node_budget_s(1, 8) <= 150
    assert b.node_budget_s(8, 8) > b.node_budget_s(8, 4) > b.node_budget_s(2, 4)


def test_host_topology_counts_physical_cores():
    b = _bench()
    threads, reps = b.host_topology()
    assert len(reps) >= 1 and len(reps) <= len(threads) and set(reps) <= set(threads)
    # one representative per distinct sibling list
    sib = set()
    for c in reps:
        try:
            sib.add(open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read().strip())
        except OSError:
            sib.add(c)
    assert len(sib) == len(reps)
