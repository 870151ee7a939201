import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "isolated: runs in a child interpreter (stress / lifecycle tests that could take the process down)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if not os.environ.get("FFCNN_TEST_KEEP_ORDER"):            # (tools/repro_abort.sh replays round 2's alphabetical order)
        order_items(items)
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def orc():
    from oracle import orc as m
    m.build()
    return m


@pytest.fixture(scope="session")
def test_image(orc):
    return orc.load_bmp()


# ---- suite order and process isolation ---------------------------------------------------------------------------------------
# Parity against the oracle / golden fixtures comes first (the BASELINE configs at the very front), stress and lifecycle tests that
# create and destroy hundreds of executors come last and run in a child interpreter: a runtime abort() inside one of them is then one
# red test with its stderr attached, not the end of the session.
_FILE_ORDER = ["test_gpu_parity", "test_gpu_round2", "test_gpu_round3", "test_gpu_round4", "test_gpu_round5", "test_gpu_kernels", "test_gpu_bench_modes", "test_gpu_geometries",
               "test_gpu_cfg_styles", "test_gpu_load_failures", "test_gpu_hazard_controls", "test_gpu_fuzz_input", "test_gpu_fuzz_nets", "test_gpu_fuzz",
               "test_gpu_node_rccl", "test_gpu_fuzz_api"]
# the BASELINE.json configs and the golden fixtures, in this order, before everything else that needs the GPU
_FRONT = ["test_dw_config1_full_batch_sampled", "test_pw_config2_against_oracle", "test_net_api_single_frame", "test_cli_geometry_640x448",
          "test_batch64_plans", "test_groupconv_dropin_golden", "test_golden_layer_samples", "test_every_layer_keep_all",
          "test_fused_executor_materialised_layers", "test_fused_graph_executor_boxes", "test_big_batch_plans_activations"]
_CHILD_ENV = "FFCNN_TEST_IN_CHILD"


def _order_key(item):
    mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    name = item.name.split("[")[0]
    if name in _FRONT and mod.startswith("test_gpu"):
        return (-0.5, _FRONT.index(name))
    rank = _FILE_ORDER.index(mod) if mod in _FILE_ORDER else (len(_FILE_ORDER) if mod.startswith("test_gpu") else -1)
    return (rank, 1 if item.get_closest_marker("isolated") else 0)


def order_items(items):
    """stable sort: CPU tests, then the GPU files in _FILE_ORDER, the process-isolated (stress) tests of each file after its others,
    and the whole of test_gpu_fuzz_api (API walks, lifecycle walks, the leak test) at the very end"""
    items.sort(key=_order_key)


def run_isolated(nodeid, timeout=1200):
    """run one test id in a child interpreter; returns (returncode, tail of its output).  A negative return code is the signal
    that killed the child (-6 = SIGABRT)."""
    import subprocess
    env = dict(os.environ)
    env[_CHILD_ENV] = "1"
    env["PYTHONFAULTHANDLER"] = "1"
    r = subprocess.run([sys.executable, "-X", "faulthandler", "-m", "pytest", nodeid, "-x", "-q", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
    return r.returncode, r.stdout[-6000:]


@pytest.hookimpl(tryfirst=True)
def pytest_pyfunc_call(pyfuncitem):
    if pyfuncitem.get_closest_marker("isolated") is None or os.environ.get(_CHILD_ENV):
        return None                                             # the ordinary in-process call
    rc, tail = run_isolated(pyfuncitem.nodeid)
    if rc != 0:
        pytest.fail("child interpreter for %s ended with %s\n%s" % (pyfuncitem.nodeid, "signal %d" % -rc if rc < 0 else "rc %d" % rc, tail), pytrace=False)
    return True
