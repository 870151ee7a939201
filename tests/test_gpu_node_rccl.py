"""The C node API (include/ffcnn_hip.h ffgpu_node_*) on SEVERAL devices: the RCCL branch of ffgpu_node.inc -- ncclCommInitAll,
the grouped ncclBroadcast of the weights, the grouped ncclSend / ncclRecv gather of the packed records.  These tests need >= 2
visible GPUs and skip on a one-GPU box; there the same multi-rank logic (shards, offsets, packing, unpacking, pipelining) is
covered through FFGPU_NODE_LOOPBACK (tests/test_gpu_round2.py, tests/test_gpu_fuzz_api.py), which differs from this path only
in the two RCCL calls."""
import numpy as np
import pytest

from test_gpu_parity import boxes_match
from test_gpu_round2 import F, eight, net  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu


def ndev():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def check_records(dets, runs, frames, what):
    for f, k in enumerate(frames):
        assert dets[f]["ncand"] == len(runs[k]["cand"]), "%s frame %d: ncand %d" % (what, f, dets[f]["ncand"])
        boxes_match(dets[f]["box"][:dets[f]["count"]], runs[k]["boxes"], "%s frame %d" % (what, f))


def run_node(F, net, eight, n, total, depth=1):
    fr, runs = eight
    frames = [k % 8 for k in range(total)]
    with F.Node(net, n, total, exec_flags=0, node_flags=F.Node.DEPTH(depth)) as nd:
        assert [nd.shard(r)[2] for r in range(n)] == list(range(n))          # rank r on device r
        # ranks > 0 hold a zeroed copy of the filter rows until the broadcast: had it not happened, their shards would see no boxes
        for rep in range(3):
            dets = nd.forward_host(fr[frames])
            check_records(dets, runs, frames, "%d devices, rep %d" % (n, rep))
        if depth > 1:
            tickets = [nd.submit(fr[[(k + s) % 8 for k in frames]]) for s in range(depth)]
            for s, t in enumerate(tickets):
                check_records(nd.wait(t), runs, [(k + s) % 8 for k in frames], "pipelined step %d" % s)


@pytest.mark.skipif(ndev() < 2, reason="needs >= 2 GPUs (the RCCL exchange of the C node path)")
def test_node_rccl_two_devices(F, net, eight):
    run_node(F, net, eight, 2, 8)
    run_node(F, net, eight, 2, 7, depth=3)                                  # uneven shards, three steps in flight


@pytest.mark.skipif(ndev() < 3, reason="needs >= 3 GPUs")
def test_node_rccl_all_devices(F, net, eight):
    n = ndev()
    run_node(F, net, eight, n, 4 * n, depth=2)
    run_node(F, net, eight, n, 4 * n + 3, depth=1)


@pytest.mark.skipif(ndev() < 2, reason="needs >= 2 GPUs")
def test_node_rccl_equals_loopback_bytes(F, net, eight):
    """the records the RCCL gather delivers are byte-identical to the loopback (peer copy) gather's"""
    fr, runs = eight
    with F.Node(net, 2, 8, node_flags=F.Node.LOOPBACK) as nd:
        want = nd.forward_host(fr)
    with F.Node(net, 2, 8) as nd:
        got = nd.forward_host(fr)
    assert got.tobytes() == want.tobytes()


def test_node_gated_tests_skip_cleanly_on_one_gpu():
    """bookkeeping: on the one-GPU box the three tests above are skipped, not silently absent"""
    assert ndev() >= 1
