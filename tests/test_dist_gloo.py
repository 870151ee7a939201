"""The N > 1 path on CPU: world_size 2, gloo.  Exercises exactly what bench.py does between the
kernels -- contiguous frame shards, one broadcast of the filter rows, a gather of fixed-size
detection records to rank 0, merged back into global frame order."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ffcnn_amd import dist as ffdist
from packref import pack_records
from ffcnn_amd.capi import DETS_DTYPE, FFGPU


def test_shard_range_covers_batch():
    for total in (1, 7, 64, 256, 257):
        for world in (1, 2, 3, 8):
            spans = [ffdist.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert ffdist.shard_range(256, 3, 8) == (96, 128)       # BASELINE config[4]: 32 contiguous frames per GPU


def fake_records(lo, hi):
    """deterministic per-frame records: frame g has (g % 5) boxes whose fields encode g"""
    rec = np.zeros(hi - lo, DETS_DTYPE)
    for i, g in enumerate(range(lo, hi)):
        k = g % 5
        rec[i]["count"] = k
        rec[i]["ncand"] = 2 * k
        for b in range(k):
            rec[i]["box"][b] = (g % 80, 0.5 + 0.01 * b, g, g + 1, g + 2 + b, g + 3)
    return rec


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # 1. weights: rank 0 holds them, everybody else zeros -> broadcast
        w = torch.arange(356576, dtype=torch.float32) if rank == 0 else torch.zeros(356576)
        ffdist.broadcast_weights(dist, w)
        ok_w = bool(torch.equal(w, torch.arange(356576, dtype=torch.float32)))
        # 2. shard + per-rank "forward" (fabricated records) + gather to rank 0, three steps
        lo, hi = ffdist.shard_range(total, rank, world)
        per = -(-total // world)
        local = np.zeros(per, DETS_DTYPE)
        local[: hi - lo] = fake_records(lo, hi)
        t = torch.from_numpy(local.view(np.uint8).copy())
        out = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
        merged = None
        for _ in range(3):
            got = ffdist.gather_records(dist, t, dst=0, out=out)
            if rank == 0:
                sizes = [ffdist.shard_range(total, r, world) for r in range(world)]
                merged = ffdist.merge_records([g.numpy() for g in got], [b - a for a, b in sizes], DETS_DTYPE)
        dist.barrier()
        if rank == 0:
            want = fake_records(0, total)
            q.put((ok_w, bool(np.array_equal(merged, want)), len(merged)))
        else:
            q.put((ok_w, True, 0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 7])
def test_two_rank_broadcast_and_gather(total):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[0] and r[1] for r in res), res
    assert sorted(r[2] for r in res) == [0, total]
    assert DETS_DTYPE.itemsize == 16 + 24 * FFGPU.MAX_DET


# ---- the record ring of the multi-GPU step: forwards write slots, halves travel in groups, the last half is flushed
def _ring_worker(rank, world, port, steps, M, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rec_bytes = 48
        ring = torch.zeros((2, M, rec_bytes), dtype=torch.uint8)                # what ffgpu_exec_set_ring hands the kernel
        out = [torch.empty(M * rec_bytes, dtype=torch.uint8) for _ in range(world)] if rank == 0 else None
        seen = {}

        def ship(g, n_valid, first_step):
            got = ffdist.gather_records(dist, ring[g].view(-1), dst=0, out=out)
            if rank == 0:
                for s, per_rank in enumerate(ffdist.unpack_group([t.numpy() for t in got], n_valid, rec_bytes)):
                    seen[first_step + s] = per_rank

        for i in range(steps):
            g, slot = ffdist.ring_slot(i, M)
            ring[g, slot] = torch.full((rec_bytes,), (i * 7 + rank * 3) % 251, dtype=torch.uint8)   # "the NMS kernel"
            if ffdist.group_due(i, M):
                ship(g, M, i - M + 1)
        if steps % M:
            ship(ffdist.ring_slot(steps, M)[0], steps % M, steps - steps % M)                        # flush
        ok = None
        if rank == 0:
            ok = sorted(seen) == list(range(steps)) and all(
                seen[i][r] == bytes([(i * 7 + r * 3) % 251]) * rec_bytes for i in range(steps) for r in range(world))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("steps,M", [(10, 4), (8, 4), (3, 8), (17, 1)])
def test_record_ring_groups_gloo(steps, M):
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ring_worker, args=(r, world, port, steps, M, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] is True


# ---- compact records: the numpy mirror of ffgpu_pack_records packs / unpacks without loss, truncates with flags, and a
# partial group of packed steps travels through a two-rank gather
def _random_records(rng, batch, max_per_frame):
    recs = np.zeros(batch, DETS_DTYPE)
    for n in range(batch):
        c = int(rng.integers(0, max_per_frame + 1))
        recs[n]["count"], recs[n]["ncand"], recs[n]["overflow"] = c, c + int(rng.integers(0, 5)), 0
        for k in range(c):
            recs[n]["box"][k] = (int(rng.integers(0, 80)), rng.random(), *(rng.random(4) * 320).tolist())
    return recs


def test_pack_unpack_records():
    rng = np.random.default_rng(5)
    for batch, per, cap in ((64, 3, 1024), (4, 6, 64), (7, 0, 8), (5, 9, 10), (3, 4, 1)):
        recs = _random_records(rng, batch, per)
        blk = pack_records(recs, cap)
        assert len(blk) == ffdist.packed_bytes(batch, cap) and len(blk) % 16 == 0
        back = ffdist.unpack_records(blk, DETS_DTYPE)
        total = int(recs["count"].sum())
        hdr = blk[:16].view(np.int32)
        assert tuple(hdr) == (min(total, cap), int(total > cap), batch, cap)
        if total <= cap:
            assert back.tobytes() == recs.tobytes()
        else:
            assert int(back["count"].sum()) == cap
            lost = back["count"] < recs["count"]
            assert lost.any() and ((back["overflow"] & 2) != 0).tolist() == lost.tolist()
            first = 0
            for n in range(batch):                                  # what is kept is a prefix of every frame's boxes, in frame order
                k = int(back[n]["count"])
                assert back[n]["box"][:k].tobytes() == recs[n]["box"][:k].tobytes()
                assert k == max(0, min(int(recs[n]["count"]), cap - first))
                first += int(recs[n]["count"])


def _packed_worker(rank, world, port, steps, M, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        batch, cap = 6, 24
        pb = ffdist.packed_bytes(batch, cap)
        rng = np.random.default_rng(100 + rank)
        mine = [_random_records(rng, batch, 5) for _ in range(steps)]
        cring = torch.zeros((2, M, pb), dtype=torch.uint8)
        out = [torch.empty(M * pb, dtype=torch.uint8) for _ in range(world)] if rank == 0 else None
        seen = {}

        def ship(g, n_valid, first_step):                          # only the written slots travel (bench.py's flush)
            nb = n_valid * pb
            got = ffdist.gather_records(dist, cring[g].view(-1)[:nb], dst=0, out=[t[:nb] for t in out] if out else None)
            if rank == 0:
                for s, per_rank in enumerate(ffdist.unpack_group([t.numpy() for t in got], n_valid, pb)):
                    seen[first_step + s] = [ffdist.unpack_records(b, DETS_DTYPE) for b in per_rank]

        for i in range(steps):
            g, slot = ffdist.ring_slot(i, M)
            cring[g, slot] = torch.from_numpy(pack_records(mine[i], cap))
            if ffdist.group_due(i, M):
                ship(g, M, i - M + 1)
        if steps % M:
            ship(ffdist.ring_slot(steps, M)[0], steps % M, steps - steps % M)
        ok = None
        if rank == 0:
            ok = sorted(seen) == list(range(steps))
            for r in range(world):
                ref = [_random_records(np.random.default_rng(100 + r), batch, 5)]
                gen = np.random.default_rng(100 + r)
                for i in range(steps):
                    want = _random_records(gen, batch, 5)
                    w2 = ffdist.unpack_records(pack_records(want, cap), DETS_DTYPE)
                    ok = ok and seen[i][r].tobytes() == w2.tobytes()
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("steps,M", [(5, 4), (8, 4), (3, 8)])
def test_packed_record_groups_gloo(steps, M):
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_packed_worker, args=(r, world, port, steps, M, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] is True
