"""Property tests (hypothesis) of the host-side sharding logic the multi-GPU paths share: ffcnn_amd.dist (torchrun path) and
ffgpu_shard_range (the C node path, ffgpu_node.inc) -- SURVEY.md section 8(e): contiguous shards of a batch of independent
frames, one gather of fixed-size records.  No GPU involved."""
import ctypes as C

import numpy as np
from hypothesis import given, settings, strategies as st

from ffcnn_amd import capi, dist as ffdist


@settings(max_examples=300, deadline=None)
@given(total=st.integers(0, 5000), world=st.integers(1, 64))
def test_shards_tile_the_batch(total, world):
    prev = 0
    sizes = []
    for r in range(world):
        lo, hi = ffdist.shard_range(total, r, world)
        assert lo == prev and hi >= lo                       # contiguous, in rank order, no gaps and no overlap
        prev = hi
        sizes.append(hi - lo)
    assert prev == total
    assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)     # balanced, earlier ranks take the extra


@settings(max_examples=200, deadline=None)
@given(total=st.integers(0, 5000), world=st.integers(1, 64), data=st.data())
def test_c_and_python_shards_agree(total, world, data):
    L = capi.lib()
    r = data.draw(st.integers(0, world - 1))
    lo, hi = C.c_int(), C.c_int()
    L.ffgpu_shard_range(total, r, world, C.byref(lo), C.byref(hi))
    assert (lo.value, hi.value) == ffdist.shard_range(total, r, world)


@settings(max_examples=200, deadline=None)
@given(M=st.integers(1, 64), steps=st.integers(1, 400))
def test_ring_slots_and_due_groups(M, steps):
    """every step lands in exactly one slot of one half; a half is shipped when its last slot is written, and is not written
    again before the OTHER half has been shipped (the side stream has a whole group's time to read it)"""
    seen = {}
    shipped = []
    for i in range(steps):
        g, s = ffdist.ring_slot(i, M)
        assert g in (0, 1) and 0 <= s < M
        assert s == i % M and g == (i // M) % 2
        if s == 0 and i >= 2 * M:
            assert shipped and shipped[-1] == 1 - g           # the other half went out last; this one the time before
            assert shipped[-2] == g
        seen[(g, s)] = i
        if ffdist.group_due(i, M):
            assert s == M - 1
            shipped.append(g)
    assert len(shipped) == steps // M
    tail = steps % M                                          # what bench.py's flush() still has to ship
    assert all(seen[(ffdist.ring_slot(steps - 1 - k, M))] == steps - 1 - k for k in range(min(tail, steps)))


def _pack(records, cap):
    """numpy restatement of ffgpu_pack_records' block layout (include/ffcnn_hip.h) -- what unpack_records must invert"""
    batch = len(records)
    blk = np.zeros(ffdist.packed_bytes(batch, cap), np.uint8)
    fr = np.zeros((batch, 4), np.int32)
    boxes = np.zeros(cap, capi.BOX_DTYPE)
    total = over = 0
    for n, r in enumerate(records):
        kept = int(r["count"])
        if total + kept > cap:
            kept = max(0, cap - total)
            over = 1
        fr[n] = (kept, r["ncand"], r["overflow"], total)
        boxes[total:total + kept] = r["box"][:kept]
        total += kept
    blk[:16] = np.array([total, over, batch, cap], np.int32).view(np.uint8)
    blk[16:16 + 16 * batch] = fr.reshape(-1).view(np.uint8)
    blk[16 + 16 * batch:16 + 16 * batch + 24 * cap] = boxes.view(np.uint8)
    return blk


@settings(max_examples=100, deadline=None)
@given(batch=st.integers(1, 40), seed=st.integers(0, 2**31 - 1), roomy=st.booleans())
def test_packed_records_round_trip(batch, seed, roomy):
    rng = np.random.default_rng(seed)
    rec = np.zeros(batch, capi.DETS_DTYPE)
    for n in range(batch):
        k = int(rng.integers(0, 9))
        rec[n]["count"], rec[n]["ncand"], rec[n]["overflow"] = k, k + int(rng.integers(0, 50)), int(rng.integers(0, 2))
        rec[n]["box"]["type"][:k] = rng.integers(0, 80, k)
        for fld in ("score", "x1", "y1", "x2", "y2"):
            rec[n]["box"][fld][:k] = rng.uniform(0, 320, k).astype(np.float32)
    need = int(rec["count"].sum())
    cap = need + 3 if roomy else max(1, need // 2)
    assert ffdist.packed_bytes(batch, cap) % 16 == 0 and ffdist.packed_bytes(batch, cap) >= 16 + 16 * batch + 24 * cap
    assert capi.packed_records_bytes(batch, cap) == ffdist.packed_bytes(batch, cap)           # the C side sizes it the same
    out = ffdist.unpack_records(_pack(rec, cap), capi.DETS_DTYPE)
    left = cap
    for n in range(batch):
        kept = min(int(rec[n]["count"]), left)
        left -= kept
        assert out[n]["count"] == kept and out[n]["ncand"] == rec[n]["ncand"] and out[n]["overflow"] == rec[n]["overflow"]
        assert np.array_equal(out[n]["box"][:kept], rec[n]["box"][:kept])
        assert not out[n]["box"][kept:].view(np.uint8).any()


@settings(max_examples=100, deadline=None)
@given(world=st.integers(1, 8), total=st.integers(1, 300), seed=st.integers(0, 2**31 - 1))
def test_merge_restores_global_frame_order(world, total, seed):
    """rank-ordered shards of records -> the global frame order (frame f of the job = record f)"""
    rng = np.random.default_rng(seed)
    allrec = np.zeros(total, capi.DETS_DTYPE)
    allrec["ncand"] = np.arange(total)
    allrec["count"] = rng.integers(0, 5, total)
    shards = [ffdist.shard_range(total, r, world) for r in range(world)]
    room = max(hi - lo for lo, hi in shards)                  # the gather moves equal-sized buffers: the largest shard's
    bufs = []
    for lo, hi in shards:
        b = np.zeros(room, capi.DETS_DTYPE)
        b[:hi - lo] = allrec[lo:hi]
        b[hi - lo:]["ncand"] = -1                             # padding must not leak into the merge
        bufs.append(b.tobytes())
    merged = ffdist.merge_records(bufs, [hi - lo for lo, hi in shards], capi.DETS_DTYPE)
    assert np.array_equal(merged, allrec)


def test_c_unpack_of_packed_records_is_the_inverse_of_packing():
    """ffgpu_unpack_records (host code of the C node path: what ffgpu_node_wait runs on the gathered blocks) against the numpy mirror of
    the device packer (tests/packref.py) and the Python unpacker of the torchrun path: random record sets round-trip byte for byte;
    a block that had to drop boxes, or one of another (batch, cap), is reported (1) so that the caller fetches the full records"""
    import ctypes as C
    from ffcnn_amd import capi
    from ffcnn_amd import dist as ffdist
    from packref import pack_records
    L = capi.lib()
    rng = np.random.default_rng(5)
    for trial in range(40):
        batch = int(rng.integers(1, 40))
        recs = np.zeros(batch, capi.DETS_DTYPE)
        for n in range(batch):
            k = int(rng.integers(0, 6)) if rng.random() < 0.8 else int(rng.integers(0, capi.FFGPU.MAX_DET + 1))
            recs[n]["count"] = k
            recs[n]["ncand"] = k + int(rng.integers(0, 50))
            recs[n]["overflow"] = int(rng.integers(0, 2)) * 4
            recs[n]["nfull"] = k + int(rng.integers(0, 3))
            for f in ("score", "x1", "y1", "x2", "y2"):
                recs[n]["box"][f][:k] = rng.uniform(0, 300, k)
            recs[n]["box"]["type"][:k] = rng.integers(0, 80, k)
        total = int(recs["count"].sum())
        for cap in (max(1, total), total + 7, max(1, total // 2)):
            blk = pack_records(recs, cap)
            out = np.zeros(batch, capi.DETS_DTYPE)
            out["count"] = -7                                         # (every field must be written)
            rc = L.ffgpu_unpack_records(blk.ctypes.data, batch, cap, out.ctypes.data)
            if cap >= total:
                assert rc == 0 and out.tobytes() == recs.tobytes(), (trial, cap)
                assert ffdist.unpack_records(blk, capi.DETS_DTYPE).tobytes() == recs.tobytes()
            else:
                assert rc == 1, (trial, cap, total)                   # boxes were dropped: fetch the full-size records instead
        blk = pack_records(recs, total + 3)
        assert L.ffgpu_unpack_records(blk.ctypes.data, batch + 1, total + 3, out.ctypes.data) == 1      # not a block of this (batch, cap)
        assert L.ffgpu_unpack_records(None, batch, 4, out.ctypes.data) == -1
