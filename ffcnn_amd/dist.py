"""Data-parallel sharding of a batch of frames over the GPUs of one node.

The forward path has no cross-frame state, so the batch is cut into
contiguous per-rank slices, every rank runs the whole net on its slice, and exactly two
collectives exist:
  * once, at load: broadcast of the folded filter rows (NET.weight_buf layout) from rank 0;
  * gather of the fixed-size per-frame detection records (ffgpu_frame_dets) to rank 0 -- in GROUPS of steps: forward k
    writes its records into slot k % (2 M) of a ring (ffgpu_exec_set_ring), and every M steps the finished half of the
    ring travels in one collective (ring_slot / group_due / unpack_group below).
One process per GPU; `torch.distributed` with backend "nccl" (= RCCL over xGMI) on the GPUs and
"gloo" in the CPU tests.  Nothing here touches a kernel: the functions move opaque byte tensors.
"""
import numpy as np


def shard_range(total, rank, world):
    """Contiguous frames [lo, hi) of `rank`: sizes differ by at most one, earlier ranks take the extra."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def broadcast_weights(dist, weights, src=0):
    """In-place broadcast of the weight tensor (a view of the library's device buffer on GPU runs)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(weights, src=src)
    return weights


def gather_records(dist, local, dst=0, out=None):
    """Gather equal-sized uint8 record tensors to `dst`.  Returns the list of per-rank tensors on dst,
    None elsewhere.  `out` (a list of preallocated tensors) avoids allocations in a timed loop."""
    if not dist.is_initialized():
        return [local]
    world = dist.get_world_size()
    rank = dist.get_rank()
    if rank == dst and out is None:
        out = [local.new_empty(local.shape) for _ in range(world)]
    dist.gather(local, out if rank == dst else None, dst=dst)
    return out if rank == dst else None


def merge_records(per_rank_bytes, frames_per_rank, dets_dtype):
    """rank-ordered record buffers -> one structured array in global frame order."""
    parts = []
    for buf, n in zip(per_rank_bytes, frames_per_rank):
        a = np.frombuffer(bytes(buf), dets_dtype)
        parts.append(a[:n])
    return np.concatenate(parts) if parts else np.zeros(0, dets_dtype)


# ---- record ring: which half / slot a step's records sit in, when a half is due, how rank 0 unpacks a gathered half
def ring_slot(step, group_steps):
    """(half of the ring, slot inside the half) that forward number `step` (0-based since the ring was set) writes."""
    return (step // group_steps) % 2, step % group_steps


def group_due(step, group_steps):
    """True when forward number `step` completes a half of the ring (it is then gathered on the side stream)."""
    return step % group_steps == group_steps - 1


def unpack_group(gathered, n_steps, rec_bytes):
    """gathered: per-rank byte buffers of one half of the ring (group_steps * rec_bytes each).
    Returns [step][rank] -> bytes of that step's records, for the first n_steps slots (a flushed, partial half has
    fewer valid slots than the half has room for)."""
    out = []
    for s in range(n_steps):
        out.append([bytes(memoryview(g)[s * rec_bytes:(s + 1) * rec_bytes]) for g in gathered])
    return out


# ---- compact records (ffgpu_pack_records): what actually travels in the gather.  Layout of one step's block:
#     int32 total, over, batch, cap | {int32 count, ncand, overflow, first} x batch | BBOX box[cap]
def packed_bytes(batch, cap, box_bytes=24):
    return (16 + 16 * batch + box_bytes * cap + 15) & ~15


def unpack_records(block, dets_dtype):
    """one step's packed block (bytes / uint8 array) -> structured array of `batch` full-size records (unused boxes zero)."""
    b = np.frombuffer(bytes(block), np.uint8)
    total, over, batch, cap = (int(v) for v in b[:16].view(np.int32))
    box_dtype = dets_dtype["box"].subdtype[0]
    fr = b[16:16 + 16 * batch].view(np.int32).reshape(batch, 4)
    box = b[16 + 16 * batch:16 + 16 * batch + box_dtype.itemsize * cap].view(box_dtype)
    out = np.zeros(batch, dets_dtype)
    first = 0                                                   # boxes lie behind each other in frame order
    for n in range(batch):
        kept, ncand, ovf, nfull = (int(v) for v in fr[n])
        out[n]["count"], out[n]["ncand"], out[n]["overflow"], out[n]["nfull"] = kept, ncand, ovf, nfull
        out[n]["box"][:kept] = box[first:first + kept]
        first += kept
    return out
