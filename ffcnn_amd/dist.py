"""Data-parallel sharding of a batch of frames over the GPUs of one node.

The forward path has no cross-frame state, so the batch is cut into
contiguous per-rank slices, every rank runs the whole net on its slice, and exactly two
collectives exist:
  * once, at load: broadcast of the folded filter rows (NET.weight_buf layout) from rank 0;
  * per step: gather of the fixed-size per-frame detection records (ffgpu_frame_dets) to rank 0.
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
