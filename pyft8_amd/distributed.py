"""Multi-GPU sharding of independent 15-s frames (one process per GPU, torch.distributed).

Frames are fully independent (per-frame hash table, per-frame duplicate filter), so the decode itself needs
no collective: rank r decodes the contiguous block shard(n, r, world).  The only exchange is the final gather
of fixed-capacity record/event blocks to rank 0 (RCCL over xGMI when the backend is "nccl"; "gloo" on CPU for
tests).  The reference has no counterpart."""
import numpy as np
import torch
import torch.distributed as dist


def shard(n_frames, rank, world):
    """Contiguous block of frames for `rank`: (start, count); the first n % world ranks get one extra."""
    base, extra = divmod(int(n_frames), int(world))
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def _device():
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def _gather_bytes(arr, dst, group):
    """Gather one numpy array per rank (same dtype/shape[1:], possibly different shape[0]) to `dst`."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = _device()
    n = torch.tensor([arr.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    row = int(np.prod(arr.shape[1:])) * arr.dtype.itemsize
    pad = np.zeros((max(counts), row), np.uint8)
    pad[:arr.shape[0]] = np.ascontiguousarray(arr).view(np.uint8).reshape(arr.shape[0], row)
    t = torch.from_numpy(pad).to(dev)
    out = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
    dist.gather(t, out, dst=dst, group=group)
    if rank != dst:
        return None
    parts = [o.cpu().numpy()[:c].reshape(-1).view(arr.dtype).reshape((c,) + arr.shape[1:]) for o, c in zip(out, counts)]
    return np.concatenate(parts)


def gather_results(rec, cnt, ev, evc, dst=0, group=None):
    """Per-rank (records[B_r, max_cands], counts[B_r], events[B_r, CAP], event_counts[B_r]) -> on `dst` the
    concatenation over ranks in rank order (= global frame order for shard()); None elsewhere."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return rec, cnt, ev, evc
    outs = [_gather_bytes(a, dst, group) for a in (rec, cnt, ev, evc)]
    return tuple(outs) if dist.get_rank(group) == dst else None
