"""Multi-GPU sharding of independent 15-s frames (one process per GPU, torch.distributed).

Frames are fully independent (per-frame hash table, per-frame duplicate filter), so the decode itself needs
no collective: rank r decodes the contiguous block shard(n, r, world).  The only exchange is the final gather
of fixed-capacity record/event blocks to rank 0 (RCCL over xGMI when the backend is "nccl"; "gloo" on CPU for
tests).  The reference has no counterpart.

Two gather entry points, same result on rank `dst` (the concatenation over ranks in rank order = global frame order for shard()):
  gather_results(rec, cnt, ev, evc)       host arrays in, through the backend's device (gloo: CPU tensors; nccl: one H2D per rank)
  gather_results_device(handle, n_frames) nccl only: the latest batch's results go device -> device into torch buffers
                                          (ft8rx_results_to_device), RCCL gathers them over xGMI, and rank `dst` makes the single
                                          D2H copy -- no host round trip on the sending ranks."""
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


def _counts(n, dev, group):
    world = dist.get_world_size(group)
    t = torch.tensor([n], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(counts, t, group=group)
    return [int(c.item()) for c in counts]


def _gather_rows(t, counts, dst, group):
    """t: uint8 [counts[rank], row] on the backend's device -> on dst a list of per-rank tensors (padded to max(counts) rows)."""
    rank = dist.get_rank(group)
    nmax = max(counts)
    if t.shape[0] < nmax:
        t = torch.cat([t, torch.zeros((nmax - t.shape[0], t.shape[1]), dtype=t.dtype, device=t.device)])
    out = [torch.empty_like(t) for _ in counts] if rank == dst else None
    dist.gather(t.contiguous(), out, dst=dst, group=group)
    return out


def _gather_bytes(arr, dst, group):
    """Gather one numpy array per rank (same dtype/shape[1:], possibly different shape[0]) to `dst`."""
    dev = _device()
    counts = _counts(arr.shape[0], dev, group)
    row = int(np.prod(arr.shape[1:], dtype=np.int64)) * arr.dtype.itemsize
    flat = np.ascontiguousarray(arr).view(np.uint8).reshape(arr.shape[0], row)
    out = _gather_rows(torch.from_numpy(flat.copy()).to(dev), counts, dst, group)
    if out is None:
        return None
    parts = [o.cpu().numpy()[:c].reshape(-1).view(arr.dtype).reshape((c,) + arr.shape[1:]) for o, c in zip(out, counts)]
    return np.concatenate(parts)


def _used_events(evc, cap, dev, group):
    """Event-log columns in use anywhere in the job: max over all ranks and frames of min(event_count, cap).  One all_reduce."""
    t = torch.clamp(evc.to(torch.int64).max() if evc.numel() else torch.zeros((), dtype=torch.int64, device=dev), 0, cap).reshape(1).to(dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


def gather_results(rec, cnt, ev, evc, dst=0, group=None, force=False):
    """Per-rank (records[B_r, max_cands], counts[B_r], events[B_r, CAP], event_counts[B_r]) -> on `dst` the
    concatenation over ranks in rank order (= global frame order for shard()); None elsewhere.  Only the event-log columns in use
    travel; the rest of each returned row is zero.  force=True runs the collectives even in a one-rank group (tests)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return rec, cnt, ev, evc
    used = _used_events(torch.from_numpy(np.ascontiguousarray(evc)), ev.shape[1], _device(), group)
    outs = [_gather_bytes(a, dst, group) for a in (rec, cnt, ev[:, :max(used, 1)], evc)]
    if dist.get_rank(group) != dst:
        return None
    full = np.zeros((outs[2].shape[0], ev.shape[1]), ev.dtype)
    full[:, :outs[2].shape[1]] = outs[2]
    return outs[0], outs[1], full, outs[3]


def gather_results_device(handle, n_frames, dst=0, group=None, force=False):
    """RCCL gather of the handle's latest batch straight from device memory (backend nccl).  -> like gather_results.
    force=True runs the whole device path (D2D into torch buffers, RCCL all_gather / gather, one D2H on dst) even in a one-rank
    group: that is how the path is exercised on a one-GPU test box."""
    from . import _lib
    if not dist.is_initialized():
        if force:
            raise _lib.Ft8rxError("gather_results_device(force=True) needs an initialised process group")
        return handle.fetch(n_frames)
    if dist.get_world_size(group) == 1 and not force:
        return handle.fetch(n_frames)
    if dist.get_backend(group) != "nccl":
        raise _lib.Ft8rxError("gather_results_device needs the nccl (RCCL) backend; use gather_results with gloo")
    dev = _device()
    B, mc = int(n_frames), int(handle.cfg.max_cands)
    shapes = [(B, mc * _lib.RECORD_DTYPE.itemsize), (B, 4), (B, _lib.EVENT_CAP * _lib.EVENT_DTYPE.itemsize), (B, 4)]
    bufs = [torch.empty(s, dtype=torch.uint8, device=dev) for s in shapes]
    handle.results_to_device(B, bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(), bufs[3].data_ptr())
    counts = _counts(B, dev, group)
    # only the event-log columns in use cross xGMI and PCIe (the log is [B][512] x 24 B, typically a tenth used)
    used = max(1, _used_events(bufs[3].view(torch.int32).reshape(-1), _lib.EVENT_CAP, dev, group))
    bufs[2] = bufs[2][:, :used * _lib.EVENT_DTYPE.itemsize].contiguous()
    outs = [_gather_rows(b, counts, dst, group) for b in bufs]
    if dist.get_rank(group) != dst:
        return None
    dts = [(_lib.RECORD_DTYPE, (mc,)), (np.dtype(np.int32), ()), (_lib.EVENT_DTYPE, (used,)), (np.dtype(np.int32), ())]
    res = []
    for out, (dt, tail) in zip(outs, dts):
        parts = [o[:c].cpu().numpy().reshape(-1).view(dt).reshape((c,) + tail) for o, c in zip(out, counts)]
        res.append(np.concatenate(parts))
    full = np.zeros((res[2].shape[0], _lib.EVENT_CAP), _lib.EVENT_DTYPE)
    full[:, :used] = res[2]
    res[2] = full
    return tuple(res)
