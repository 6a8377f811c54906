"""Multi-GPU sharding of independent 15-s frames (one process per GPU, torch.distributed).

Frames are fully independent (per-frame hash table, per-frame duplicate filter), so the decode itself needs
no collective: rank r decodes the contiguous block shard(n, r, world).  The only exchange is the final gather
of fixed-capacity record/event blocks to rank 0 (RCCL over xGMI when the backend is "nccl"; "gloo" on CPU for
tests).  The reference has no counterpart.

The gather the measured path uses (bench.py, inside the timed step):
  PackedGather(handle, n_frames)          every batch ends with three small kernels that pack what the host message layer reads of it
                                          (records of the candidates that decoded or logged an unpack() call + the used event log,
                                          ~4 KB per config-1 frame instead of 13-24 KB; include/ft8rx.h "packed results") into a
                                          device buffer; submit() exchanges the byte counts on the host (gloo, 8 bytes per rank) and
                                          gathers the packed buffers to rank `dst` on a side stream (RCCL over xGMI), where one D2H
                                          copy per rank lands them in
                                          page-locked host memory -- all of it overlapping the next batch's kernels.  Rank `dst`
                                          keeps the packed form (a view per rank / frame, `_lib.Packed`) and can render any frame's
                                          messages from it (`_lib.package_packed`).  With gloo (flow tests) the GPU writes the packed
                                          buffer straight into page-locked host memory and the gather moves host bytes.
The dense variants (fixed-capacity blocks; kept for tools and as a cross-check, same result on rank `dst` = the concatenation over
ranks in rank order = global frame order for shard()):
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
    """RCCL gather of the handle's latest batch straight from device memory (backend nccl).  -> like gather_results, except that
    only events[f, :event_counts[f]] is meaningful (columns between a frame's count and the largest count of the job hold stale
    log entries here and zeros in gather_results).
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


class PackedGather:
    """Gather of every batch's PACKED results to rank `dst`, overlapped with the next batch (module docstring).

        g = PackedGather(handle, B)                  # all ranks; sets the handle's packed output
        handle.enqueue(...); view = handle.fetch_view(B)
        g.submit()                                   # all ranks, after the fetch of that batch: asynchronous from here on
        ...                                          # (next enqueue / host message layer of this rank's own frames)
        parts = g.collect()                          # rank dst: [Packed of rank 0, rank 1, ...] of the oldest submitted batch; else None

    At most two gathers are in flight (the handle has two result slots); submit() first completes the one that used the same slot.
    per_frame: bytes of records + events reserved per frame in each rank's two pack buffers (default None = the worst case, 48 max_cands
    + 12 KB: a batch always fits; config-1 frames use ~6 KB, and a smaller figure makes an oversized batch raise instead of being
    cut).  Rank dst's receive and page-locked buffers are sized from the byte counts actually seen (twice the first batch's, grown if
    a later batch needs more), not from the capacity.  repeat > 1 (measurement aid) gathers every batch `repeat` times into distinct
    buffers: rank dst's receive + D2H load of `repeat` ranks in a one-rank group."""

    MIN_ROW = 1 << 20          # smallest receive row rank dst allocates (bytes); tests lower it to reach the growth path with small frames

    def __init__(self, handle, n_frames, dst=0, group=None, force=False, per_frame=None, repeat=1):
        from . import _lib
        self._lib, self.h, self.B, self.dst, self.group, self.repeat = _lib, handle, int(n_frames), dst, group, max(1, int(repeat))
        self.active = dist.is_initialized() and (dist.get_world_size(group) > 1 or force)
        self.world = dist.get_world_size(group) if self.active else 1
        self.rank = dist.get_rank(group) if self.active else 0
        self.nccl = self.active and dist.get_backend(group) == "nccl"
        self.cap = (_lib.packed_capacity(self.B, handle.cfg.max_cands, per_frame) + 255) & ~255
        self.pending = []                  # (slot, sizes, event-or-None, part stride) in submit order
        self._fence = [None, None]         # keeps the event handed to ft8rx_packed_output_fence alive
        self.ready, self._last = [], None  # completed gathers not collected yet (oldest first); the most recent result
        self.seconds = []                  # host time spent inside submit() per call (the size exchange blocks; the rest is asynchronous)
        self.phases = []                   # the same, split: header / wait_slot (the gather two batches ago) / sizes / issue
        # the byte counts are exchanged on the host (a gloo group next to the RCCL one): a device-side exchange would make every
        # submit() wait for the side stream, and a side stream shares one of the runtime's few hardware queues with a chunk stream
        # of the decode -- measured: the 8-byte all_gather then waits for the whole batch in front of it (135 ms at config 3)
        self.cpu_group = group
        if self.nccl:
            try:
                self.cpu_group = dist.new_group(ranks=(dist.get_process_group_ranks(group) if group is not None else None), backend="gloo")
            except Exception as e:                   # no gloo transport on this box: the counts go over RCCL (submit() then waits for the side stream)
                import warnings
                warnings.warn(f"PackedGather: no gloo group for the size exchange ({type(e).__name__}: {e}); using the device path", RuntimeWarning)
                self.cpu_group = None
        if self.nccl:
            dev = torch.device("cuda", torch.cuda.current_device())
            self.stream = torch.cuda.Stream(device=dev)
            self.src = [torch.empty(self.cap, dtype=torch.uint8, device=dev) for _ in range(2)]
            ptrs = [t.data_ptr() for t in self.src]
            self.recv = self.host = None               # rank dst: allocated by _room() from the first batch's byte counts
            self.row = 0
        else:
            # host path (gloo, or no process group at all): the pack kernels write page-locked host memory directly
            self.stream = None
            self._pin = [handle.pinned_bytes(self.cap) for _ in range(2)]
            self.src = [torch.from_numpy(a) for a in self._pin]
            ptrs = [a.ctypes.data for a in self._pin]
            self.recv = self.host = None
            self.row = 0
        handle.set_packed_output(ptrs[0], ptrs[1], self.cap)

    def _room(self, m):
        """Rank dst: receive (nccl) and host buffers with rows of at least m bytes for world x repeat parts, two sets (one per result
        slot).  Sized to twice the first batch's largest part; a later batch that needs more first waits for the gathers in flight."""
        if m <= self.row:
            return
        while self.pending:
            self._retire()
        self.row = min(self.cap, max(2 * m, self.MIN_ROW))
        self.row = (self.row + 255) & ~255
        n = self.world * self.repeat
        if self.nccl:
            dev = self.src[0].device
            self.recv = [torch.empty((n, self.row), dtype=torch.uint8, device=dev) for _ in range(2)]
            self.host = [torch.empty((n, self.row), dtype=torch.uint8, pin_memory=True) for _ in range(2)]
        else:
            self.host = [torch.empty((n, self.row), dtype=torch.uint8) for _ in range(2)]

    def close(self):
        self.drain()
        self.h.set_packed_output(None, None, 0)

    def _finish(self, item):
        slot, sizes, ev, m = item
        if ev is not None:
            ev.synchronize()
        if not self.active:
            return [self._lib.Packed(self._pin[slot][:sizes[0]])]
        if self.rank != self.dst:
            return None
        flat = self.host[slot].view(-1).numpy()          # this batch's parts lie back to back, m bytes apart
        return [self._lib.Packed(flat[i * m:i * m + sizes[i % self.world]]) for i in range(self.world * self.repeat)]

    def submit(self):
        """Start the gather of the batch the last fetch returned (every rank calls this once per fetched batch, in the same order)."""
        import time
        t0 = time.perf_counter()
        ph = {}

        def mark(name, _t=[t0]):
            now = time.perf_counter()
            ph[name] = ph.get(name, 0.0) + now - _t[0]
            _t[0] = now
        slot, hdr = self.h.packed_results()
        if hdr["overflow"]:
            raise self._lib.Ft8rxError(f"PackedGather: a batch needs {hdr['bytes']} packed bytes, the buffers hold {self.cap} (raise per_frame)")
        mark("header")
        while any(p[0] == slot for p in self.pending):          # the gather that last used this slot's buffers
            self._retire()
        mark("wait_slot")
        self.phases.append(ph)
        nbytes = int(hdr["bytes"])
        if not self.active:
            self.pending.append((slot, [nbytes], None, 0))
            self.seconds.append(time.perf_counter() - t0)
            return
        if self.nccl and self.cpu_group is None:
            with torch.cuda.stream(self.stream):
                mine = torch.tensor([nbytes], dtype=torch.int64).to(self.src[slot].device)
                allsz = torch.empty(self.world, dtype=torch.int64, device=mine.device)
                dist.all_gather_into_tensor(allsz, mine, group=self.group)
                sizes = [int(x) for x in allsz.tolist()]
        else:
            mine = torch.tensor([nbytes], dtype=torch.int64)
            lst = [torch.zeros_like(mine) for _ in range(self.world)]
            dist.all_gather(lst, mine, group=self.cpu_group)
            sizes = [int(x.item()) for x in lst]
        mark("sizes")
        n = self.world * self.repeat
        if self.rank == self.dst:
            self._room((max(sizes) + 255) & ~255)
        if self.nccl:
            with torch.cuda.stream(self.stream):
                m = (max(sizes) + 255) & ~255
                # the parts of this batch are received back to back (m bytes apart) so that ONE D2H copy moves them all
                flat = self.recv[slot].view(-1) if self.rank == self.dst else None
                for rep in range(self.repeat):
                    out = [flat[(rep * self.world + r) * m:(rep * self.world + r + 1) * m] for r in range(self.world)] if flat is not None else None
                    dist.gather(self.src[slot][:m], out, dst=self.dst, group=self.group)
                if flat is not None:
                    self.host[slot].view(-1)[:n * m].copy_(flat[:n * m], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.stream)
                # the pack kernels that next write this slot's buffer (two enqueues from now) wait for the send on the device
                self._fence[slot] = ev
                self.h.packed_fence(slot, ev.cuda_event)
                mark("issue")
        else:
            m = max(sizes)
            flat = self.host[slot].view(-1) if self.rank == self.dst else None
            for rep in range(self.repeat):
                out = [flat[(rep * self.world + r) * m:(rep * self.world + r + 1) * m] for r in range(self.world)] if flat is not None else None
                dist.gather(self.src[slot][:m], out, dst=self.dst, group=self.group)
            mark("issue")
            ev = None
        self.pending.append((slot, sizes, ev, m))
        self.seconds.append(time.perf_counter() - t0)

    def _retire(self):
        """Complete the oldest gather in flight; its result waits in self.ready (the two most recent are kept) until collected."""
        res = self._finish(self.pending.pop(0))
        self.ready.append(res)
        del self.ready[:-2]
        self._last = res
        return res

    def outstanding(self):
        """Gathers submitted and not yet collected (in flight or completed)."""
        return len(self.pending) + len(self.ready)

    def collect(self):
        """Rank dst: the per-rank Packed views of the OLDEST submitted batch not collected yet (valid until two more submits);
        other ranks: None.  Waits for that gather if it is still in flight; with nothing outstanding: the last result again."""
        if not self.ready and self.pending:
            self._retire()
        if self.ready:
            return self.ready.pop(0)
        return self._last

    def drain(self):
        """Wait for every gather in flight; -> the last one's result (as collect)."""
        while self.pending:
            self._retire()
        self.ready.clear()
        return self._last
