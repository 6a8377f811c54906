"""Multi-GPU sharding of independent 15-s frames (one process per GPU, torch.distributed).

Frames are fully independent (per-frame hash table, per-frame duplicate filter), so the decode itself needs
no collective: rank r decodes the contiguous block shard(n, r, world).  The only exchange is the final gather
of fixed-capacity record/event blocks to rank 0 (RCCL over xGMI when the backend is "nccl"; "gloo" on CPU for
tests).  The reference has no counterpart.

The gather the measured path uses (bench.py, inside the timed step):
  PackedGather(handle, n_frames)          every batch ends with three small kernels that pack what the host message layer reads of it
                                          (records of the candidates that decoded or logged an unpack() call + the used event log,
                                          ~6 KB per config-1 frame instead of 13-24 KB; include/ft8rx.h "packed results") into a
                                          device buffer.  submit() has NO rendezvous: a sending rank starts a point-to-point send of
                                          its buffer to rank `dst` on a side stream (RCCL over xGMI) and drops a three-number
                                          announcement into the process group's store; rank `dst` polls that mailbox, posts the
                                          receives announced so far into one flat device buffer per batch, parts back to back,
                                          and moves a complete batch to page-locked host memory with ONE D2H copy and one event
                                          (rounds 4-5: one of each per part) -- all of it overlapping the next batches' kernels.  Rank `dst` keeps the packed form (a view per rank / frame, `_lib.Packed`) and
                                          can render any frame's messages from it (`_lib.package_packed`).  With gloo (flow tests)
                                          the GPU writes the packed buffer straight into page-locked host memory and the gather
                                          moves host bytes.
The dense variants (fixed-capacity blocks; kept for tools and as a cross-check, same result on rank `dst` = the concatenation over
ranks in rank order = global frame order for shard()):
  gather_results(rec, cnt, ev, evc)       host arrays in, through the backend's device (gloo: CPU tensors; nccl: one H2D per rank)
  gather_results_device(handle, n_frames) nccl only: the latest batch's results go device -> device into torch buffers
                                          (ft8rx_results_to_device), RCCL gathers them over xGMI, and rank `dst` makes the single
                                          D2H copy -- no host round trip on the sending ranks."""
import os
import weakref

import numpy as np
import torch
import torch.distributed as dist


_DIAG = os.environ.get("FT8RX_GATHER_DIAG", "")      # measurement aid only (tools/r06_gather_diag.sh)


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


def _store():
    """The process group's rendezvous store (a TCPStore served by a background thread of global rank 0): PackedGather's mailbox for its
    32-byte control messages.  A set() is one small TCP message to that server thread -- no rank's Python thread is involved in taking
    it -- and check() polls without blocking, which torch.distributed's own point-to-point receives cannot do on gloo (a gloo Work
    only learns that it is complete inside wait())."""
    from torch.distributed import distributed_c10d as c10d
    return c10d._get_default_store()


class PackedGather:
    """Gather of every batch's PACKED results to rank `dst`, overlapped with the following batches (module docstring).

        g = PackedGather(handle, B)                  # all ranks; sets the handle's packed output
        handle.enqueue(...); view = handle.fetch_view(B)
        g.submit()                                   # all ranks, after the fetch of that batch: asynchronous from here on
        ...                                          # (next enqueue / host message layer of this rank's own frames)
        parts = g.collect()                          # rank dst: [Packed of rank 0, rank 1, ...] of the oldest submitted batch; else None

    No collective and no cross-rank rendezvous per batch (round 5; the constructor holds the only one, _handshake): a rank other than dst starts the point-to-point send of its buffer (RCCL
    send on a side stream / gloo isend) and then drops one control message (batch number, byte count, overflow flag) into the process
    group's store -- it never waits for another rank's batch, only (two batches later, before the same buffer is packed again) for its
    OWN send to have been taken.  Rank dst looks into the mailbox whenever it is in submit() / collect() / drain() and posts the data
    receive of every part that has been announced: parts land in arrival order, peer by peer, into a ring of `depth` receive sets
    (set = batch number % depth, row = rank).  A peer may therefore run up to `depth` batches ahead of the slowest one before its
    sends wait; rank dst itself waits only when it is `depth` batches ahead of the slowest peer, and in collect() / drain().
    (Round 4 exchanged the byte counts with a blocking all_gather and used dist.gather: every step was a rendezvous of all ranks.)

    per_frame: bytes of records + events reserved per frame in each rank's two pack buffers (default None = the worst case, 48 max_cands
    + 12 KB: a batch always fits; config-1 frames use ~6 KB, and a smaller figure makes an oversized batch raise -- on the rank it
    happens on and, through the control message, on rank dst -- instead of being cut).  Rank dst's receive and page-locked buffers are
    sized from the byte counts actually seen (twice the first batch's, grown if a later batch needs more), not from the capacity.
    repeat > 1 (measurement aid) gathers every batch `repeat` times into distinct rows: rank dst's receive + D2H load of `repeat` times
    as many ranks.  Every rank of `group` constructs the object, and all ranks construct their PackedGathers in the same order (the
    n-th one of a process uses mailbox n)."""

    MIN_ROW = 1 << 20          # smallest receive row rank dst allocates (bytes); tests lower it to reach the growth path with small frames
    TIMEOUT = 300.0           # seconds any wait may take before it raises instead of hanging (a peer died, a transport is stuck)
    TAG_DATA = 12
    TAG_HELLO = 13
    _instances = 0             # PackedGathers constructed in this process: the mailbox prefix

    def __init__(self, handle, n_frames, dst=0, group=None, force=False, per_frame=None, repeat=1, depth=4):
        from . import _lib
        self._lib, self.h, self.B, self.dst, self.group, self.repeat = _lib, handle, int(n_frames), dst, group, max(1, int(repeat))
        self.depth = max(2, int(depth))
        self.active = dist.is_initialized() and (dist.get_world_size(group) > 1 or force)
        self.world = dist.get_world_size(group) if self.active else 1
        self.rank = dist.get_rank(group) if self.active else 0
        self.nccl = self.active and dist.get_backend(group) == "nccl"
        self.cap = (_lib.packed_capacity(self.B, handle.cfg.max_cands, per_frame) + 255) & ~255
        self.seq = 0                       # batches submitted by this rank
        self.sends = []                    # completions of this rank's sends in flight, oldest first
        self.batches = {}                  # rank dst: batch number -> {"parts": [...], "left": n, "err": str or None}
        self.announced = {}                # rank dst: peer -> control messages received and not yet turned into data receives
        self.ready, self._last = [], None  # rank dst: completed batches not collected yet (oldest first); the most recent result
        self.collected = 0                 # batches handed out / dropped so far (all ranks count alike)
        self.done_upto = 0                 # rank dst: batches [collected, done_upto) are complete and wait in `ready`
        self.seconds = []                  # host time spent inside submit() per call
        self.phases = []                   # the same, split: header / wait_slot (this rank's own send two batches ago) / sizes (control message; rank dst: polling) / issue
        self.row = 0
        self.row_history = []              # rank dst: every row size the receive sets have had (growth is visible to tests)
        self.recv = self.host = None       # rank dst: allocated by _room() from the first byte counts seen
        self.d2h_copies = 0                # rank dst, device path: D2H copies issued so far (own part + one per run of landed parts)
        self._local_reads = None
        # control messages travel on the host, through the store: a device-side exchange would make the host wait for the side stream,
        # which shares one of the runtime's few hardware queues with a chunk stream of the decode (135 ms per step at config 3, r04)
        PackedGather._instances += 1
        self.store = _store() if (self.active and self.world > 1) else None
        # mailbox prefix: the n-th PackedGather of this process, and rank dst's GLOBAL rank -- two disjoint sub-groups that each build
        # their first PackedGather have different dst ranks, so their control messages cannot cross
        self.prefix = f"ft8rx_pg{PackedGather._instances}_d{self._g(dst) if self.active else 0}"
        if self.nccl:
            dev = torch.device("cuda", torch.cuda.current_device())
            self.stream = torch.cuda.Stream(device=dev)
            self.src = [torch.empty(self.cap, dtype=torch.uint8, device=dev) for _ in range(2)]
            ptrs = [t.data_ptr() for t in self.src]
            owners = list(self.src)
        else:
            # host path (gloo, or no process group at all): the pack kernels write page-locked host memory directly
            self.stream = None
            self._pin = [handle.pinned_bytes(self.cap) for _ in range(2)]
            self.src = [torch.from_numpy(a) for a in self._pin]
            ptrs = [a.ctypes.data for a in self._pin]
            owners = list(self._pin)
        self._set_output(ptrs[0], ptrs[1], self.cap, owners)
        # if this object is dropped without close(), the packed output is switched off (which waits for the batches in flight) before
        # the buffers go; the handle also holds references to them (Handle.set_packed_output keep=)
        self._fin = weakref.finalize(self, PackedGather._shutoff, weakref.ref(handle))
        self.next_ctrl = {}                # rank dst: peer -> batch number of the next control message expected from it
        if self.active and self.rank == self.dst:
            for r in range(self.world):
                if r != self.dst:
                    self.announced[r] = []
                    self.next_ctrl[r] = 0
        self._handshake()

    def _handshake(self):
        """One tiny blocking send / recv per (peer, dst) pair, peers in rank order -- the only rendezvous of the object, at construction.
        torch's NCCL backend creates a communicator per pair of ranks at their FIRST point-to-point call, and creating it is a
        rendezvous of the two (the later one blocks the earlier one's host thread).  submit() starts its send BEFORE it announces it and
        rank dst posts a receive only AFTER the announcement: left to the first batch, that first send would wait for a receive that is
        never posted.  Here both sides call unconditionally, so the pair's communicator exists before the first batch and every later
        isend / irecv is an asynchronous launch.  (gloo connects all pairs when the group is made; the same exchange runs there so that
        the CPU tests walk this code.)"""
        if not (self.active and self.world > 1):
            return
        dev = self.src[0].device if self.nccl else torch.device("cpu")
        token = torch.zeros(64, dtype=torch.uint8, device=dev)
        kw = {} if self.nccl else {"tag": self.TAG_HELLO}          # (the NCCL backend has no tags)
        if self.rank == self.dst:
            for r in range(self.world):
                if r != self.dst:
                    dist.recv(token, src=self._g(r), group=self.group, **kw)
        else:
            dist.send(token, dst=self._g(self.dst), group=self.group, **kw)
        if self.nccl:
            torch.cuda.current_stream().synchronize()

    # ---- plumbing
    def _set_output(self, p0, p1, cap, owners):
        try:
            self.h.set_packed_output(p0, p1, cap, keep=owners)
        except TypeError:                  # a stand-in handle without the keep= argument (CPU tests)
            self.h.set_packed_output(p0, p1, cap)

    @staticmethod
    def _shutoff(href):
        h = href()
        try:
            if h is not None and getattr(h, "_h", None) is not None and (not hasattr(h._h, "value") or h._h.value):
                h.set_packed_output(None, None, 0)
        except Exception:
            pass

    def _g(self, r):
        """global rank of group rank r"""
        return dist.get_global_rank(self.group, r) if self.group is not None else r

    def _wait(self, cond, what):
        """Poll until cond() -- progress is made by this thread only (_progress), so every wait goes through here."""
        import time
        t0 = time.perf_counter()
        spins = 0
        while True:
            self._progress()
            if cond():
                return
            spins += 1
            if spins > 50:
                time.sleep(50e-6)
            if time.perf_counter() - t0 > self.TIMEOUT:
                raise self._lib.Ft8rxError(f"PackedGather: timed out after {self.TIMEOUT:.0f} s waiting for {what} (rank {self.rank})")

    @staticmethod
    def _done(c):
        """Has this transfer completed?  A CUDA event is polled.  A gloo work object cannot be (it learns of its completion only in
        wait()), so it is waited for: that is only ever asked of a transfer whose other side is known to have been posted -- a receive
        whose announcement has arrived, or this rank's own send when its buffer is needed again."""
        if c is None:
            return True
        if hasattr(c, "query"):
            return c.query()
        c.wait()
        return True

    def _room(self, m):
        """Rank dst: `depth` receive sets with room for world x repeat parts of at least m bytes each.  Sized to twice the first part seen;
        a later part that needs more gets new, larger sets -- parts already received (or being received) keep their old memory alive.
        A set is ONE flat device buffer and ONE flat page-locked host buffer: a batch's parts are packed back to back in arrival order
        (256-byte aligned), so that one D2H copy can move all of a batch's received parts (_copy_landed)."""
        if m <= self.row:
            return
        self.row = min(self.cap, max(2 * m, self.MIN_ROW))
        self.row = (self.row + 255) & ~255
        self.row_history.append(self.row)
        n = self.world * self.repeat
        if self.nccl:
            dev = self.src[0].device
            self.recv = [torch.empty(n * self.row, dtype=torch.uint8, device=dev) for _ in range(self.depth)]
            self.host = [torch.empty(n * self.row, dtype=torch.uint8, pin_memory=True) for _ in range(self.depth)]
        else:
            self.host = [torch.empty((n, self.row), dtype=torch.uint8) for _ in range(self.depth)]

    def _batch(self, k):
        if k not in self.batches:
            self.batches[k] = {"parts": [None] * (self.world * self.repeat), "left": self.world * self.repeat, "err": None, "open": [],
                               "posted": 0, "seg": None, "items": []}
        return self.batches[k]

    def _copy_landed(self, b):
        """Device path, rank dst: move what has LANDED in device memory to the page-locked host buffer -- one D2H copy per run of
        neighbouring parts (all remote parts of a batch in one copy when the peers run in lockstep), issued through the handle on ITS
        result-copy stream (ft8rx_d2h_async).  Rounds 4-5 made a copy and an event per part on the torch side stream; the side stream
        shares a hardware queue with one of the decode streams, whose kernels then waited behind every copy: 12 MB per step -- rank
        dst's load in an 8-rank config-1 job -- cost 4.7 % of the step (profiles/r06_notes.md).  Nothing is copied while a receive of
        this batch is still in flight (it lands within the millisecond and extends the run); a peer that has not even announced its
        part holds nothing back."""
        items = b["items"]
        for it in items:
            if it["state"] == "receiving" and it["ev"].query():
                it["state"] = "landed"
        if any(it["state"] == "receiving" for it in items):
            return
        run = []
        for it in items + [None]:
            if it is not None and it["state"] == "landed" and (not run or (it["seg"] is run[-1]["seg"] and it["o"] == run[-1]["o"] + run[-1]["m"])):
                run.append(it)
                continue
            if run:
                seg, o0 = run[0]["seg"], run[0]["o"]
                n = run[-1]["o"] + run[-1]["m"] - o0
                t = self.h.d2h_async(seg["host"].data_ptr() + o0, seg["dev"].data_ptr() + o0, n)
                self.d2h_copies += 1
                for x in run:
                    x["state"], x["ticket"] = "copying", t
                run = []
            if it is not None and it["state"] == "landed":
                run = [it]

    def _set_free(self, k):
        """May parts of batch k be received now?  (a) Its receive set last held batch k - depth, which must be complete -- an uncollected
        one is dropped now, its memory is needed; (b) k <= own submits + depth - 3, so that what collect() handed out stays valid until
        rank dst itself has submitted two more batches, however far the peers run ahead."""
        old = k - self.depth
        if old >= self.done_upto or k > self.seq + self.depth - 3:
            return False
        while self.ready and self.ready[0][0] <= old:
            self.ready.pop(0)
        self.collected = max(self.collected, min(old + 1, self.done_upto))
        return True

    def _receive(self, r, k, nbytes, flag, slot=None):
        """Rank dst: post the data receive(s) of peer r's batch k (or take the local copy for r == dst).  Device path: the part gets its
        place in the batch's segment (flat buffer, parts back to back); _copy_landed() moves what has landed to the host."""
        b = self._batch(k)
        if flag:
            b["err"] = f"rank {r} reported a packed-buffer overflow for batch {k} ({nbytes} bytes needed; raise per_frame)"
            b["left"] -= self.repeat               # no data follows; this rank's parts stay None
            b["posted"] += self.repeat
            return
        m = (nbytes + 255) & ~255
        self._room(m)
        s = k % self.depth
        if self.nccl:
            seg = b["seg"]
            if seg is None or seg["dev"] is not self.recv[s]:       # first part of the batch, or the sets have just been replaced by larger ones
                seg = b["seg"] = {"dev": self.recv[s], "host": self.host[s], "hi": 0}
            for rep in range(self.repeat):
                i = rep * self.world + r
                o = seg["hi"]
                seg["hi"] = o + m
                it = {"i": i, "seg": seg, "o": o, "m": m, "nbytes": nbytes, "state": "receiving", "ev": None, "ticket": None}
                if r == self.dst and rep == 0:
                    # this rank's own part: straight from the pack buffer to the host buffer on the handle's copy stream -- no device
                    # copy, nothing on the side stream; the next pack into this buffer waits for the copy's event
                    it["ticket"] = self.h.d2h_async(seg["host"].data_ptr() + o, self.src[slot].data_ptr(), m)
                    it["state"] = "copying"
                    self.d2h_copies += 1
                    self.h.packed_fence(slot, self.h.d2h_event(it["ticket"]))
                else:
                    drow = seg["dev"][o:o + m]
                    with torch.cuda.stream(self.stream):
                        if r == self.dst:                               # repeat > 1 (measurement aid): a device copy stands in for a receive
                            drow.copy_(self.src[slot][:m], non_blocking=True)
                        else:
                            # a plain irecv per part, matching the peers' plain isend on the per-pair communicator _handshake created
                            # (dist.batch_isend_irecv would move the receives to the group's collective communicator, where a
                            # peer's plain isend never meets them)
                            dist.irecv(drow, src=self._g(r), group=self.group).wait()      # (wait = the side stream waits; the host does not)
                        it["ev"] = torch.cuda.Event()
                        it["ev"].record(self.stream)
                    if r == self.dst:
                        self._local_reads = it["ev"]
                b["items"].append(it)
            b["posted"] += self.repeat
            return
        for rep in range(self.repeat):
            i = rep * self.world + r
            hrow = self.host[s][i]
            if r == self.dst:
                hrow[:m].copy_(self.src[slot][:m])
                comp = None
            else:
                comp = dist.irecv(hrow[:m], src=self._g(r), group=self.group, tag=self.TAG_DATA)
            b["open"].append((i, comp, hrow, nbytes))
        b["posted"] += self.repeat
        return

    def _progress(self):
        """Rank dst: take the control messages that have arrived, post the data receives they announce (in order per peer, while the ring
        has room), note the parts that have landed, move completed batches to `ready` (in batch order).  Never blocks."""
        if not (self.active and self.rank == self.dst):
            return
        # in lockstep operation every peer's next announcement is there at the same time: one store round trip tells (check() is true
        # only if ALL keys exist); otherwise the peers are asked one by one
        peers = list(self.announced)
        all_there = len(peers) > 1 and self.store.check([f"{self.prefix}/{r}/{self.next_ctrl[r]}" for r in peers])
        for r, q in self.announced.items():
            first = True
            while True:
                key = f"{self.prefix}/{r}/{self.next_ctrl[r]}"
                if not (first and all_there) and not self.store.check([key]):
                    break
                first = False
                k, nbytes, flag = (int(x) for x in self.store.get(key).decode().split(","))
                self.store.delete_key(key)
                q.append((k, nbytes, flag))
                self.next_ctrl[r] += 1
            while q and self._set_free(q[0][0]):
                k, nbytes, flag = q.pop(0)
                self._receive(r, k, nbytes, flag)
        for k in sorted(self.batches):
            b = self.batches[k]
            if self.nccl:
                self._copy_landed(b)
                for it in b["items"]:
                    if it["state"] == "copying" and self.h.d2h_done(it["ticket"]):
                        it["state"] = "done"
                        b["parts"][it["i"]] = self._lib.Packed(it["seg"]["host"].numpy()[it["o"]:it["o"] + it["nbytes"]])
                        b["left"] -= 1
            still = []
            for i, comp, hrow, nbytes in b["open"]:
                if self._done(comp):
                    b["parts"][i] = self._lib.Packed(hrow.numpy()[:nbytes])
                    b["left"] -= 1
                else:
                    still.append((i, comp, hrow, nbytes))
            b["open"] = still
        while self.done_upto in self.batches and self.batches[self.done_upto]["left"] == 0:
            b = self.batches.pop(self.done_upto)
            self.ready.append((self.done_upto, b["parts"], b["err"]))
            self.done_upto += 1

    # ---- the API
    def close(self):
        """Drain, then switch the handle's packed output off."""
        self.drain()
        self.h.set_packed_output(None, None, 0)
        self._fin.detach()

    def submit(self):
        """Start the gather of the batch the last fetch returned (every rank calls this once per fetched batch).  Asynchronous: returns
        as soon as this rank's part is on its way (rank dst: as soon as its own part is copied and whatever has arrived is posted)."""
        import time
        t0 = time.perf_counter()
        ph = {}

        def mark(name, _t=[t0]):
            now = time.perf_counter()
            ph[name] = ph.get(name, 0.0) + now - _t[0]
            _t[0] = now
        slot, hdr = self.h.packed_results()
        if _DIAG == "header":              # measurement aid (tools/r06_gather_diag.sh): the pack kernels and the header read only
            self.seq += 1
            self.collected = self.done_upto = self.seq
            self.seconds.append(time.perf_counter() - t0)
            self.phases.append(ph)
            return
        overflow = bool(hdr["overflow"])
        nbytes = int(hdr["bytes"])
        mark("header")
        k = self.seq
        self.seq += 1
        self.phases.append(ph)
        if not self.active:
            if overflow:
                raise self._lib.Ft8rxError(f"PackedGather: a batch needs {nbytes} packed bytes, the buffers hold {self.cap} (raise per_frame)")
            self.ready.append((k, [self._lib.Packed(self._pin[slot][:nbytes])], None))
            del self.ready[:-2]
            self.seconds.append(time.perf_counter() - t0)
            return
        if self.rank != self.dst:
            # Only this rank's OWN sends are ever waited for, and only when `depth` of them are in flight.  The buffer of this slot is
            # safe without the host: on the device path the pack kernels of batch k + 2 wait for the send's event (packed_fence), on
            # the host path the bytes leave from a private copy.
            self.sends = [c for c in self.sends if not (hasattr(c, "query") and c.query())]
            while len(self.sends) >= self.depth:
                s0 = self.sends.pop(0)
                self._wait(lambda: self._done(s0), f"an earlier send to be taken by rank {self.dst}")
            mark("wait_slot")
            comp = None
            if not overflow:
                m = (nbytes + 255) & ~255
                if self.nccl:
                    with torch.cuda.stream(self.stream):
                        for rep in range(self.repeat):
                            dist.isend(self.src[slot][:m], dst=self._g(self.dst), group=self.group).wait()      # (the side stream waits, not the host)
                        comp = torch.cuda.Event()
                        comp.record(self.stream)
                        # the pack kernels that next write this slot's buffer (two enqueues from now) wait for the send on the device
                        self.h.packed_fence(slot, comp.cuda_event, keep=comp)
                else:
                    # (host path: nothing on the device orders the pack kernels of batch k + 2 behind a gloo send, so the bytes leave
                    # from a private copy -- the handle's buffer is free again when submit() returns)
                    stage = self.src[slot][:m].clone()
                    comp = _All([dist.isend(stage, dst=self._g(self.dst), group=self.group, tag=self.TAG_DATA) for rep in range(self.repeat)], keep=stage)
            mark("issue")
            # the announcement goes out AFTER the send has been started: when rank dst sees it, the matching receive cannot wait long
            self.store.set(f"{self.prefix}/{self.rank}/{k}", f"{k},{nbytes},{1 if overflow else 0}")
            mark("sizes")
            self.sends.append(comp)
            mark("issue")
            self.seconds.append(time.perf_counter() - t0)
            if overflow:
                raise self._lib.Ft8rxError(f"PackedGather: batch {k} needs {nbytes} packed bytes, the buffers hold {self.cap} (raise per_frame); "
                                           f"rank {self.dst} has been told")
            return
        # rank dst: its own part needs the receive set of batch k - depth back -- the only place it can wait for the slowest peer
        self._wait(lambda: self._set_free(k), f"batch {k - self.depth} to complete (a peer is {self.depth} batches behind)")
        mark("wait_slot")
        self._progress()
        mark("sizes")
        self._receive(self.dst, k, nbytes, 1 if overflow else 0, slot)
        if self.nccl and not overflow and self.repeat > 1:
            # (measurement aid) the stand-in copies read src[slot] on the side stream after the D2H was enqueued: fence on them instead
            ev = self._local_reads
            self.h.packed_fence(slot, ev.cuda_event, keep=ev)
        self._progress()
        mark("issue")
        self.seconds.append(time.perf_counter() - t0)
        if overflow:
            raise self._lib.Ft8rxError(f"PackedGather: batch {k} needs {nbytes} packed bytes, the buffers hold {self.cap} (raise per_frame)")

    def outstanding(self):
        """Batches submitted by this rank and not yet collected (in flight or completed)."""
        return self.seq - self.collected

    def _take(self):
        k, parts, err = self.ready.pop(0)
        self.collected = k + 1
        if err:
            raise self._lib.Ft8rxError("PackedGather: " + err)
        self._last = parts
        return parts

    def collect(self):
        """Rank dst: the per-rank Packed views of the OLDEST submitted batch not collected yet (valid until `depth` - 1 more batches have
        been submitted); waits until every rank's part of it has landed.  Other ranks: None, without waiting.  With nothing outstanding:
        the last result again."""
        if not self.active:
            if self.ready:
                return self._take()
            return self._last
        if self.rank != self.dst:
            self.collected = self.seq
            return None
        if self.collected >= self.seq:
            return self._last
        want = self.collected
        self._wait(lambda: self.done_upto > want or self.collected > want, f"batch {want} from every rank")
        if self.collected > want or not self.ready:        # it was dropped meanwhile (its receive set was needed): the oldest one still there
            return self._take() if self.ready else self._last
        return self._take()

    def drain(self):
        """Wait for everything this rank has submitted: its sends taken (other ranks) / every rank's part of every batch up to this
        rank's last submit landed (rank dst); -> the last batch's result (as collect)."""
        if not self.active:
            while self.ready:
                self._take()
            return self._last
        if self.rank != self.dst:
            while self.sends:
                s0 = self.sends.pop(0)
                self._wait(lambda: self._done(s0), f"the send of a batch to be taken by rank {self.dst}")
            self.collected = self.seq
            return None
        self._wait(lambda: self.done_upto >= self.seq, f"batch {self.seq - 1} from every rank")
        while self.ready:
            self._take()
        return self._last


class _All:
    """several gloo sends as one"""
    def __init__(self, items, keep=None):
        self.items = [i for i in items if i is not None]
        self.keep = keep

    def wait(self):
        for i in self.items:
            i.wait()
        self.items = []
