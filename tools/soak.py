"""Soak test (GPU box): many decode calls with varying batch sizes, stream counts and handles; results must stay identical to the
first pass and device/host memory must not grow.  Usage: python tools/soak.py [seconds]"""
import os
import resource
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyft8_amd import _lib, synth  # noqa: E402


def digest(res):
    rec, cnt, ev, evc = res
    out = []
    for f in range(len(cnt)):
        r = rec[f, :cnt[f]]
        out.append((int(cnt[f]), int(evc[f]), r["status"].tobytes(), r["msg_lo"].tobytes(), r["msg_hi"].tobytes(), r["ipass"].tobytes(),
                    tuple(sorted(ev[f, :min(evc[f], _lib.EVENT_CAP)].tolist()))))
    return out


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    frames = synth.make_batch(424242, 48)
    h = _lib.Handle(max_frames=48)
    ref = digest(h.decode_batch(frames))
    rng = np.random.default_rng(1)
    t0 = time.time()
    n = 0
    rss0 = None
    dev = None
    trend = []
    while time.time() - t0 < budget:
        B = int(rng.choice([1, 2, 7, 8, 16, 33, 48]))
        ns = int(rng.choice([1, 2, 4, 8]))
        start = int(rng.integers(0, 48 - B + 1))
        if rng.random() < 0.1:
            h.close()
            h = _lib.Handle(max_frames=48)
        h.set_streams(ns)
        mode = int(rng.integers(0, 4))
        if mode == 3:                                   # pipelined host entry: two batches in flight from host memory (pageable here)
            s2 = int(rng.integers(0, 48 - B + 1))
            a1, a2 = np.ascontiguousarray(frames[start:start + B]), np.ascontiguousarray(frames[s2:s2 + B])
            h.enqueue_host(a1)
            h.enqueue_host(a2)
            assert digest(h.fetch(B)) == ref[start:start + B], ("host pipelined", B, ns, start)
            assert digest(h.fetch(B)) == ref[s2:s2 + B], ("host pipelined 2", B, ns, s2)
        elif mode == 0:                                 # synchronous host-pointer entry
            got = digest(h.decode_batch(frames[start:start + B]))
            assert got == ref[start:start + B], (B, ns, start)
        else:                                           # device-resident, pipelined: two batches in flight, fetched oldest first
            if dev is None:
                dev = _lib.Handle(max_frames=48)        # its staging buffer holds the frames on the device
                dev.decode_batch(frames)
            base = dev.staging_ptr()
            s2 = int(rng.integers(0, 48 - B + 1))
            h.enqueue(base + start * _lib.NSAMP * 2, B)
            h.enqueue(base + s2 * _lib.NSAMP * 2, B)
            assert digest(h.fetch(B)) == ref[start:start + B], ("pipelined", B, ns, start)
            if mode == 1:
                assert digest(h.fetch(B)) == ref[s2:s2 + B], ("pipelined 2", B, ns, s2)
        n += 1
        if n == 50:
            rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
        if n % 5000 == 0:                              # the trend, not only the end points: a leak grows linearly, an allocator's high-water mark does not
            with open("/proc/self/statm") as f:
                trend.append((n, int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") // 1024))
    rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    with open("/proc/self/statm") as f:
        cur = int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") // 1024
    print(f"{n} decode calls in {time.time() - t0:.0f} s, all identical to the first pass; max RSS after 50 calls {rss0} kB, at the end {rss1} kB (resident now: {cur} kB)")
    print("resident kB every 5000 calls:", " ".join(f"{k}" for _, k in trend))
    assert rss0 is None or rss1 < rss0 * 1.2 + 50000


if __name__ == "__main__":
    main()
