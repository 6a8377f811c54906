"""Where the host side of the pipelined decode loop spends its time (GPU box):  python tools/host_breakdown.py [frames] [steps]
Per step: enqueue (launches), fetch (wait for the previous batch + result hand-over), package (native host message layer)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyft8_amd import _lib  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    h = _lib.Handle(max_frames=B)
    ptr = h.staging_ptr()
    h.synth_frames(ptr, 0, B)
    for nt in (32, 16, 8, 4):
        te = tf = tp = 0.0
        h.enqueue(ptr, B); h.enqueue(ptr, B); h.fetch_view(B); h.fetch_view(B); h.sync()
        t0 = time.perf_counter()
        for i in range(steps):
            a = time.perf_counter()
            h.enqueue(ptr, B)
            b = time.perf_counter()
            if i > 0:
                v = h.fetch_view(B)
                c = time.perf_counter()
                _lib.package_batch(*v, n_threads=nt)
                d = time.perf_counter()
                tf += c - b; tp += d - c
            te += b - a
        v = h.fetch_view(B); _lib.package_batch(*v, n_threads=nt)
        h.sync()
        dt = time.perf_counter() - t0
        print(f"B={B} threads={nt:2d}: {B * steps / dt:8.0f} frames/s, {1e3 * dt / steps:.3f} ms/step | enqueue {1e3 * te / steps:.3f}  fetch(wait+handover) {1e3 * tf / steps:.3f}  package {1e3 * tp / steps:.3f} ms")
    # the GPU alone
    h.sync(); t0 = time.perf_counter()
    for i in range(steps):
        h.enqueue(ptr, B)
    t1 = time.perf_counter(); h.sync(); dt = time.perf_counter() - t0
    print(f"kernels only: {1e3 * dt / steps:.3f} ms/step; host time to enqueue one batch {1e3 * (t1 - t0) / steps:.3f} ms")
    # package alone on one fetched batch
    v = h.fetch(B)
    for nt in (64, 48, 32, 16, 8, 4, 1):
        _lib.package_batch(*v, n_threads=nt)
        t0 = time.perf_counter()
        for _ in range(20):
            _lib.package_batch(*v, n_threads=nt)
        print(f"package_batch alone, {nt:2d} threads: {1e3 * (time.perf_counter() - t0) / 20:.3f} ms per {B} frames")
    v = h.fetch_view(B)
    t0 = time.perf_counter()
    for _ in range(20):
        _lib.package_batch(*v, n_threads=8)
    print(f"package_batch on the page-locked view, 8 threads: {1e3 * (time.perf_counter() - t0) / 20:.3f} ms")


if __name__ == "__main__":
    main()
