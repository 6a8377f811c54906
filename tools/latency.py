"""Host-to-host latency of one decode call (ft8rx_decode_batch: H2D + kernels + D2H + native message layer) for small
batches -- the live-receiver case (one 15-s frame per cycle).  Usage (GPU box): python tools/latency.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyft8_amd import _lib, synth  # noqa: E402


def main():
    frames = synth.make_batch(777000, 32)
    print("frames  streams   ms/call (median of 20)   frames/s")
    for B in (1, 2, 8, 32):
        for ns in (1, 4):
            h = _lib.Handle(max_frames=B)
            h.set_streams(ns)
            a = frames[:B]
            for _ in range(3):
                _lib.package_batch(*h.decode_batch(a))
            ts = []
            for _ in range(20):
                t0 = time.perf_counter()
                _lib.package_batch(*h.decode_batch(a), n_threads=1 if B == 1 else None)
                ts.append(time.perf_counter() - t0)
            ms = 1e3 * float(np.median(ts))
            print(f"{B:6d} {ns:8d} {ms:12.3f} {B / ms * 1e3:22.0f}")
            h.close()


if __name__ == "__main__":
    main()
