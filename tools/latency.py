"""Host-to-host latency of one decode call (ft8rx_decode_batch: H2D + kernels + D2H + native message layer) for small
batches -- the live-receiver case (one 15-s frame per cycle).  Usage (GPU box): python tools/latency.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyft8_amd import _lib, synth  # noqa: E402


def main():
    sizes = (1, 2, 8, 32, 64, 128)
    frames = synth.make_batch(777000, max(sizes))
    print("frames  streams  ladder mode   ms/call (median of 20)   frames/s")
    for B in sizes:
        for ns in (1, 2):
            for mode, mname in ((0, "ladder order"), (1, "one launch")):
                h = _lib.Handle(max_frames=B)
                h.set_streams(ns)
                h.set_ladder_mode(mode)
                a = frames[:B]
                for _ in range(3):
                    _lib.package_batch(*h.decode_batch(a))
                ts = []
                for _ in range(20):
                    t0 = time.perf_counter()
                    _lib.package_batch(*h.decode_batch(a), n_threads=1 if B == 1 else None)
                    ts.append(time.perf_counter() - t0)
                ms = 1e3 * float(np.median(ts))
                print(f"{B:6d} {ns:8d}  {mname:12s} {ms:12.3f} {B / ms * 1e3:22.0f}")
                h.close()


if __name__ == "__main__":
    main()
