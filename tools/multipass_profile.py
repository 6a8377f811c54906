"""Where the time of a two-pass decode goes (GPU box): python tools/multipass_profile.py [frames]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyft8_amd import _lib  # noqa: E402
from pyft8_amd.receiver import Receiver  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    rx = Receiver("x", None)
    h = rx._handle(B)
    h.synth_frames(h.staging_ptr(), 4242, B, n_signals=50, snr_range=(-10.0, 10.0))
    audio = h.download_audio(h.staging_ptr(), B)
    for _ in range(2):
        rx.decode_frames_arrays(audio, passes=2)
    T = {}

    def tick(name, t0):
        T[name] = T.get(name, 0.0) + time.perf_counter() - t0

    n = 5
    for _ in range(n):
        t0 = time.perf_counter(); rec, cnt, ev, evc = h.decode_batch(audio); tick("pass 1: decode_batch (H2D + kernels + D2H)", t0)
        t0 = time.perf_counter(); msgs, mcnt = _lib.package_batch(rec, cnt, ev, evc); tick("pass 1: package_batch", t0)
        t0 = time.perf_counter(); sigs = rx._subtraction_list(msgs, mcnt, rec, -10); tick("subtraction list (native: ft8rx_subtraction_list)", t0)
        t0 = time.perf_counter(); h.subtract(h.staging_ptr(), B, sigs, refine=rx.subtract_refine); tick("ft8rx_subtract (refine mode %d + subtract, %d sequential signals)" % (rx.subtract_refine, int(sigs[1].max())), t0)
        t0 = time.perf_counter(); h.enqueue(h.staging_ptr(), B); r2 = h.fetch(B); tick("pass 2: kernels + D2H", t0)
        t0 = time.perf_counter(); m2, c2 = _lib.package_batch(*r2); tick("pass 2: package_batch", t0)
    t0 = time.perf_counter()
    for _ in range(n):
        rx.decode_frames_arrays(audio, passes=2)
    tot = (time.perf_counter() - t0) / n
    for k, v in T.items():
        print(f"{1e3 * v / n:8.2f} ms  {k}")
    print(f"{1e3 * sum(T.values()) / n:8.2f} ms  sum of the parts; decode_frames_arrays(passes=2) end to end: {1e3 * tot:.2f} ms per {B} frames = {B / tot:.0f} frames/s")


if __name__ == "__main__":
    main()
