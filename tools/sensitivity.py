"""Decode probability vs SNR of the GPU receive path on device-generated frames with known truth
(BASELINE config-4 style: few signals per frame so collisions do not dominate).
Usage (GPU box): python tools/sensitivity.py [n_frames] -> table on stdout"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyft8_amd import _lib, messages as M  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    nsig = 8
    h = _lib.Handle(max_frames=n)
    ptr = h.staging_ptr()
    truth = h.synth_frames(ptr, 3000000, n, n_signals=nsig, snr_range=(-26.0, -6.0))
    h.enqueue(ptr, n)
    rec, cnt, ev, evc = h.fetch(n)
    bins = np.arange(-26, -5, 2)
    tot, hit = np.zeros(len(bins) - 1), np.zeros(len(bins) - 1)
    false = 0
    for f in range(n):
        got = {" ".join(m["msg_tuple"]) for m in M.package_frame(rec[f], int(cnt[f]), ev[f], int(evc[f]))}
        want = {t["msg"] for t in truth[f]}
        false += len(got - want)
        for t in truth[f]:
            b = int(np.searchsorted(bins, t["snr"], side="right") - 1)
            if 0 <= b < len(tot):
                tot[b] += 1
                hit[b] += t["msg"] in got
    print(f"{n} frames x {nsig} signals, Receiver defaults; false decodes: {false} ({false / n:.2f} per frame)")
    print("SNR bin (dB)   signals  decoded  P(decode)")
    for i in range(len(tot)):
        print(f"[{bins[i]:+3d},{bins[i + 1]:+3d})   {int(tot[i]):7d}  {int(hit[i]):7d}  {hit[i] / max(1, tot[i]):8.3f}")


if __name__ == "__main__":
    main()
