"""Decode probability vs SNR of the GPU receive path on device-generated frames with known truth
(BASELINE config-4 style: few signals per frame so collisions do not dominate), for a set of decoder knob settings:
the reference's osd_012(30, 2), the build's OSD order-3 extension without and with the Hamming-distance acceptance gate
(ft8rx_config.osd_triple / osd_max_hd).  Also prints the distance histogram of true vs false OSD decodes (what the gate cuts).
Usage (GPU box): python tools/sensitivity.py [n_frames] [snr_lo snr_hi] -> table on stdout"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyft8_amd import _lib  # noqa: E402


def run(n, nsig, snr, kw, bins):
    cfg = _lib.default_config(**kw)
    h = _lib.Handle(cfg, max_frames=n)
    ptr = h.staging_ptr()
    truth = h.synth_frames(ptr, 3000000, n, n_signals=nsig, snr_range=snr)
    h.enqueue(ptr, n)
    h.sync()
    t0 = time.perf_counter()
    h.enqueue(ptr, n)
    rec, cnt, ev, evc = h.fetch(n)
    dt = time.perf_counter() - t0
    msgs, mcnt, flags = _lib.package_batch(rec, cnt, ev, evc, return_flags=True)
    tot, hit = np.zeros(len(bins) - 1), np.zeros(len(bins) - 1)
    false = 0
    hd_true, hd_false = [], []
    for f in range(n):
        want = {t["msg"] for t in truth[f]}
        got = set()
        for m in msgs[f, :mcnt[f]]:
            txt = b" ".join(m["f"]).decode()
            got.add(txt)
            if m["method"] in (_lib.M_OSD, _lib.M_LDPC_B_OSD):
                (hd_true if txt in want else hd_false).append(int(rec[f, m["cand"]]["osd_hd"]))
        false += len(got - want)
        for t in truth[f]:
            b = int(np.searchsorted(bins, t["snr"], side="right") - 1)
            if 0 <= b < len(tot):
                tot[b] += 1
                hit[b] += t["msg"] in got
    h.close()
    return tot, hit, false, dt, hd_true, hd_false, int((flags != 0).sum())


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    snr = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (-26.0, -6.0)
    nsig = 8
    bins = np.arange(int(np.floor(snr[0])), int(np.ceil(snr[1])) + 1, 2 if snr[1] - snr[0] > 8 else 1)
    settings = [("reference knobs: osd_012(30, 2)", {}),
                ("order 3 over 30 positions, reference acceptance (first CRC-valid trial)", dict(osd_triple=30)),
                ("order 3 over 30 positions + distance gate hd <= 36", dict(osd_triple=30, osd_max_hd=36)),
                ("order 3 over 30 positions + distance gate hd <= 32", dict(osd_triple=30, osd_max_hd=32)),
                ("order 3 over 30 positions + distance gate hd <= 28", dict(osd_triple=30, osd_max_hd=28)),
                ("reference orders + distance gate hd <= 32", dict(osd_max_hd=32))]
    res = [(name, run(n, nsig, snr, kw, bins)) for name, kw in settings]
    print(f"{n} device-generated frames x {nsig} signals, SNR uniform in [{snr[0]:+.0f}, {snr[1]:+.0f}] dB / 2500 Hz; truth known")
    for name, (tot, hit, false, dt, hd_t, hd_f, nflag) in res:
        print(f"\n== {name}: {hit.sum() / max(1, tot.sum()):.4f} of all signals decoded, {false} false decodes ({false / n:.3f} per frame), "
              f"{n / dt:.0f} frames/s, {nflag} frames with an overflowed event log")
        print("   SNR bin (dB)   signals  decoded  P(decode)")
        for i in range(len(tot)):
            print(f"   [{bins[i]:+3d},{bins[i + 1]:+3d})   {int(tot[i]):7d}  {int(hit[i]):7d}  {hit[i] / max(1, tot[i]):8.3f}")
        if hd_t or hd_f:
            q = lambda a: ("-" if not a else " ".join(f"{int(v)}" for v in np.percentile(a, [5, 25, 50, 75, 95])))      # noqa: E731
            print(f"   OSD decodes: {len(hd_t)} true (distance pct 5/25/50/75/95: {q(hd_t)}), {len(hd_f)} false ({q(hd_f)})")


if __name__ == "__main__":
    main()
