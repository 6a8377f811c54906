"""Decode yield of the subtraction extension on device-generated frames with known truth.
Usage (GPU box): python tools/two_pass_yield.py [n_frames] [n_signals]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyft8_amd.receiver import Receiver  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    nsig = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    rx = Receiver("", None, max_frames=n)
    h = rx._handle(n)
    ptr = h.staging_ptr()
    truth = h.synth_frames(ptr, 8100000, n, n_signals=nsig, snr_range=(-10.0, 10.0))
    audio = h.download_audio(ptr, n)
    want = [{t["msg"] for t in truth[f]} for f in range(n)]
    print(f"{n} frames x {nsig} signals, -10..+10 dB")
    print("passes  true decodes/frame  false/frame   s per batch")
    modes = [int(a) for a in sys.argv[3:]] or [rx.subtract_refine]
    for mode in modes:
      rx.subtract_refine = mode
      print(f"-- origin refinement mode {mode} (1 = full-rate scans, 2 = decimated baseband)")
      for passes, osd in ((1, True), (2, True), (3, True), (2, False), (3, False)):
        if passes == 1 and mode != modes[0]:
            continue
        rx.decode_frames(audio[:2], passes=passes)
        t0 = time.perf_counter()
        out = rx.decode_frames(audio, passes=passes, sub_pass_osd=osd)
        dt = time.perf_counter() - t0
        got = [{" ".join(d["msg_tuple"]) for d in out[f]} for f in range(n)]
        true = sum(len(got[f] & want[f]) for f in range(n)) / n
        false = sum(len(got[f] - want[f]) for f in range(n)) / n
        print(f"{passes:6d} {true:19.2f} {false:12.2f} {dt:12.3f}" + ("" if osd else "   (no OSD decodes accepted in passes > 1)"))
      # the reference experiment's local re-search (receiver_sub.py:434-445), batched: columns f0 - 2 .. f0 + 1 of the subtracted signals only
      for osd in (True, False):
        t0 = time.perf_counter()
        out = rx.decode_frames(audio, passes=2, sub_pass_osd=osd, research="local")
        dt = time.perf_counter() - t0
        got = [{" ".join(d["msg_tuple"]) for d in out[f]} for f in range(n)]
        true = sum(len(got[f] & want[f]) for f in range(n)) / n
        false = sum(len(got[f] - want[f]) for f in range(n)) / n
        print(f"{2:6d} {true:19.2f} {false:12.2f} {dt:12.3f}   (local re-search, sync threshold ignored" + (")" if osd else "; no OSD decodes accepted in pass 2)"))


if __name__ == "__main__":
    main()
