"""Where the cycles of a fine-sync candidate go (timing-only build of libft8rx.so with -DFINE_TIMING, see kernels/fine_sync.hpp):
wave 0 of every k_fine block accumulates shader cycles between marks.  Usage on the GPU box:
    python -c "from pyft8_amd import _lib; _lib.build_variant('build/ab/fine_timing.so', ['-DFINE_TIMING'])"
    FT8RX_LIB=build/ab/fine_timing.so python tools/fine_timing.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyft8_amd import _lib  # noqa: E402

NAMES = ["(outside fft: scoring tail of the previous step, loop)", "stage 1: slice loads, radix-8, twiddles, LDS stores", "barrier", "stage 2: LDS loads",
         "barrier", "stage 2: [4,4] butterflies + LDS stores", "barrier", "stage 3: LDS loads", "barrier", "stage 3: [5,5] (pruned) + LDS stores",
         "barrier", "  step 1: H stores", "  barrier after step 1",
         "LLRs of the candidates that pass the gate (41 %), records", "  step 2: DPP row sums, magnitudes", "  (loop)",
         "  step 1: slice + phase loads, taper, 10 complex multiplies", "  step 1: the 70 complex-by-real multiply-adds (K in registers)", "phases of the frequency scan (table loads after the time tweak is known), (cos, sin) table, barrier",
         "  step 2: the multiply-adds (H, cos / sin loads)", "scores of the eight tweaks (after the scan), first maximum",
         "final grid: phases + H for eight tones (incl. its barrier)", "final grid: first round of 10-point transforms + twiddles, barrier", "final grid: second round, magnitudes; clamped rows (30 % of the candidates); barrier",
         "Costas gate", "time scan: requests of the scan's constants, 56 symbol DFTs on lane quads, (on, off) sums", "time scan: barrier, block scores, first maximum"]


def main():
    B = 256
    h = _lib.Handle(max_frames=B)
    ptr = h.staging_ptr()
    h.synth_frames(ptr, 0, B, n_signals=50)
    h.set_streams(1)
    h.enqueue(ptr, B); h.sync()
    L = _lib.lib()
    L.ft8rx_debug_fine_times.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    out = np.zeros(32, np.uint64)
    L.ft8rx_debug_fine_times(h._h, None, 1)
    h.enqueue(ptr, B); h.sync()
    L.ft8rx_debug_fine_times(h._h, out.ctypes.data, 0)
    rec, cnt, ev, evc = h.fetch(B)
    tot = float(out[:27].sum())
    print(f"k_fine, {B} frames: cycles of wave 0 summed over all blocks (share of the total)")
    for i, n in enumerate(NAMES):
        print(f"  {i:2d} {n:<58s} {int(out[i]):>16,d}  {100 * out[i] / tot:5.1f} %")
    h.close()


if __name__ == "__main__":
    main()
