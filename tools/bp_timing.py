"""Where the cycles of a BP attempt go (timing-only build of libft8rx.so with -DBP_TIMING, see kernels/bp.hpp): lane 0 of every
attempt adds the shader cycles between marks.  All four k_bp launches of a batch together.  Usage on the GPU box:
    python -c "from pyft8_amd import _lib; _lib.build_variant('build/ab/bp_timing.so', ['-DBP_TIMING'])"
    FT8RX_LIB=build/ab/bp_timing.so python tools/bp_timing.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyft8_amd import _lib  # noqa: E402

NAMES = ["LLR loads, AP override, GOOD91 of the fine stage, check masks", "per iteration: parity of every check, stop tests (+ CRC when all checks hold)",
         "per iteration: edge tables (first one only), nine tanh per lane", "per iteration: barrier, check products", "per iteration: barrier, nine messages per lane (one division each)",
         "per iteration: barrier, variable updates, barrier", "result, saved LLRs"]


def main():
    B = 256
    h = _lib.Handle(max_frames=B)
    ptr = h.staging_ptr()
    h.synth_frames(ptr, 0, B, n_signals=50)
    h.set_streams(1)
    h.enqueue(ptr, B); h.sync()
    L = _lib.lib()
    L.ft8rx_debug_bp_times.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    out = np.zeros(8, np.uint64)
    L.ft8rx_debug_bp_times(h._h, None, 1)
    h.enqueue(ptr, B); h.sync()
    L.ft8rx_debug_bp_times(h._h, out.ctypes.data, 0)
    h.fetch(B)
    tot = float(out[:7].sum())
    print(f"k_bp, {B} frames, the four launches of a batch: shader cycles of lane 0 summed over all attempts (share)")
    for i, nm in enumerate(NAMES):
        print(f"  {i} {nm:<80s} {int(out[i]):>16,d}  {100 * out[i] / tot:5.1f} %")
    h.close()


if __name__ == "__main__":
    main()
