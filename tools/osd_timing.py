"""Where the cycles of an OSD attempt go (timing-only build of libft8rx.so with -DOSD_TIMING, see kernels/osd.hpp): lane 0 of every
attempt accumulates shader cycles between marks.  Usage on the GPU box:
    python -c "from pyft8_amd import _lib; _lib.build_variant('build/ab/osd_timing.so', ['-DOSD_TIMING'])"
    FT8RX_LIB=build/ab/osd_timing.so python tools/osd_timing.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyft8_amd import _lib  # noqa: E402

NAMES = ["LLR loads, AP override, sort keys", "np.argsort network (36 stages, keys in registers)", "generator columns in sorted order (d_G0T loads), hard decisions",
         "Gauss-Jordan: visited columns until 91 are accepted", "flip rows published, hard-decision mask", "per-column flip words (nflip broadcast reads), un-permutation",
         "CRC syndromes of the order-0 codeword and the flips", "trials (+ slow path of zero-syndrome trials), result"]


def main():
    B = 256
    h = _lib.Handle(max_frames=B)
    ptr = h.staging_ptr()
    h.synth_frames(ptr, 0, B, n_signals=50)
    h.set_streams(1)
    h.enqueue(ptr, B); h.sync()
    L = _lib.lib()
    L.ft8rx_debug_osd_times.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    out = np.zeros(16, np.uint64)
    L.ft8rx_debug_osd_times(h._h, None, 1)
    h.enqueue(ptr, B); h.sync()
    L.ft8rx_debug_osd_times(h._h, out.ctypes.data, 0)
    h.fetch(B)
    tot = float(out[:8].sum())
    n = int(out[15])
    print(f"k_osd, {B} frames: {n} attempts that ran; shader cycles of lane 0 summed over all attempts (share; cycles per attempt)")
    for i, nm in enumerate(NAMES):
        print(f"  {i} {nm:<90s} {int(out[i]):>16,d}  {100 * out[i] / tot:5.1f} %  {out[i] / max(n, 1):9.0f}")
    print(f"  visited columns per attempt: {out[8] / max(n, 1):.1f}")
    h.close()


if __name__ == "__main__":
    main()
