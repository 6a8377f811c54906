"""Drop-in surface of PyFT8/decoders.py for the MI355X build.

Same names, argument meaning and return shapes as the reference functions; the arithmetic runs in the
HIP kernels of libft8rx.so (one wavefront per vector), only string rendering happens on the host:

    ldpc_decode(llr, max_ncheck0, max_iters) -> (msg_tuple | None, n_its, [] | llr)   reference decoders.py:153-171
    osd_012(llr, singleflips=30, doubleflips=2) -> msg_tuple | None                    reference decoders.py:223-272
    crc_unpack91(codeword91) -> msg_tuple | None                                       reference decoders.py:117-131
    unpack(bits77_int) -> msg_tuple | None                                             reference decoders.py:16-49

`call_hashes` is the process-global hash table the reference keeps in databases.py:8.
Batched variants (`*_batch`) take [n,174] arrays.
"""
import numpy as np

from . import _lib
from . import messages as _m

call_hashes = _m.CallHashes()


def add_call_hashes(call):
    """reference databases.py:10-26"""
    call_hashes.add(call)


def unpack(bits):
    return _m.unpack(int(bits), call_hashes)


def _msg(lo, hi):
    return (int(hi) << 64) | int(lo)


def ldpc_decode_batch(llr, max_ncheck0, max_iters):
    h = _lib.default_handle()
    return h.ldpc(llr, max_ncheck0, max_iters)


def ldpc_decode(llr, max_ncheck0, max_iters):
    """Like the reference, mutates `llr` in place when belief propagation ran without a decode."""
    ok, lo, hi, nits, has, out = ldpc_decode_batch(np.asarray(llr, np.float32)[None], max_ncheck0, max_iters)
    if ok[0]:
        return unpack(_msg(lo[0], hi[0])), int(nits[0]), []
    if not has[0]:
        return None, -1, []
    try:
        llr[:] = out[0]
        return None, -1, llr
    except (TypeError, ValueError):
        return None, -1, out[0]


def osd_012_batch(llr, singleflips=30, doubleflips=2):
    return _lib.default_handle().osd(llr, singleflips, doubleflips)


def osd_012(llr, singleflips=30, doubleflips=2):
    """reference decoders.py:223-272.  Like the reference, any singleflips / doubleflips up to the 91 basis positions (its own callers
    use 30 / 2, its subtraction experiment 40 / 1); beyond 91 the reference indexes past its flip list, here Ft8rxError."""
    ok, lo, hi, trial = osd_012_batch(np.asarray(llr, np.float32)[None], singleflips, doubleflips)
    if ok[0]:
        return unpack(_msg(lo[0], hi[0]))
    return None


def crc_unpack91(codeword91):
    res, lo, hi = _lib.default_handle().crc_valid(np.asarray(codeword91, np.float32)[None, :91])
    if res[0] >= 1:
        return unpack(_msg(lo[0], hi[0]))      # res == 1: CRC matched but unpack() yields None (side effects kept)
    return None
