"""Test helpers: turn the oracle's whole-frame output into the product's record/event arrays."""
import numpy as np

import oracle as O
from pyft8_amd import _lib


def records_from_oracle(r, max_cands=200):
    rec = np.zeros(max_cands, _lib.RECORD_DTYPE)
    for i, c in enumerate(r["cands"]):
        g = rec[i]
        g["msg_lo"], g["msg_hi"] = c.msg_lo, c.msg_hi
        g["score"], g["grid_sd"], g["fine_sd"] = c.score, c.grid_sd, c.fine_sd
        g["f0_idx"], g["h0_idx"] = c.f0_idx, c.h0_idx
        g["ttweak"], g["ftweak"], g["snr_grid"], g["snr_fine"] = c.ttweak, c.ftweak, c.snr_grid, c.snr_fine
        g["status"] = c.status
        g["ipass"] = c.ipass if c.ipass >= 0 else 255
        g["ap"], g["method"], g["n_its"], g["nsync"] = c.ap, c.method, c.n_its, c.nsync
    ev = np.zeros(max(1, len(r["events"])), _lib.EVENT_DTYPE)
    for i, e in enumerate(r["events"]):
        ev[i]["msg_lo"], ev[i]["msg_hi"] = e.msg_lo, e.msg_hi
        ev[i]["cand"], ev[i]["ipass"] = e.cand, e.ipass
        ev[i]["slot"], ev[i]["seq"], ev[i]["valid"] = (e.pad >> 16) & 0xff, e.pad & 0xffff, e.valid
    return rec, len(r["cands"]), ev, len(r["events"])


def oracle_frame(audio):
    return O.decode_frame(audio, O.default_config(**_lib.fft_plans()))
