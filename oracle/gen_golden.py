"""Generate tests/golden/* by running the real PyFT8 reference in THIS container.

    python oracle/gen_golden.py

Outputs (committed; the reference itself never travels):
  tests/golden/test_08.wav, test_09.wav   -- the reference's own fixture audio (data files,
                                             reference: tests/pipeline/test_0[89].wav)
  tests/golden/<frame>.npz + <frame>.json -- per-stage vectors captured from the reference
                                             through oracle/ref_harness.py
Frames: the two wavs, synthetic frame 0 (50 signals, -10..+10 dB; BASELINE config 1 recipe),
synthetic low-SNR frame (seed index 100000, 50 signals at -21..-11 dB; exercises OSD), and a
sparse frame (index 200000, 6 signals, -18..-8 dB).
"""
import json
import os
import shutil
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from ref_harness import run_frame, read_wav_i16, REF_ROOT  # noqa: E402
from pyft8_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def msg_str(r):
    return None if r is None else " ".join(r)


def dump(name, audio, meta):
    cands, tr, rx = run_frame(audio)
    n = len(cands)
    grid = rx.audio_in.search_grid
    rows = np.array([1, 2, 7, 8, 9, 50, 111, 187, 188, 258, 300, 374, 375])
    arrs = dict(
        grid_rows_idx=rows, grid_rows=grid[rows].copy(),
        row0=grid[0].copy(), row376=grid[376].copy(),
        f0_idx=np.array([f["f0_idx"] for f in tr.final], dtype=np.int32),
        h0_idx=np.array([f["h0_idx"] for f in tr.final], dtype=np.int32),
        score=np.array([f["score"] for f in tr.final], dtype=np.float64),
    )
    npay = n if meta["kind"] == "wav" else min(n, 40)
    arrs["payload"] = np.stack([c.payload_on_search_grid for c in cands[:npay]]).astype(np.float32) \
        if npay else np.zeros((0, 58, 8), np.float32)
    gl = np.full((n, 174), np.nan, np.float32)
    gsd = np.zeros(n, np.float64)
    gsnr = np.zeros(n, np.int32)
    for i, g in tr.grid.items():
        if g["llr"] is not None:
            gl[i] = g["llr"]
        gsd[i], gsnr[i] = g["sd"], g["snr"]
    arrs.update(grid_llr=gl, grid_sd=gsd, grid_snr=gsnr)
    # cycle spectrum samples
    spec = rx.audio_in.get_cycle_spectrum()
    sidx = np.unique(np.concatenate([np.arange(1400, 48900, 61), np.arange(11000, 12100),
                                     np.arange(30000, 31100), [0, 1, 2, 96000]]))
    arrs.update(spec_idx=sidx.astype(np.int32), spec_val=spec[sidx].astype(np.complex64))
    # fine
    fi = sorted(tr.fine.keys())
    arrs["fine_idx"] = np.array(fi, dtype=np.int32)
    arrs["fine_tt"] = np.array([int(tr.fine[i]["tweaks"].split()[0][2:]) for i in fi], dtype=np.int32)
    arrs["fine_ft"] = np.array([int(tr.fine[i]["tweaks"].split()[1][2:]) for i in fi], dtype=np.int32)
    arrs["fine_nsync"] = np.array([tr.fine[i]["n_sync"] for i in fi], dtype=np.int32)
    arrs["fine_stopped"] = np.array([tr.fine[i]["stopped"] for i in fi], dtype=np.bool_)
    arrs["fine_sd"] = np.array([tr.fine[i]["sd"] for i in fi], dtype=np.float64)
    arrs["fine_snr"] = np.array([tr.fine[i]["snr"] for i in fi], dtype=np.int32)
    arrs["fine_llr"] = np.stack([tr.fine[i]["llr"] for i in fi]).astype(np.float32) if fi else np.zeros((0, 174), np.float32)
    nsg = min(len(fi), 24)
    arrs["fine_sgrid"] = np.stack([tr.fine[i]["signal_grid"] for i in fi[:nsg]]).astype(np.float32) \
        if nsg else np.zeros((0, 79, 8), np.float32)
    # BP call sample: all successes, all with NaN, all early-outs among the first 40, then strided
    bp = tr.bp_calls
    sel = set()
    for k, c in enumerate(bp):
        if c["result"] is not None or (len(c["llr_out"]) == 174 and np.isnan(c["llr_out"]).any()):
            sel.add(k)
    sel.update(range(0, len(bp), max(1, len(bp) // 60)))
    sel = sorted(sel)[:160]
    arrs["bp_llr_in"] = np.stack([bp[k]["llr_in"] for k in sel]).astype(np.float32) if sel else np.zeros((0, 174), np.float32)
    arrs["bp_nc0max"] = np.array([bp[k]["nc0max"] for k in sel], dtype=np.int32)
    arrs["bp_iters"] = np.array([bp[k]["iters"] for k in sel], dtype=np.int32)
    arrs["bp_nits"] = np.array([bp[k]["n_its"] for k in sel], dtype=np.int32)
    arrs["bp_has_out"] = np.array([len(bp[k]["llr_out"]) == 174 for k in sel], dtype=np.bool_)
    arrs["bp_llr_out"] = np.stack([bp[k]["llr_out"] if len(bp[k]["llr_out"]) == 174 else np.zeros(174, np.float32)
                                   for k in sel]).astype(np.float32) if sel else np.zeros((0, 174), np.float32)
    bp_res = [msg_str(bp[k]["result"]) for k in sel]
    osd = tr.osd_calls
    osel = set(k for k, c in enumerate(osd) if c["result"] is not None)
    osel.update(range(0, len(osd), max(1, len(osd) // 50)))
    osel = sorted(osel)[:120]
    arrs["osd_llr_in"] = np.stack([osd[k]["llr_in"] for k in osel]).astype(np.float32) if osel else np.zeros((0, 174), np.float32)
    osd_res = [msg_str(osd[k]["result"]) for k in osel]
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrs)
    msgs = []
    for m in tr.messages:
        d = {k: v for k, v in m.items() if k != "decode_completed"}
        d["msg_tuple"] = list(d["msg_tuple"])
        d["tsec"], d["fHz"] = float(d["tsec"]), float(d["fHz"])
        msgs.append(d)
    js = dict(
        meta=meta, numpy=np.__version__, n_cands=n,
        n_bp_calls=len(bp), n_osd_calls=len(osd),
        bp_result=bp_res, osd_result=osd_res,
        bp_meta=[dict(cand=bp[k]["cand"], ipass=bp[k]["ipass"], ap=bp[k]["ap"]) for k in sel],
        osd_meta=[dict(cand=osd[k]["cand"], ipass=osd[k]["ipass"], ap=osd[k]["ap"]) for k in osel],
        unpack=[dict(bits77=f"{b:020x}", result=msg_str(r), cand=c, ipass=ip) for b, r, c, ip in tr.unpack_calls],
        final=[dict(f, result=msg_str(f["result"])) for f in tr.final],
        messages=msgs,
    )
    with open(os.path.join(OUT, name + ".json"), "w") as f:
        json.dump(js, f, indent=0)
    print(f"{name}: {n} cands, {len(msgs)} msgs, {len(bp)} bp calls ({len(sel)} kept), "
          f"{len(osd)} osd calls ({len(osel)} kept), {len(tr.unpack_calls)} unpack calls")
    return tr


def main():
    os.makedirs(OUT, exist_ok=True)
    for w in ("test_08", "test_09"):
        src = os.path.join(REF_ROOT, "tests", "pipeline", w + ".wav")
        shutil.copyfile(src, os.path.join(OUT, w + ".wav"))
        dump(w, read_wav_i16(src), dict(kind="wav", file=w + ".wav"))
    dump("synth_000000", synth.make_frame(0), dict(kind="synth", index=0, n_signals=50, snr=[-10.0, 10.0]))
    dump("synth_100000", synth.make_frame(100000, snr_range=(-21.0, -11.0)),
         dict(kind="synth", index=100000, n_signals=50, snr=[-21.0, -11.0]))
    dump("synth_200000", synth.make_frame(200000, n_signals=6, snr_range=(-18.0, -8.0)),
         dict(kind="synth", index=200000, n_signals=6, snr=[-18.0, -8.0]))


if __name__ == "__main__":
    main()
