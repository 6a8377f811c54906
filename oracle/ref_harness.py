"""Drive the *real* PyFT8 reference (read-only at /root/reference) under a virtual clock.

TEST INFRASTRUCTURE ONLY.  This module exists to generate golden vectors in the
build container; the reference cannot travel to the GPU box, so nothing in
tests/, bench.py or the product imports this at run time -- only
oracle/gen_golden.py does (and the optional `-m ref` cross-checks).

Protocol = SURVEY.md section 8(c) "frame-complete" semantics:
  * stub `pyaudio` and `paho.mqtt.client` (absent here; receiver.py:3, pskreporter.py:1)
  * time_utils.time -> virtual clock, time_utils.sleep -> no-op (time_utils.py:7-14)
  * threads are never started while the Receiver is constructed (receiver.py:252,336)
  * 375 hops of 480 samples are pushed through AudioIn._callback (receiver.py:295-306)
  * Receiver.search(...) then the ipass ladder exactly as manage_cycle does
    (receiver.py:389-398): per round, undecoded candidates sorted by llr_sd
    (stable, descending) advance one ipass.
Every ldpc_decode / osd_012 / unpack call is recorded through wrappers so that
per-stage goldens can be dumped.
"""
import os
import sys
import types
import threading
import numpy as np

REF_ROOT = os.environ.get("PYFT8_REFERENCE", "/root/reference")


def _install_stubs():
    if "pyaudio" not in sys.modules:
        pa = types.ModuleType("pyaudio")
        pa.paInt16 = 8
        pa.paContinue = 0

        class _PyAudio:
            def get_device_count(self):
                return 0

            def get_device_info_by_index(self, i):
                return {"name": ""}

            def open(self, *a, **k):
                class _S:
                    def start_stream(self):
                        pass

                    def write(self, b):
                        pass

                    def stop_stream(self):
                        pass

                    def close(self):
                        pass
                return _S()
        pa.PyAudio = _PyAudio
        sys.modules["pyaudio"] = pa
    if "paho" not in sys.modules:
        paho = types.ModuleType("paho")
        mqtt = types.ModuleType("paho.mqtt")
        client = types.ModuleType("paho.mqtt.client")

        class _Client:
            def __init__(self, *a, **k):
                pass

            def __getattr__(self, name):
                return lambda *a, **k: None
        client.Client = _Client
        paho.mqtt = mqtt
        mqtt.client = client
        sys.modules["paho"] = paho
        sys.modules["paho.mqtt"] = mqtt
        sys.modules["paho.mqtt.client"] = client


_ref = None


def load_reference():
    """Import PyFT8.receiver/decoders from the read-only reference tree."""
    global _ref
    if _ref is not None:
        return _ref
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError(f"reference tree not present at {REF_ROOT}")
    _install_stubs()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import PyFT8.time_utils as tu
    import PyFT8.databases as db
    import PyFT8.decoders as dec
    import PyFT8.receiver as rcv
    _ref = types.SimpleNamespace(tu=tu, db=db, dec=dec, rcv=rcv)
    return _ref


class Trace:
    """Everything observed while decoding one frame with the reference."""

    def __init__(self):
        self.bp_calls = []      # dict(cand, ipass, ap, llr_in, nc0max, iters, result, n_its, llr_out)
        self.osd_calls = []     # dict(cand, ipass, ap, llr_in, result)
        self.unpack_calls = []  # (bits77:int, result)
        self.fine = {}          # cand -> dict
        self.grid = {}          # cand -> dict(llr, sd, snr)
        self.messages = []      # message dicts in emit order
        self.final = []         # per candidate summary


def run_frame(audio_i16, workdir="/tmp/pyft8_ref_scratch", collect=True, **rx_kwargs):
    """Decode one 15-s frame (180000 int16 @ 12 kHz) with the reference.

    Returns (candidates, trace, receiver)."""
    ref = load_reference()
    tu, db, dec, rcv = ref.tu, ref.db, ref.dec, ref.rcv
    audio_i16 = np.ascontiguousarray(audio_i16, dtype=np.int16)
    assert audio_i16.shape == (180000,)
    os.makedirs(workdir, exist_ok=True)
    cwd = os.getcwd()
    os.chdir(workdir)  # decoders.py:114 appends rejected_callsigns.txt to cwd
    trace = Trace()
    vt = [0.0]
    orig_time, orig_sleep = tu.time_utils.time, tu.time_utils.sleep
    orig_start = threading.Thread.start
    orig_ldpc, orig_osd = rcv.ldpc_decode, rcv.osd_012
    orig_unpack = dec.unpack
    cur = {"cand": -1, "ipass": -1}
    try:
        tu.time_utils.time = lambda: vt[0]
        tu.time_utils.sleep = lambda t: None
        threading.Thread.start = lambda self: None
        db.call_hashes.clear()
        db.hashes_for_calls.clear()

        def unpack_w(bits):
            r = orig_unpack(bits)
            trace.unpack_calls.append((int(bits), r, cur["cand"], cur["ipass"]))
            return r
        dec.unpack = unpack_w  # crc_unpack91 looks the name up in module globals

        def ldpc_w(llr, nc0, its):
            llr_in = llr.copy()
            res = orig_ldpc(llr, nc0, its)
            if collect:
                trace.bp_calls.append(dict(cand=cur["cand"], ipass=cur["ipass"], ap=cur.get("ap"),
                                           llr_in=llr_in, nc0max=nc0, iters=its, result=res[0],
                                           n_its=res[1], llr_out=(np.array(res[2], dtype=np.float32))))
            return res

        def osd_w(llr, *a, **k):
            res = orig_osd(llr, *a, **k)
            if collect:
                trace.osd_calls.append(dict(cand=cur["cand"], ipass=cur["ipass"], ap=cur.get("ap"),
                                            llr_in=np.array(llr, dtype=np.float32).copy(), result=res))
            return res
        rcv.ldpc_decode, rcv.osd_012 = ldpc_w, osd_w

        rx = rcv.Receiver("x", trace.messages.append, **rx_kwargs)
        rx.audio_in.search_grid_ptr = 0
        for k in range(375):
            vt[0] = (k + 1) * 0.04
            rx.audio_in._callback(audio_i16[480 * k:480 * k + 480].tobytes(), 480, None, None)
        vt[0] = 15.0
        cs = tu.time_utils.cyclestart_string(vt[0])
        f_rng = rx.audio_in.search_f0_idx_range
        cands = rx.search(cs, 0, range(f_rng[0], f_rng[1]))
        for i, c in enumerate(cands):
            c._idx = i
            # record the AP pattern name at each decoder call
            orig_set_ap = c._set_AP

            def set_ap(p, _o=orig_set_ap):
                cur["ap"] = p[0]
                return _o(p)
            c._set_AP = set_ap
        dup = set()
        for rnd in range(8):
            todo = [c for c in cands if not c.decode_result]
            todo.sort(key=lambda c: c.llr_sd, reverse=True)
            for c in todo:
                cur["cand"], cur["ipass"] = c._idx, c.ipass
                ip = c.ipass
                if ip == 6:
                    # saved llrs: AP name comes from the tuple (receiver.py:101-103)
                    cur["ap"] = "saved"
                c.decode(10 + rnd)
                if ip == 0 and collect:
                    trace.grid[c._idx] = dict(llr=np.array(c.llr0, dtype=np.float32).copy()
                                              if hasattr(c, "llr0") else None,
                                              sd=float(c.llr_sd), snr=int(c.snr))
                if ip == 1 and collect:
                    trace.fine[c._idx] = dict(
                        tweaks=c.tweaks, n_sync=int(c.n_sync_matches), sd=float(c.llr_sd),
                        snr=int(c.snr), stopped=(c.decode_result == 'stop'),
                        llr=np.array(c.llr, dtype=np.float32).copy(),
                        signal_grid=np.array(c.signal_grid, dtype=np.float32).copy())
                if c.decode_result is not None and c.decode_result != 'stop':
                    c._decoded_at = ip
                    c._result = c.decode_result
                    c._notes = c.decode_notes
                    c.check_and_package(dup)
        for c in cands:
            trace.final.append(dict(idx=c._idx, f0_idx=int(c.origin['f0_idx']), h0_idx=int(c.origin['h0_idx']),
                                    score=float(c.origin['score']),
                                    result=getattr(c, "_result", None), ipass=getattr(c, "_decoded_at", -1),
                                    notes=getattr(c, "_notes", ""), tweaks=c.tweaks,
                                    tsec=float(c.origin['tsec']), fHz=float(c.origin['fHz']),
                                    snr=int(getattr(c, "snr", 0))))
        return cands, trace, rx
    finally:
        tu.time_utils.time, tu.time_utils.sleep = orig_time, orig_sleep
        threading.Thread.start = orig_start
        rcv.ldpc_decode, rcv.osd_012 = orig_ldpc, orig_osd
        dec.unpack = orig_unpack
        os.chdir(cwd)


def read_wav_i16(path):
    import wave
    with wave.open(path, "rb") as w:
        assert w.getnchannels() == 1 and w.getsampwidth() == 2 and w.getframerate() == 12000
        data = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16)
    out = np.zeros(180000, dtype=np.int16)
    n = min(len(data), 180000)
    out[:n] = data[:n]
    return out


if __name__ == "__main__":
    import time
    for name in ("test_08.wav", "test_09.wav"):
        a = read_wav_i16(os.path.join(REF_ROOT, "tests/pipeline", name))
        t0 = time.time()
        cands, tr, rx = run_frame(a)
        msgs = [" ".join(m["msg_tuple"]) for m in tr.messages]
        print(name, len(cands), "cands", len(msgs), "msgs", f"{time.time()-t0:.1f}s")
        for m in tr.messages:
            print("   ", m["all_txt_format"], "|", m["decode_notes"])
