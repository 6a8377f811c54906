"""The reference author's subtraction EXPERIMENT, run as it stands, on the two fixture recordings.

    python oracle/gen_golden_sandbox.py        -> tests/golden/sandbox_multipass.json        (build container only: needs /root/reference)

tests/pipeline/receiver_sub.py is a sandbox copy of the receiver with its own scheduler (Receiver.manage_cycle, :407-445: after every
decode with SNR > -10 the signal is refined, re-synthesised and subtracted from the audio buffer, the hops it touched are recomputed,
and the columns f0 - 2 .. f0 + 1 are searched again with the sync threshold ignored) and its own constants (BP 25 iterations :119, the
ladder one step shorter :94-127, search over every second f0 bin :456, search_timerange [-2.5, 3.5] :325).  This script imports that
file from the read-only reference tree and drives ITS manage_cycle under a virtual clock:
  * pyaudio / paho stubs, no thread is started, time_utils.time = virtual clock (oracle/ref_harness.py);
  * 375 hops of 480 samples go through ITS AudioIn._callback at virtual times (k + 0.5) x 40 ms, so that hop k lands in grid row k;
  * the clock then stays at 14.98 s (cycle 0, grid row 374 > its search_start_hop 272) and manage_cycle() is entered: it searches once,
    then advances the candidates step by step; time_utils.sleep(0.01) -- the top of its loop -- ends the run when nothing it would still
    decode is left.  Its rule "a candidate is decoded once the grid pointer has left its rows" (:424) is kept: candidates whose rows
    reach row 374 (start later than ~3.1 s into the cycle) never run, as in the live program at that instant;
  * its subtract_signal raises ValueError for a signal that ends beyond the 15 s of audio (:389-391 guard against 192000 samples, the
    buffer has 180000); such a subtraction is skipped and counted (the live program's scheduler thread would end there).
Stored: the messages in emit order (text, decode_notes, SNR, dt, frequency) and which of them carry the experiment's '_SUB' tag.  DATA
only; no reference source text.  This build's batched multi-pass composition (Receiver.decode_frames(passes=2, research="local")) is an
extension with constants of the PRODUCT receiver, so equality is not expected: tools/sandbox_overlap.py REPORTS the overlap
(profiles/r06_multipass_vs_reference_sandbox.txt)."""
import importlib.util
import json
import os
import sys
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_harness  # noqa: E402


class _Done(Exception):
    pass


def run_sandbox(audio_i16, subtract=True, workdir="/tmp/pyft8_ref_scratch"):
    ref = ref_harness.load_reference()
    tu, db = ref.tu, ref.db
    spec = importlib.util.spec_from_file_location("receiver_sub", os.path.join(ref_harness.REF_ROOT, "tests", "pipeline", "receiver_sub.py"))
    rs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rs)
    os.makedirs(workdir, exist_ok=True)
    cwd = os.getcwd()
    os.chdir(workdir)                     # subtraction.txt / rejected_callsigns.txt are appended in the cwd
    vt = [0.0]
    msgs = []
    state = {"rx": None, "searched": False, "loops": 0}
    orig_time, orig_sleep, orig_start = tu.time_utils.time, tu.time_utils.sleep, threading.Thread.start

    def sleep(t):
        rx = state["rx"]
        if t != 0.01 or rx is None:        # Candidate.decode's sleep(0), the constructor's sleep(0.5)
            return
        state["loops"] += 1
        if state["searched"]:
            ptr = rx.audio_in.search_grid_ptr
            left = [c for c in rx.candidates if (not c.decode_result) and not (c.search_grid_bounds[0] <= ptr <= c.search_grid_bounds[1])]
            if not left:
                raise _Done
        if state["loops"] > 10000:
            raise RuntimeError("the sandbox's manage_cycle does not come to an end")
    try:
        tu.time_utils.time = lambda: vt[0]
        tu.time_utils.sleep = sleep
        threading.Thread.start = lambda self: None
        db.call_hashes.clear()
        db.hashes_for_calls.clear()
        rx = rs.Receiver("x", msgs.append)
        orig_subtract = rx.subtract_signal

        def subtract_signal(c):
            # receiver_sub.py:389-391 guards against 192000 samples but slices a 180000-sample buffer: for a signal that ends
            # beyond the 15 s of audio (tsec > 2.36 s) the experiment raises ValueError -- in the live program that ends its
            # manage_cycle thread.  Here such a subtraction is skipped (counted), so that the run reports what the scheduler finds.
            if not subtract:
                return
            try:
                orig_subtract(c)
                state["subtracted"] = state.get("subtracted", 0) + 1
            except ValueError:
                state["subtract_raised"] = state.get("subtract_raised", 0) + 1
        rx.subtract_signal = subtract_signal
        orig_search = rx.search

        def search(*a, **k):
            state["searched"] = True
            return orig_search(*a, **k)
        rx.search = search
        for k in range(375):
            vt[0] = (k + 0.5) * 0.04
            rx.audio_in._callback(np.ascontiguousarray(audio_i16[480 * k:480 * k + 480]).tobytes(), 480, None, None)
        state["rx"] = rx
        try:
            rx.manage_cycle()
        except _Done:
            pass
        n_cands = len(rx.candidates)
        n_local = sum(1 for c in rx.candidates if getattr(c, "subtracted", False))
        n_skipped = sum(1 for c in rx.candidates if not c.decode_result)
        return msgs, dict(candidates=n_cands, local_candidates=n_local, never_run=n_skipped, loops=state["loops"],
                          subtracted=state.get("subtracted", 0), subtract_raised=state.get("subtract_raised", 0))
    finally:
        tu.time_utils.time, tu.time_utils.sleep = orig_time, orig_sleep
        threading.Thread.start = orig_start
        os.chdir(cwd)


def main():
    out = {"numpy": np.__version__, "protocol": __doc__.split("This script")[1].split("Stored:")[0].strip(), "frames": {}}
    for name in ("test_08", "test_09"):
        audio = ref_harness.read_wav_i16(os.path.join(ROOT, "tests", "golden", name + ".wav"))
        e = {}
        for label, sub in (("experiment", True), ("experiment_without_subtraction", False)):
            msgs, info = run_sandbox(audio, subtract=sub)
            rows = [dict(text=" ".join(m["msg_tuple"]), notes=m["decode_notes"], snr=m["their_snr"], tsec=float(m["tsec"]), fHz=float(m["fHz"]))
                    for m in msgs]
            e[label] = dict(messages=rows, info=info)
            print(f"{name} {label}: {len(rows)} messages ({sum('_SUB' in r['notes'] for r in rows)} tagged _SUB), {info}")
        out["frames"][name] = e
    with open(os.path.join(ROOT, "tests", "golden", "sandbox_multipass.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote tests/golden/sandbox_multipass.json")


if __name__ == "__main__":
    main()
