"""The build's multi-pass composition (an EXTENSION: batched sweeps, product constants) next to the reference author's own subtraction
experiment (tests/pipeline/receiver_sub.py run by oracle/gen_golden_sandbox.py -> tests/golden/sandbox_multipass.json) and next to the
plain reference receiver (tests/golden/test_0x.json), on the two fixture recordings.  A REPORT, not a parity test: the experiment has its
own scheduler and constants (BP 25 iterations, search over every second f0 bin, search_timerange [-2.5, 3.5], one subtraction per
decode), so equality is not expected -- VERDICT r5 item 7 asks for one reference-held number beside the extension's.

    python tools/sandbox_overlap.py            (GPU box: the product, Receiver.decode_frames)
    python tools/sandbox_overlap.py --oracle   (CPU: the oracle's composition, oracle.decode_frame_passes)
"""
import json
import os
import sys
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLD = os.path.join(ROOT, "tests", "golden")


def read_wav(path):
    with wave.open(path, "rb") as w:
        data = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16)
    out = np.zeros(180000, np.int16)
    out[:min(len(data), 180000)] = data[:180000]
    return out


def main():
    use_oracle = "--oracle" in sys.argv
    sand = json.load(open(os.path.join(GOLD, "sandbox_multipass.json")))
    wsjtx = json.load(open(os.path.join(GOLD, "wsjtx_cycles_1_2.json"))) if os.path.exists(os.path.join(GOLD, "wsjtx_cycles_1_2.json")) else None
    if use_oracle:
        import oracle as O
        from pyft8_amd import _lib

        def decode(audio, **kw):
            if kw.get("passes", 1) == 1:
                return [(" ".join(m["msg_tuple"]), O.notes_of(m)) for m in O.decode_frame(audio, O.default_config(**_lib.fft_plans()))["msgs"]]
            r = O.decode_frame_passes(audio, O.default_config(**_lib.fft_plans()), passes=kw["passes"], research=kw.get("research", "full"))
            return [(" ".join(m["msg_tuple"]), O.notes_of(m) + ("_SUB" if p else "")) for p, m in r["msgs"]]
        who = "CPU oracle composition (oracle.decode_frame_passes)"
    else:
        from pyft8_amd.receiver import Receiver
        rx = Receiver("", None, max_frames=1)

        def decode(audio, **kw):
            return [(" ".join(d["msg_tuple"]), d["decode_notes"]) for d in rx.decode_frames(audio[None], **kw)[0]]
        who = "product on the GPU (Receiver.decode_frames)"
    print(f"multi-pass composition vs the reference's own subtraction experiment -- {who}")
    print("sets of message texts per recording; 'experiment' = tests/pipeline/receiver_sub.py's manage_cycle under the virtual clock of")
    print("oracle/gen_golden_sandbox.py (numpy " + sand["numpy"] + "); 'plain reference' = PyFT8/receiver.py (tests/golden/test_0x.json)\n")
    tot = {}
    for name in ("test_08", "test_09"):
        audio = read_wav(os.path.join(GOLD, name + ".wav"))
        g = json.load(open(os.path.join(GOLD, name + ".json")))
        plain_ref = {" ".join(m["msg_tuple"]) for m in g["messages"]} if "messages" in g else None
        exp = {m["text"] for m in sand["frames"][name]["experiment"]["messages"]}
        exp_sub = {m["text"] for m in sand["frames"][name]["experiment"]["messages"] if "_SUB" in m["notes"]}
        exp_nosub = {m["text"] for m in sand["frames"][name]["experiment_without_subtraction"]["messages"]}
        p1 = {t for t, _ in decode(audio, passes=1)}
        p2l = decode(audio, passes=2, research="local")
        p2f = decode(audio, passes=2, research="full")
        s2l, s2f = {t for t, _ in p2l}, {t for t, _ in p2f}
        info = sand["frames"][name]["experiment"]["info"]
        print(f"{name}.wav")
        print(f"  experiment (reference sandbox):            {len(exp):3d} messages, {len(exp_sub)} tagged _SUB; {info['subtracted']} subtractions, "
              f"{info['subtract_raised']} raised, {info['local_candidates']} local candidates; without its subtraction: {len(exp_nosub)}")
        if plain_ref is not None:
            print(f"  plain reference receiver:                  {len(plain_ref):3d} messages; experiment & plain reference: {len(exp & plain_ref)}")
        print(f"  this build, 1 pass:                        {len(p1):3d} messages; & experiment: {len(p1 & exp)}")
        print(f"  this build, passes=2, research='local':    {len(s2l):3d} messages ({len(s2l - p1)} from the second pass); & experiment: {len(s2l & exp)}; "
              f"experiment only: {sorted(exp - s2l)}; build only: {sorted(s2l - exp)}")
        print(f"  this build, passes=2, research='full':     {len(s2f):3d} messages ({len(s2f - p1)} from the second pass); & experiment: {len(s2f & exp)}")
        gained_exp, gained_build = exp - exp_nosub, s2l - p1
        print(f"  what subtraction ADDS: experiment {sorted(gained_exp)}; build (local) {sorted(gained_build)}; common {sorted(gained_exp & gained_build)}")
        if wsjtx:
            for mode, lst in wsjtx["listings"].items():
                ws = set(lst.get(name, []))
                if ws:
                    print(f"  (WSJT-X {mode} listing of this cycle: {len(ws)} messages; & experiment {len(exp & ws)}, & build 1 pass {len(p1 & ws)}, "
                          f"& build passes=2 local {len(s2l & ws)}, & build passes=2 full {len(s2f & ws)})")
        for k, v in (("experiment", len(exp)), ("build_1pass", len(p1)), ("build_2pass_local", len(s2l)), ("build_2pass_full", len(s2f)),
                     ("overlap_local", len(s2l & exp))):
            tot[k] = tot.get(k, 0) + v
        print()
    print("both recordings:", tot)
    print("The '+32 %' yield quoted for the multi-pass EXTENSION (profiles/r05_multi_pass_yield.txt: synthetic 50-signal frames with known truth)")
    print("is a figure of this build's own composition against its own oracle; the reference-held figure is the experiment's above.")


if __name__ == "__main__":
    main()
