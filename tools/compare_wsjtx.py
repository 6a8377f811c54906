"""Decodes of the reference's two fixture recordings against the WSJT-X listings the reference keeps for the same cycles
(tests/golden/wsjtx_cycles_1_2.json, from the reference's tests/*.txt).  Usage (GPU box): python tools/compare_wsjtx.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden  # noqa: E402
from pyft8_amd.receiver import Receiver  # noqa: E402


def norm(t):          # hashed calls print differently per decoder: compare on the non-hash words
    return " ".join(w for w in t.split() if not w.startswith("<"))


def main():
    L = json.load(open(os.path.join(ROOT, "tests", "golden", "wsjtx_cycles_1_2.json")))["listings"]
    rx = Receiver("", None, max_frames=2)
    audio = np.stack([load_golden("test_08")[0], load_golden("test_09")[0]])
    print("recording  WSJT-X FAST  WSJT-X NORM  PyFT8 live |  passes  decodes  in WSJT-X NORM  not in any listing")
    for passes in (1, 2, 3):
        for osd in ((True,) if passes == 1 else (True, False)):
            out = rx.decode_frames(audio, passes=passes, sub_pass_osd=osd)
            for f, name in enumerate(("test_08", "test_09")):
                got = [norm(" ".join(d["msg_tuple"])) for d in out[f]]
                wn = {norm(t) for t in L["NORM"][name]}
                anyl = wn | {norm(t) for t in L["FAST"][name]} | {norm(t) for t in L["PyFT8_live"][name]}
                print(f"{name:9s} {len(L['FAST'][name]):11d} {len(L['NORM'][name]):12d} {len(L['PyFT8_live'][name]):11d} | {passes:6d}"
                      f"{'' if osd else '*'} {len(got):8d} {sum(g in wn for g in got):15d} {sum(g not in anyl for g in got):18d}")
    print("* = no OSD decodes accepted in the passes after the first")


if __name__ == "__main__":
    main()
