"""Summarise rocprofv3 --pmc SQ_* passes per kernel: average counter value per launch and the derived fractions.
Usage: python tools/pmc_sq_summary.py out.txt pass1.csv pass2.csv ...
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md); fractions are of
SQ_WAVE_CYCLES, i.e. per resident wave."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from src_hash import source_hash  # noqa: E402


def main():
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in sys.argv[2:]:
        per_dispatch = collections.defaultdict(float)
        names = {}
        for r in csv.DictReader(open(path)):
            key = (r["Dispatch_Id"], r["Counter_Name"])
            per_dispatch[key] += float(r["Counter_Value"])           # one row per XCD / SE instance: sum them
            names[r["Dispatch_Id"]] = r["Kernel_Name"].split("(")[0]
        for (d, c), v in per_dispatch.items():
            vals[names[d]][c].append(v)
    lines = []
    ctrs = sorted({c for k in vals for c in vals[k]})
    avg = {k: {c: sum(v) / len(v) for c, v in vals[k].items()} for k in vals}
    order = sorted(avg, key=lambda k: -avg[k].get("SQ_WAVE_CYCLES", 0))
    lines.append("per-launch averages (B = 256 frames, bench.py --streams 1)")
    hdr = f"{'kernel':<16}" + "".join(f"{c.replace('SQ_', ''):>20}" for c in ctrs)
    lines.append(hdr)
    for k in order:
        lines.append(f"{k[:16]:<16}" + "".join(f"{avg[k].get(c, float('nan')):>20.4g}" for c in ctrs))
    lines.append("")
    lines.append(f"{'kernel':<16}{'VALU act':>10}{'SCA act':>10}{'LDS act':>10}{'WAIT_INST':>10}{'WAIT_ANY':>10}{'bankconf/LDS':>13}{'VALU inst/wave':>15}{'SALU inst/wave':>15}{'waves':>10}")
    for k in order:
        a = avg[k]
        wc = a.get("SQ_WAVE_CYCLES", 0) or float("nan")
        w = a.get("SQ_WAVES", 0) or float("nan")
        lines.append(f"{k[:16]:<16}{a.get('SQ_ACTIVE_INST_VALU', 0) / wc:>10.3f}{a.get('SQ_ACTIVE_INST_SCA', 0) / wc:>10.3f}"
                     f"{a.get('SQ_ACTIVE_INST_LDS', 0) / wc:>10.3f}{a.get('SQ_WAIT_INST_ANY', 0) / wc:>10.3f}{a.get('SQ_WAIT_ANY', 0) / wc:>10.3f}"
                     f"{a.get('SQ_LDS_BANK_CONFLICT', 0) / (a.get('SQ_LDS_IDX_ACTIVE', 0) or float('nan')):>13.3f}"
                     f"{a.get('SQ_INSTS_VALU', 0) / w:>15.1f}{a.get('SQ_INSTS_SALU', 0) / w:>15.1f}{w:>10.0f}")
    txt = "\n".join(lines) + "\n"
    open(sys.argv[1], "w").write(txt)
    # the same per-launch averages as JSON (bench.py reads INSTS_VALU from the committed copy, profiles/sq_latest.json)
    js = {k: {c.replace("SQ_", ""): v for c, v in avg[k].items()} for k in avg}
    js["_source_hash"] = source_hash()
    json.dump(js, open(os.path.splitext(sys.argv[1])[0] + ".json", "w"), indent=1)
    print(txt)


if __name__ == "__main__":
    main()
