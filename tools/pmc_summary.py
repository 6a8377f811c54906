"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (csv) per kernel.
Usage: python tools/pmc_summary.py fetch_counter_collection.csv write_counter_collection.csv out.json [out.txt]
FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 reports half of a coalesced stream; calibrated here on
k_cyc_c whose read is exactly 2 x 49152 x 8 B per frame); WRITE_SIZE is used as is (calibrated on the
spectrogram grid write)."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from src_hash import source_hash  # noqa: E402


def agg(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return d


def main():
    f, w = agg(sys.argv[1]), agg(sys.argv[2])
    out, lines = {}, [f"{'kernel':<26}{'launches':>9}{'fetch_MB(x2)':>14}{'write_MB':>11}{'hbm_MB/launch':>15}"]
    for k in sorted(f, key=lambda k: -sum(f[k])):
        fa = 2.0 * 1024 * sum(f[k]) / len(f[k])
        wa = 1024 * sum(w.get(k, [0.0])) / max(1, len(w.get(k, [0.0])))
        out[k] = {"launches": len(f[k]), "fetch_bytes": fa, "write_bytes": wa, "hbm_bytes": fa + wa}
        lines.append(f"{k:<26}{len(f[k]):>9}{fa/1e6:>14.1f}{wa/1e6:>11.1f}{(fa+wa)/1e6:>15.1f}")
    out["_source_hash"] = source_hash()          # bench.py: traffic_stale when the benchmarked tree differs
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    txt = "\n".join(lines) + "\n"
    if len(sys.argv) > 4:
        open(sys.argv[4], "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
