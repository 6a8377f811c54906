#!/bin/bash
# A/B timing of alternative builds of libft8rx.so (same ABI) on the GPU box:  tools/ab_variants.sh <tag> lib1.so lib2.so ...
# For each library: bench.py (one stream, per-stage HIP-event times) -> gpurun_out/<tag>_<name>.json ; prints a stage table.
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
for L in "$@"; do
  N=$(basename "$L" .so)
  FT8RX_LIB=$PWD/$L timeout 600 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-entry --no-other-configs > "$OUT/${TAG}_${N}.json" 2> "$OUT/${TAG}_${N}.err" || echo "$N failed: $(tail -3 $OUT/${TAG}_${N}.err)"
done
python3 - "$TAG" "$@" <<'PY'
import json, sys, os
tag = sys.argv[1]
rows = {}
for L in sys.argv[2:]:
    n = os.path.basename(L)[:-3]
    try:
        d = json.loads(open(f"gpurun_out/{tag}_{n}.json").read().strip().splitlines()[-1])
        rows[n] = (d["value"], d["stage_ms"])
    except Exception as e:
        rows[n] = (0.0, {"error": str(e)})
stages = []
for v, s in rows.values():
    for k in s:
        if k not in stages: stages.append(k)
print("%-14s" % "stage" + "".join("%14s" % n[:13] for n in rows))
print("%-14s" % "frames/s" + "".join("%14.0f" % v for v, _ in rows.values()))
for st in stages:
    print("%-14s" % st + "".join("%14s" % (("%.4f" % s[st]) if st in s and not isinstance(s[st], str) else "-") for _, s in rows.values()))
PY
