#!/bin/bash
# A/B of alternative builds of libft8rx.so with the FULL default bench (timed loop, host entries):  tools/ab_full.sh <tag> lib1.so lib2.so ...
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
for L in "$@"; do
  N=$(basename "$L" .so)
  FT8RX_LIB=$PWD/$L timeout 600 python3 bench.py --no-cpu-baseline --no-other-configs > "$OUT/${TAG}_${N}.json" 2> "$OUT/${TAG}_${N}.err" || echo "$N failed: $(tail -3 $OUT/${TAG}_${N}.err)"
done
python3 - "$TAG" "$@" <<'PY'
import json, sys, os
tag = sys.argv[1]
print("%-10s %9s %8s %9s %9s %9s %8s %8s %8s %8s" % ("lib", "value", "ms/step", "incl_h2d", "sync_call", "kern_only", "fine", "bp_fine", "osd", "spectro"))
for L in sys.argv[2:]:
    n = os.path.basename(L)[:-3]
    try:
        d = json.loads([l for l in open(f"gpurun_out/{tag}_{n}.json") if l.startswith("{")][-1])
        s = d["stage_ms"]
        print("%-10s %9.0f %8.3f %9.0f %9.0f %9.0f %8.3f %8.3f %8.3f %8.3f" % (n, d["value"], d["ms_per_step"], d["value_incl_h2d"] or 0, d["value_incl_h2d_sync_call"] or 0,
              d["config"]["kernel_only_frames_per_s_this_rank"], s["fine"], s["bp_fine"], s["osd"], s["spectrogram"]))
    except Exception as e:
        print(n, "error", e)
PY
