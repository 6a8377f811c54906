#!/bin/bash
# three / four chunk streams with hardware queues of their own (FT8RX_SUBS_FIRST, a measurement aid of ft8rx_create) against the default two
set -u
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
B="python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-host-entry --no-other-configs"
export FT8RX_LIB=$PWD/build/ab/subsfirst.so
for i in 1 2; do
  timeout 300 $B --streams 2 2>/dev/null | tail -1 > "$OUT/r06_str_s2_$i.json"
  FT8RX_SUBS_FIRST=2 timeout 300 $B --streams 3 2>/dev/null | tail -1 > "$OUT/r06_str_s3q_$i.json"
  timeout 300 $B --streams 3 2>/dev/null | tail -1 > "$OUT/r06_str_s3_$i.json"
  FT8RX_SUBS_FIRST=3 timeout 300 $B --streams 4 2>/dev/null | tail -1 > "$OUT/r06_str_s4q_$i.json"
  FT8RX_SUBS_FIRST=1 timeout 300 $B --streams 2 --frames 512 2>/dev/null | tail -1 > "$OUT/r06_str_s2b512_$i.json"
  FT8RX_SUBS_FIRST=2 timeout 300 $B --streams 3 --frames 384 2>/dev/null | tail -1 > "$OUT/r06_str_s3qb384_$i.json"
done
python3 - <<'PY'
import json, glob
for kind in ("s2", "s3q", "s3", "s4q", "s2b512", "s3qb384"):
    vals = []
    for f in sorted(glob.glob(f"gpurun_out/r06_str_{kind}_*.json")):
        try:
            d = json.loads(open(f).read()); vals.append((round(d["value"]), round(d["config"]["kernel_only_frames_per_s_this_rank"]), round(d["step_gap_ms"]["p50"], 3)))
        except Exception as e:
            vals.append(("FAILED", str(e)[:60]))
    print(kind, vals)
PY
