#!/bin/bash
# rank 0's gather load on one GPU -- the RCCL path in a one-rank nccl group with rank 0's eight-rank load
# (--gather-repeat 8) against --no-gather, interleaved, config 1.  -> gpurun_out/r06_gather_*.json + a table
set -u
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
G="--gpus 1 --backend nccl --force-gather --no-cpu-baseline --no-host-entry --no-other-configs --steps 100"
for i in 1 2 3; do
  timeout 600 python3 bench.py $G --no-gather > "$OUT/r06_gather_none_$i.json" 2>> "$OUT/r06_gather.err"
  timeout 600 python3 bench.py $G --gather-repeat 8 > "$OUT/r06_gather_x8_$i.json" 2>> "$OUT/r06_gather.err"
  timeout 600 python3 bench.py $G > "$OUT/r06_gather_x1_$i.json" 2>> "$OUT/r06_gather.err"
done
python3 - <<'PY'
import json, glob
for kind in ("none", "x1", "x8"):
    vals = []
    for f in sorted(glob.glob(f"gpurun_out/r06_gather_{kind}_*.json")):
        try:
            d = json.loads([l for l in open(f) if l.startswith("{")][-1])
            vals.append((round(d["value"]), round(d["step_gap_ms"]["p50"], 3), (d["per_rank"]["gather"] or {}).get("submit_ms_per_step")))
        except Exception as e:
            vals.append(("FAILED", str(e)))
    print(kind, vals)
PY
