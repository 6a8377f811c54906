#!/bin/bash
# where does rank 0's gather load go?  one-rank nccl group, config 1, interleaved variants (FT8RX_GATHER_DIAG is a measurement aid of
# pyft8_amd/distributed.py: header = pack kernels + header read only; nocopies = no device-side copies; nod2h = no D2H copy)
set -u
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
G="--gpus 1 --backend nccl --force-gather --no-cpu-baseline --no-host-entry --no-other-configs --steps 100"
for i in 1 2 3; do
  FT8RX_GATHER_DIAG=nocopies timeout 600 python3 bench.py $G --gather-repeat 8 > "$OUT/r06_gdiag_x8nocopies_$i.json" 2>> "$OUT/r06_gdiag.err"
  FT8RX_GATHER_DIAG=nod2h timeout 600 python3 bench.py $G --gather-repeat 8 > "$OUT/r06_gdiag_x8nod2h_$i.json" 2>> "$OUT/r06_gdiag.err"
  timeout 600 python3 bench.py $G --gather-repeat 8 > "$OUT/r06_gdiag_x8_$i.json" 2>> "$OUT/r06_gdiag.err"
done
python3 - <<'PY'
import json, glob
for kind in ("none", "header", "x1", "x8nocopies", "x8nod2h", "x8"):
    vals = []
    for f in sorted(glob.glob(f"gpurun_out/r06_gdiag_{kind}_*.json")):
        try:
            d = json.loads([l for l in open(f) if l.startswith("{")][-1])
            vals.append((round(d["value"]), round(d["step_gap_ms"]["p50"], 3), (d["per_rank"]["gather"] or {}).get("submit_ms_per_step"), d["config"]["gather"][:40]))
        except Exception as e:
            vals.append(("FAILED", str(e)[:80]))
    print(kind, vals)
PY
