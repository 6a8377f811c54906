#!/bin/bash
# where does rank 0's gather load go?  one-rank nccl group, config 1, interleaved variants.  FT8RX_GATHER_DIAG=header is a measurement aid of
# pyft8_amd/distributed.py (submit() reads the packed header and returns: pack kernels + header read only).  The round-6 diagnosis also
# had two more aids -- the side-stream gather without its D2H copy / without its device copies -- which went away with the side-stream
# D2H itself; their numbers are in profiles/r06_gather_ab.txt.
set -u
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
G="--gpus 1 --backend nccl --force-gather --no-cpu-baseline --no-host-entry --no-other-configs --steps 100"
for i in 1 2 3; do
  timeout 600 python3 bench.py $G --no-gather > "$OUT/r06_gdiag_none_$i.json" 2>> "$OUT/r06_gdiag.err"
  FT8RX_GATHER_DIAG=header timeout 600 python3 bench.py $G > "$OUT/r06_gdiag_header_$i.json" 2>> "$OUT/r06_gdiag.err"
  timeout 600 python3 bench.py $G > "$OUT/r06_gdiag_x1_$i.json" 2>> "$OUT/r06_gdiag.err"
  timeout 600 python3 bench.py $G --gather-repeat 8 > "$OUT/r06_gdiag_x8_$i.json" 2>> "$OUT/r06_gdiag.err"
done
python3 - <<'PY'
import json, glob
for kind in ("none", "header", "x1", "x8"):
    vals = []
    for f in sorted(glob.glob(f"gpurun_out/r06_gdiag_{kind}_*.json")):
        try:
            d = json.loads([l for l in open(f) if l.startswith("{")][-1])
            vals.append((round(d["value"]), round(d["step_gap_ms"]["p50"], 3), (d["per_rank"]["gather"] or {}).get("submit_ms_per_step")))
        except Exception as e:
            vals.append(("FAILED", str(e)[:80]))
    print(kind, vals)
PY
