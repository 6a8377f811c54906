#!/bin/bash
set -u
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
timeout 1500 python3 -m pytest tests -m gpu -q -x 2>&1 | tail -6
timeout 1500 python3 tools/parity_sweep.py 130 16 > "$OUT/r04fs_parity_sweep.txt" 2>&1; tail -2 "$OUT/r04fs_parity_sweep.txt"
timeout 600 python3 bench.py > "$OUT/r04fs_bench_default.json" 2>/dev/null
python3 - "$OUT/r04fs_bench_default.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("default:", round(d["value"]), round(d["ms_per_step"], 3), "h2d", round(d["value_incl_h2d"]), "sync", round(d["value_incl_h2d_sync_call"]), "kernels", round(d["config"]["kernel_only_frames_per_s_this_rank"]))
print(d["stage_ms"]); print("roofline", d["roofline"]["kernel"], d["roofline"]["frac"])
print({k: (round(v["value"]), round(v["ms_per_step"], 2)) for k, v in d["other_configs"].items()})
PY
