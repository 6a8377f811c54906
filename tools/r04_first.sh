#!/bin/bash
# first GPU pass of round 4: the packed-gather tests, the default bench line (with other_configs), the config-3 gather A/B
set -u
OUT=$PWD/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 1700 python3 -m pytest tests/test_gpu_multirank.py -x -q > "$OUT/r04a_multirank.log" 2>&1; echo "multirank rc=$?" 
tail -5 "$OUT/r04a_multirank.log"
timeout 600 python3 bench.py > "$OUT/r04a_bench_default.json" 2> "$OUT/r04a_bench_default.err"; echo "bench rc=$?"
G="python3 bench.py --gpus 1 --backend nccl --force-gather --config 3 --no-cpu-baseline --no-host-entry --steps 4"
timeout 600 $G > "$OUT/r04a_c3_gather.json" 2> "$OUT/r04a_c3.err"; echo "c3 gather rc=$?"
timeout 600 $G --no-gather > "$OUT/r04a_c3_nogather.json" 2>> "$OUT/r04a_c3.err"; echo "c3 nogather rc=$?"
timeout 600 $G --gather-repeat 8 > "$OUT/r04a_c3_gather_x8.json" 2>> "$OUT/r04a_c3.err"; echo "c3 x8 rc=$?"
timeout 600 $G --gather-repeat 8 --render-gathered > "$OUT/r04a_c3_gather_x8_render.json" 2>> "$OUT/r04a_c3.err"; echo "c3 x8 render rc=$?"
timeout 900 python3 bench.py --gpus 8 --backend gloo --frames 1024 --steps 3 --warmup 1 --no-host-entry --min-seconds 0 > "$OUT/r04a_8ranks_gloo_b1024.json" 2> "$OUT/r04a_8ranks.err"; echo "8 ranks rc=$?"
for f in r04a_bench_default r04a_c3_gather r04a_c3_nogather r04a_c3_gather_x8 r04a_c3_gather_x8_render r04a_8ranks_gloo_b1024; do
  python3 - "$OUT/$f.json" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1].split("/")[-1], round(d["value"]), round(d["ms_per_step"], 3), d["config"]["gather"][:200], d["per_rank"].get("gather"))
    if d.get("other_configs"):
        for k, v in d["other_configs"].items():
            print("   other", k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a in ("value", "ms_per_step", "steps", "setup_s", "error")})
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
tail -3 "$OUT"/r04a_*.err
