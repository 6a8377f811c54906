#!/bin/bash
set -u
OUT=$PWD/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 2400 python3 -m pytest tests -m gpu -x -q > "$OUT/r04_gpu_tests.log" 2>&1; echo "gpu tests rc=$?"
tail -8 "$OUT/r04_gpu_tests.log"
run() { name=$1; shift; timeout 600 "$@" > "$OUT/$name.json" 2>> "$OUT/r04c.err"; echo "$name rc=$?"; }
run r04d_default python3 bench.py
run r04d_c1_x8 python3 bench.py --gpus 1 --backend nccl --force-gather --no-cpu-baseline --no-host-entry --no-other-configs --gather-repeat 8
run r04d_c1_gather python3 bench.py --gpus 1 --backend nccl --force-gather --no-cpu-baseline --no-host-entry --no-other-configs
for f in r04d_default r04d_c1_x8 r04d_c1_gather; do
  python3 - "$OUT/$f.json" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    g = d["per_rank"].get("gather") or {}
    print(f'{sys.argv[1].split("/")[-1]:34s} value {d["value"]:9.0f}  ms/step {d["ms_per_step"]:8.3f}  kernel_only {d["config"]["kernel_only_frames_per_s_this_rank"]:9.0f}  h2d {d["value_incl_h2d"]} submit {g.get("submit_ms_per_step")} unoverlapped {g.get("unoverlapped_ms")} ok {g.get("ok")}')
    print("   stages", d["stage_ms"])
    for k, v in (d.get("other_configs") or {}).items():
        print("   other", k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a in ("value", "ms_per_step", "steps", "setup_s", "error")})
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
tail -3 "$OUT"/r04c.err
