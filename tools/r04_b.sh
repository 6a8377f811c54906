#!/bin/bash
set -u
OUT=$PWD/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
G3="python3 bench.py --gpus 1 --backend nccl --force-gather --config 3 --no-cpu-baseline --no-host-entry --steps 4 --no-other-configs"
G1="python3 bench.py --gpus 1 --backend nccl --force-gather --no-cpu-baseline --no-host-entry --no-other-configs"
run() { name=$1; shift; timeout 600 "$@" > "$OUT/$name.json" 2>> "$OUT/r04b.err"; echo "$name rc=$?"; }
run r04c_c3 $G3
run r04c_c3_nogather $G3 --no-gather
run r04c_c3_x8 $G3 --gather-repeat 8
run r04c_c1 $G1
run r04c_c1_nogather $G1 --no-gather
run r04c_c1_x8 $G1 --gather-repeat 8
run r04c_8ranks_gloo_b1024 python3 bench.py --gpus 8 --backend gloo --frames 1024 --steps 3 --warmup 1 --no-host-entry --min-seconds 0
for f in r04c_c3 r04c_c3_nogather r04c_c3_x8 r04c_c1 r04c_c1_nogather r04c_c1_x8 r04c_8ranks_gloo_b1024; do
  python3 - "$OUT/$f.json" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    g = d["per_rank"].get("gather") or {}
    print(f'{sys.argv[1].split("/")[-1]:34s} value {d["value"]:9.0f}  ms/step {d["ms_per_step"]:8.3f}  kernel_only {d["config"]["kernel_only_frames_per_s_this_rank"]:9.0f}  submit {g.get("submit_ms_per_step")} phases {g.get("submit_phases_ms")} unoverlapped {g.get("unoverlapped_ms")} B/frame {g.get("bytes_per_frame")} ok {g.get("ok")}')
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
tail -3 "$OUT"/r04b.err
