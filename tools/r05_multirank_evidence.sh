#!/bin/bash
# the multi-rank flow evidence of tools/collect_profiles.sh on its own (final build): gloo ranks sharing the one GPU, and the RCCL
# gather in a one-rank nccl group inside the timed steps -> gpurun_out/r05_bench_*ranks*.json, r05_bench_config*_rccl_*.json
set -u
TAG=r05; OUT=$PWD/gpurun_out; mkdir -p "$OUT"
timeout 900 python3 bench.py --gpus 2 --backend gloo --no-host-entry > "$OUT/${TAG}_bench_2ranks_1gpu_gloo.json" 2>> "$OUT/${TAG}_big.err"
timeout 900 python3 bench.py --gpus 8 --backend gloo --total-frames 131 --steps 2 --warmup 1 --no-host-entry --min-seconds 0 > "$OUT/${TAG}_bench_8ranks_1gpu_gloo_uneven131.json" 2>> "$OUT/${TAG}_big.err"
timeout 900 python3 bench.py --gpus 8 --backend gloo --frames 1024 --steps 3 --warmup 1 --no-host-entry --min-seconds 0 > "$OUT/${TAG}_bench_8ranks_1gpu_gloo_b1024.json" 2>> "$OUT/${TAG}_big.err"
G="--gpus 1 --backend nccl --force-gather --no-cpu-baseline --no-host-entry --no-other-configs"
timeout 900 python3 bench.py $G --config 3 --steps 4 > "$OUT/${TAG}_bench_config3_rccl_gather_in_step.json" 2>> "$OUT/${TAG}_big.err"
timeout 900 python3 bench.py $G --config 3 --steps 4 --no-gather > "$OUT/${TAG}_bench_config3_rccl_no_gather.json" 2>> "$OUT/${TAG}_big.err"
timeout 900 python3 bench.py $G --config 3 --steps 4 --gather-repeat 8 > "$OUT/${TAG}_bench_config3_rccl_gather_x8.json" 2>> "$OUT/${TAG}_big.err"
timeout 900 python3 bench.py $G --config 3 --steps 4 --gather-repeat 8 --render-gathered > "$OUT/${TAG}_bench_config3_rccl_gather_x8_rank0_renders_all.json" 2>> "$OUT/${TAG}_big.err"
timeout 900 python3 bench.py $G > "$OUT/${TAG}_bench_config1_rccl_gather_in_step.json" 2>> "$OUT/${TAG}_big.err"
timeout 900 python3 bench.py $G --no-gather > "$OUT/${TAG}_bench_config1_rccl_no_gather.json" 2>> "$OUT/${TAG}_big.err"
timeout 900 python3 bench.py $G --gather-repeat 8 > "$OUT/${TAG}_bench_config1_rccl_gather_x8.json" 2>> "$OUT/${TAG}_big.err"
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_bench_*ranks*.json") + glob.glob("gpurun_out/r05_bench_config*_rccl_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], d["n_gpus"], round(d["value"]), d["config"].get("gather", "")[:90])
    except Exception as e:
        print(f, "FAILED", e)
PY
