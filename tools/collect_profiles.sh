#!/bin/bash
# Collect the evidence kept under profiles/ on a GPU box (run from the repo root through gpurun):
#   tools/collect_profiles.sh <tag>        e.g. r01 -> gpurun_out/<tag>_*
# rocprofv3 passes are separate runs (kernel trace + stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE), each with the
# program itself after `--`.  Everything is bounded by `timeout`.
set -u
TAG=${1:-r05}
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-entry --streams 1 --subbatch 0 --min-seconds 0 --no-other-configs"

timeout 900 python3 bench.py > "$OUT/${TAG}_bench_b256_default.json" 2> "$OUT/${TAG}_bench_b256_default.err"
timeout 600 python3 bench.py --streams 4 --no-cpu-baseline --no-other-configs > "$OUT/${TAG}_bench_b256_s4.json" 2>> "$OUT/${TAG}_bench_b256_default.err"
timeout 600 python3 bench.py --streams 1 --no-cpu-baseline --no-other-configs > "$OUT/${TAG}_bench_b256_s1.json" 2>> "$OUT/${TAG}_bench_b256_default.err"

rm -rf "$OUT/${TAG}_stats" "$OUT/${TAG}_pmcF" "$OUT/${TAG}_pmcW"
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/${TAG}_stats" -o s -- $BENCH > "$OUT/${TAG}_stats.log" 2>&1
DB=$(find "$OUT/${TAG}_stats" -name '*.db' | head -1)
[ -n "$DB" ] && python3 tools/rocprof_summary.py "$DB" "$OUT/${TAG}_kernel_stats_b256.txt" > /dev/null
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/${TAG}_pmcF" -o f -- $BENCH > "$OUT/${TAG}_pmcF.log" 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/${TAG}_pmcW" -o w -- $BENCH > "$OUT/${TAG}_pmcW.log" 2>&1
F=$(find "$OUT/${TAG}_pmcF" -name '*counter_collection.csv' | head -1)
W=$(find "$OUT/${TAG}_pmcW" -name '*counter_collection.csv' | head -1)
[ -n "$F" ] && [ -n "$W" ] && python3 tools/pmc_summary.py "$F" "$W" "$OUT/${TAG}_pmc.json" "$OUT/${TAG}_pmc_b256.txt" > /dev/null

# BASELINE config 2 as stated (4096 frames, BP 30 iterations, OSD depth 2 = osd_012(30, 2)): rocprofv3 kernel stats + PMC traffic
B2="python3 bench.py --config 2 --steps 2 --warmup 1 --no-cpu-baseline --no-host-entry --streams 1 --subbatch 0 --min-seconds 0"
rm -rf "$OUT/${TAG}_c2_stats" "$OUT/${TAG}_c2_pmcF" "$OUT/${TAG}_c2_pmcW"
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/${TAG}_c2_stats" -o s -- $B2 > "$OUT/${TAG}_c2_stats.log" 2>&1
DB=$(find "$OUT/${TAG}_c2_stats" -name '*.db' | head -1)
[ -n "$DB" ] && python3 tools/rocprof_summary.py "$DB" "$OUT/${TAG}_kernel_stats_b4096_config2.txt" > /dev/null
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/${TAG}_c2_pmcF" -o f -- $B2 > "$OUT/${TAG}_c2_pmcF.log" 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/${TAG}_c2_pmcW" -o w -- $B2 > "$OUT/${TAG}_c2_pmcW.log" 2>&1
F=$(find "$OUT/${TAG}_c2_pmcF" -name '*counter_collection.csv' | head -1)
W=$(find "$OUT/${TAG}_c2_pmcW" -name '*counter_collection.csv' | head -1)
[ -n "$F" ] && [ -n "$W" ] && python3 tools/pmc_summary.py "$F" "$W" "$OUT/${TAG}_c2_pmc.json" "$OUT/${TAG}_pmc_b4096_config2.txt" > /dev/null
timeout 900 $B2 > "$OUT/${TAG}_bench_b4096_config2_bp30.json" 2>> "$OUT/${TAG}_big.err"

# PMC traffic of configs 3 and 4 (bench.py --config 3 / 4 fill roofline.traffic from these)
for C in 3 4; do
  BC="python3 bench.py --config $C --steps 2 --warmup 1 --no-cpu-baseline --no-host-entry --streams 1 --subbatch 0 --min-seconds 0"
  rm -rf "$OUT/${TAG}_c${C}_pmcF" "$OUT/${TAG}_c${C}_pmcW"
  timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/${TAG}_c${C}_pmcF" -o f -- $BC > "$OUT/${TAG}_c${C}_pmcF.log" 2>&1
  timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/${TAG}_c${C}_pmcW" -o w -- $BC > "$OUT/${TAG}_c${C}_pmcW.log" 2>&1
  F=$(find "$OUT/${TAG}_c${C}_pmcF" -name '*counter_collection.csv' | head -1)
  W=$(find "$OUT/${TAG}_c${C}_pmcW" -name '*counter_collection.csv' | head -1)
  [ -n "$F" ] && [ -n "$W" ] && python3 tools/pmc_summary.py "$F" "$W" "$OUT/${TAG}_c${C}_pmc.json" "$OUT/${TAG}_pmc_config${C}.txt" > /dev/null
done

# other BASELINE configurations (single GPU): config 2 (B = 4096; reference knobs and extension knobs), the config-3 shard
# size (8192 frames per GPU) and config 4 (low SNR, few signals, truth-based decode probability)
timeout 900 python3 bench.py --frames 4096 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/${TAG}_bench_b4096_ref_knobs.json" 2> "$OUT/${TAG}_big.err"
timeout 900 python3 bench.py --frames 4096 --steps 3 --warmup 1 --no-cpu-baseline --bp-iters 30 --osd 40 4 > "$OUT/${TAG}_bench_b4096_ext_knobs.json" 2>> "$OUT/${TAG}_big.err"
timeout 900 python3 bench.py --config 3 --no-cpu-baseline > "$OUT/${TAG}_bench_config3_shard_b8192.json" 2>> "$OUT/${TAG}_big.err"
timeout 900 python3 bench.py --config 4 --no-cpu-baseline > "$OUT/${TAG}_bench_config4_b2048_order3_gate.json" 2>> "$OUT/${TAG}_big.err"
timeout 900 python3 bench.py --frames 4096 --steps 3 --warmup 1 --no-cpu-baseline --signals 8 --snr -24 -14 > "$OUT/${TAG}_bench_b4096_lowsnr.json" 2>> "$OUT/${TAG}_big.err"
# two ranks on this one GPU over gloo: the N > 1 flow of bench.py (sharding, barriers, the packed gather inside the steps) -- not a scaling number
timeout 900 python3 bench.py --gpus 2 --backend gloo --no-host-entry > "$OUT/${TAG}_bench_2ranks_1gpu_gloo.json" 2>> "$OUT/${TAG}_big.err"
# eight ranks on this one GPU over gloo: the rank-count-dependent paths at world size 8 (uneven total; 1024 frames per rank)
timeout 900 python3 bench.py --gpus 8 --backend gloo --total-frames 131 --steps 2 --warmup 1 --no-host-entry --min-seconds 0 > "$OUT/${TAG}_bench_8ranks_1gpu_gloo_uneven131.json" 2>> "$OUT/${TAG}_big.err"
timeout 900 python3 bench.py --gpus 8 --backend gloo --frames 1024 --steps 3 --warmup 1 --no-host-entry --min-seconds 0 > "$OUT/${TAG}_bench_8ranks_1gpu_gloo_b1024.json" 2>> "$OUT/${TAG}_big.err"
# the RCCL gather in a one-rank nccl group on this one GPU, INSIDE the timed steps: config-3 shard and config 1, each with the gather,
# without it, and with rank 0's eight-rank load (--gather-repeat 8)
G="--gpus 1 --backend nccl --force-gather --no-cpu-baseline --no-host-entry --no-other-configs"
timeout 900 python3 bench.py $G --config 3 --steps 4 > "$OUT/${TAG}_bench_config3_rccl_gather_in_step.json" 2>> "$OUT/${TAG}_big.err"
timeout 900 python3 bench.py $G --config 3 --steps 4 --no-gather > "$OUT/${TAG}_bench_config3_rccl_no_gather.json" 2>> "$OUT/${TAG}_big.err"
timeout 900 python3 bench.py $G --config 3 --steps 4 --gather-repeat 8 > "$OUT/${TAG}_bench_config3_rccl_gather_x8.json" 2>> "$OUT/${TAG}_big.err"
timeout 900 python3 bench.py $G --config 3 --steps 4 --gather-repeat 8 --render-gathered > "$OUT/${TAG}_bench_config3_rccl_gather_x8_rank0_renders_all.json" 2>> "$OUT/${TAG}_big.err"
timeout 900 python3 bench.py $G > "$OUT/${TAG}_bench_config1_rccl_gather_in_step.json" 2>> "$OUT/${TAG}_big.err"
timeout 900 python3 bench.py $G --no-gather > "$OUT/${TAG}_bench_config1_rccl_no_gather.json" 2>> "$OUT/${TAG}_big.err"
timeout 900 python3 bench.py $G --gather-repeat 8 > "$OUT/${TAG}_bench_config1_rccl_gather_x8.json" 2>> "$OUT/${TAG}_big.err"
python3 tools/kernel_resources.py > "$OUT/${TAG}_kernel_resources.txt" 2>> "$OUT/${TAG}_big.err"
timeout 300 tools/ubench/mfma_valu_overlap > "$OUT/${TAG}_mfma_valu_overlap.txt" 2>&1
timeout 300 python3 tools/host_breakdown.py > "$OUT/${TAG}_host_breakdown.txt" 2>> "$OUT/${TAG}_big.err"
# SQ counters per kernel (three --pmc passes)
timeout 1500 tools/pmc_sq.sh "${TAG}" > /dev/null 2>&1
timeout 1500 python3 tools/sensitivity.py 2048 -24 -20 > "$OUT/${TAG}_config4_sensitivity.txt" 2>> "$OUT/${TAG}_big.err"
timeout 300 tools/ubench/valu_rate > "$OUT/${TAG}_valu_rate.txt" 2>&1
timeout 300 python3 tools/latency.py > "$OUT/${TAG}_latency.txt" 2>> "$OUT/${TAG}_big.err"
timeout 600 python3 tools/two_pass_yield.py 256 50 2>&1 | tail -7 > "$OUT/${TAG}_multi_pass_yield.txt"
timeout 1200 python3 tools/sensitivity.py 2048 > "$OUT/${TAG}_sensitivity_wide.txt" 2>> "$OUT/${TAG}_big.err"
timeout 900 python3 tools/multipass_profile.py > "$OUT/${TAG}_multipass_profile.txt" 2>> "$OUT/${TAG}_big.err"
timeout 1500 python3 tools/parity_sweep.py 130 16 > "$OUT/${TAG}_parity_sweep.txt" 2>> "$OUT/${TAG}_big.err"
timeout 400 python3 tools/soak.py 120 > "$OUT/${TAG}_soak.txt" 2>> "$OUT/${TAG}_big.err"
timeout 1500 python3 -m pytest tests -m gpu -q > "$OUT/${TAG}_gpu_tests.log" 2>&1
ls -la "$OUT" | grep "${TAG}_" | head -60
