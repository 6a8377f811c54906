#!/bin/bash
# final validation + evidence of round 6 on the final build (one gpurun call):
#   GPU tests, smoke, 8000-frame parity sweep, sustained run, in-kernel phase timings, kernel stats, counter passes (PMC traffic for
#   configs 1-4, SQ), the default bench line
set -u
TAG=r06
OUT=$PWD/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests -m gpu -q > "$OUT/${TAG}_gpu_tests.log" 2>&1; tail -3 "$OUT/${TAG}_gpu_tests.log"
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 2400 python3 tools/parity_sweep.py 528 16 > "$OUT/${TAG}_parity_sweep_8448frames.txt" 2>&1; tail -2 "$OUT/${TAG}_parity_sweep_8448frames.txt"
timeout 400 python3 tools/soak.py 120 > "$OUT/${TAG}_soak.txt" 2>&1; tail -2 "$OUT/${TAG}_soak.txt"
timeout 300 python3 tools/sandbox_overlap.py > "$OUT/${TAG}_multipass_vs_reference_sandbox.txt" 2>&1; tail -3 "$OUT/${TAG}_multipass_vs_reference_sandbox.txt"
timeout 900 python3 bench.py --min-seconds 300 --no-cpu-baseline --no-other-configs > "$OUT/${TAG}_bench_b256_5min.json" 2> /dev/null
python3 - "$OUT/${TAG}_bench_b256_5min.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("5 min:", round(d["value"]), d["extra_steps"], round(d["extra_steps_frames_per_s"]))
PY
timeout 600 python3 -c "from pyft8_amd import _lib; _lib.build_variant('build/ab/fine_timing.so', ['-DFINE_TIMING'])" > /dev/null 2>&1 && FT8RX_LIB=build/ab/fine_timing.so timeout 300 python3 tools/fine_timing.py > "$OUT/${TAG}_fine_timing.txt" 2>&1
timeout 600 python3 -c "from pyft8_amd import _lib; _lib.build_variant('build/ab/osd_timing.so', ['-DOSD_TIMING'])" > /dev/null 2>&1 && FT8RX_LIB=build/ab/osd_timing.so timeout 300 python3 tools/osd_timing.py > "$OUT/${TAG}_osd_timing.txt" 2>&1
timeout 600 python3 -c "from pyft8_amd import _lib; _lib.build_variant('build/ab/bp_timing.so', ['-DBP_TIMING'])" > /dev/null 2>&1 && FT8RX_LIB=build/ab/bp_timing.so timeout 300 python3 tools/bp_timing.py > "$OUT/${TAG}_bp_timing.txt" 2>&1
# kernel stats + counters of `bench.py --streams 1 --subbatch 0` (whole 256-frame launches)
BENCH="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-entry --streams 1 --subbatch 0 --min-seconds 0 --no-other-configs"
rm -rf "$OUT/${TAG}_stats" "$OUT/${TAG}_pmcF" "$OUT/${TAG}_pmcW"
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/${TAG}_stats" -o s -- $BENCH > "$OUT/${TAG}_stats.log" 2>&1
DB=$(find "$OUT/${TAG}_stats" -name '*.db' | head -1)
[ -n "$DB" ] && python3 tools/rocprof_summary.py "$DB" "$OUT/${TAG}_kernel_stats_b256.txt" > /dev/null
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/${TAG}_pmcF" -o f -- $BENCH > "$OUT/${TAG}_pmcF.log" 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/${TAG}_pmcW" -o w -- $BENCH > "$OUT/${TAG}_pmcW.log" 2>&1
F=$(find "$OUT/${TAG}_pmcF" -name '*counter_collection.csv' | head -1); W=$(find "$OUT/${TAG}_pmcW" -name '*counter_collection.csv' | head -1)
[ -n "$F" ] && [ -n "$W" ] && python3 tools/pmc_summary.py "$F" "$W" "$OUT/${TAG}_pmc.json" "$OUT/${TAG}_pmc_b256.txt" > /dev/null
for C in 2 3 4; do
  BC="python3 bench.py --config $C --steps 2 --warmup 1 --no-cpu-baseline --no-host-entry --streams 1 --subbatch 0 --min-seconds 0"
  rm -rf "$OUT/${TAG}_c${C}_pmcF" "$OUT/${TAG}_c${C}_pmcW"
  timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/${TAG}_c${C}_pmcF" -o f -- $BC > "$OUT/${TAG}_c${C}_pmcF.log" 2>&1
  timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/${TAG}_c${C}_pmcW" -o w -- $BC > "$OUT/${TAG}_c${C}_pmcW.log" 2>&1
  F=$(find "$OUT/${TAG}_c${C}_pmcF" -name '*counter_collection.csv' | head -1); W=$(find "$OUT/${TAG}_c${C}_pmcW" -name '*counter_collection.csv' | head -1)
  N=$([ $C = 2 ] && echo b4096_config2 || echo config$C)
  [ -n "$F" ] && [ -n "$W" ] && python3 tools/pmc_summary.py "$F" "$W" "$OUT/${TAG}_c${C}_pmc.json" "$OUT/${TAG}_pmc_${N}.txt" > /dev/null
done
timeout 1500 tools/pmc_sq.sh "${TAG}" > /dev/null 2>&1
python3 tools/kernel_resources.py > "$OUT/${TAG}_kernel_resources.txt" 2>/dev/null
# the counter profiles bench.py reads must be in place BEFORE the final default line is taken
cp "$OUT/${TAG}_pmc.json" profiles/pmc_latest.json; cp "$OUT/${TAG}_sq.json" profiles/sq_latest.json
for C in 2 3 4; do cp "$OUT/${TAG}_c${C}_pmc.json" profiles/pmc_config${C}_latest.json; done
timeout 900 python3 bench.py > "$OUT/${TAG}_bench_b256_default.json" 2> "$OUT/${TAG}_bench_b256_default.err"
python3 - "$OUT/${TAG}_bench_b256_default.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("default:", round(d["value"]), round(d["value_incl_h2d"]), {k: round(v["value"]) for k, v in d["other_configs"].items()}, d["roofline"]["frac"], d["roofline"]["traffic_stale"], d["roofline_valu"] and (d["roofline_valu"]["frac"], d["roofline_valu"]["step_frac"], d["roofline_valu"]["stale"]))
print(d["stage_ms"])
PY
head -12 "$OUT/${TAG}_kernel_stats_b256.txt"
