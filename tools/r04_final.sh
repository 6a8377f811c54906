#!/bin/bash
# final validation of the round on the final build: GPU tests, 8000-frame parity sweep, smoke, a sustained bench run
set -u
OUT=$PWD/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests -m gpu -q > "$OUT/r04_gpu_tests.log" 2>&1; tail -3 "$OUT/r04_gpu_tests.log"
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 2400 python3 tools/parity_sweep.py 500 16 > "$OUT/r04_parity_sweep_8000frames.txt" 2>&1; tail -2 "$OUT/r04_parity_sweep_8000frames.txt"
timeout 900 python3 bench.py --min-seconds 300 --no-cpu-baseline --no-other-configs > "$OUT/r04_bench_b256_5min.json" 2> /dev/null
python3 - "$OUT/r04_bench_b256_5min.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("5 min:", round(d["value"]), d["extra_steps"], round(d["extra_steps_frames_per_s"]))
PY
timeout 600 python3 -c "from pyft8_amd import _lib; _lib.build_variant('build/ab/fine_timing.so', ['-DFINE_TIMING'])" > /dev/null 2>&1 && FT8RX_LIB=build/ab/fine_timing.so timeout 300 python3 tools/fine_timing.py > "$OUT/r04_fine_timing.txt" 2>&1
timeout 600 python3 -c "from pyft8_amd import _lib; _lib.build_variant('build/ab/osd_timing.so', ['-DOSD_TIMING'])" > /dev/null 2>&1 && FT8RX_LIB=build/ab/osd_timing.so timeout 300 python3 tools/osd_timing.py > "$OUT/r04_osd_timing.txt" 2>&1
timeout 600 python3 -c "from pyft8_amd import _lib; _lib.build_variant('build/ab/bp_timing.so', ['-DBP_TIMING'])" > /dev/null 2>&1 && FT8RX_LIB=build/ab/bp_timing.so timeout 300 python3 tools/bp_timing.py > "$OUT/r04_bp_timing.txt" 2>&1
timeout 900 bash tools/ab_streams.sh > "$OUT/r04_streams_ab.txt" 2>&1
timeout 900 python3 bench.py > "$OUT/r04_bench_b256_default_final.json" 2> /dev/null
python3 - "$OUT/r04_bench_b256_default_final.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("default:", round(d["value"]), round(d["value_incl_h2d"]), {k: round(v["value"]) for k, v in d["other_configs"].items()})
PY
