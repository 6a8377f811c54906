#!/bin/bash
# round 5: sub-batch A/B (configs 2, 3 per GPU) + the new partition test
set -x
mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "subbatch or repeatable or config" 2>&1 | tail -5 > gpurun_out/r05/sub_tests.txt
for sb in 0 64 128 256; do
  python bench.py --config 3 --steps 5 --warmup 1 --subbatch $sb --no-cpu-baseline --no-host-entry > gpurun_out/r05/c3_sub$sb.json 2> gpurun_out/r05/c3_sub$sb.err
done
for sb in 0 128; do
  python bench.py --config 2 --steps 6 --warmup 1 --subbatch $sb --no-cpu-baseline --no-host-entry > gpurun_out/r05/c2_sub$sb.json 2> gpurun_out/r05/c2_sub$sb.err
done
python bench.py --no-cpu-baseline --no-host-entry --no-other-configs > gpurun_out/r05/c1.json 2> gpurun_out/r05/c1.err
for f in gpurun_out/r05/c*_sub*.json gpurun_out/r05/c1.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],2), "kernel_only", round(d["config"]["kernel_only_frames_per_s_this_rank"]), {k:v for k,v in d["stage_ms"].items() if v>0.05})
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
