#!/bin/bash
mkdir -p gpurun_out/r05
for c in 3 4 2; do for sb in 128 192 256 384; do
  python bench.py --config $c --steps 8 --warmup 2 --subbatch $sb --no-cpu-baseline --no-host-entry --min-seconds 0 > gpurun_out/r05/z_c${c}_sub$sb.json 2> gpurun_out/r05/z_c${c}_sub$sb.err
  python - gpurun_out/r05/z_c${c}_sub$sb.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],2), "kernel_only", round(d["config"]["kernel_only_frames_per_s_this_rank"]))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done; done
