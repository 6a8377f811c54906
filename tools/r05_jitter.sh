#!/bin/bash
# how much does the default command's `value` move from run to run on one box?  N fresh processes, the timed region only
set -u
N=${1:-8}; STEPS=${2:-100}
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
for i in $(seq 1 $N); do
  timeout 300 python3 bench.py --steps $STEPS --warmup 10 --no-cpu-baseline --no-host-entry --no-other-configs --min-seconds 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(round(d['value']), round(d['per_rank']['kernel_only_frames_per_s'][0]), {k: round(v,2) if isinstance(v,float) else v for k,v in d['step_gap_ms'].items()})"
done
