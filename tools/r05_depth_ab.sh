#!/bin/bash
# queue depth 1 (round-4 loop) against 2 (copy out, re-enqueue, then package), interleaved on one box, the driver's K and W
set -u
for i in 1 2 3 4 5; do
  for d in 1 2; do
    timeout 200 python3 bench.py --gpus 1 --steps ${1:-20} --warmup 5 --queue-depth $d --no-cpu-baseline --no-host-entry --no-other-configs --min-seconds 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('depth', $d, round(d['value']), {k: round(v,2) for k,v in d['step_gap_ms'].items()})"
  done
done
uptime
