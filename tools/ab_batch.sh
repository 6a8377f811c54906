#!/bin/bash
# per-256-frame stage times and rates by batch size (one box)
for b in 128 256 512 1024 2048 4096; do
  python3 bench.py --no-other-configs --no-cpu-baseline --no-host-entry --frames $b --steps $((25600 / b > 100 ? 100 : (25600 / b < 4 ? 4 : 25600 / b))) --warmup 3 --min-seconds 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b=$b; s=d['stage_ms']
print('B', b, 'value', round(d['value']), 'kernels only', round(d['per_rank']['kernel_only_frames_per_s'][0]), 'per 256 frames:', {k: round(v*256/b,4) for k,v in s.items() if v*256/b>0.05}, 'sum', round(sum(s.values())*256/b,3))"
done
