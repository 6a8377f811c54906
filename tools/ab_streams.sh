#!/bin/bash
# stream-count A/B on one box: python bench.py --streams N, twice each
for rep in 1 2; do for s in 1 2 3 4; do
  python3 bench.py --no-other-configs --no-cpu-baseline --no-host-entry --steps 100 --warmup 10 --streams $s 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('streams', $s, round(d['value']), round(d['ms_per_step'],3), 'stage sum', round(sum(d['stage_ms'].values()),3), 'fine', d['stage_ms']['fine'])"
done; done
