#!/bin/bash
# instruction-fetch counters per kernel (one rocprofv3 --pmc pass, counters only + kernel trace for names)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; TAG=${1:-r05ic}
rm -rf $OUT/${TAG}_ic
BENCH="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-entry --streams 1 --min-seconds 0 --no-other-configs"
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/${TAG}_ic -o p -- $BENCH > $OUT/${TAG}_ic.log 2>&1
tail -3 $OUT/${TAG}_ic.log
python3 - $OUT/${TAG}_ic <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f: print("no counter file"); sys.exit()
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen = set()
for row in csv.DictReader(open(f[0])):
    k = row["Kernel_Name"].split("(")[0]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    key = (k, row["Dispatch_Id"])
    if key not in seen: seen.add(key); n[k] += 1
names = ["SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_IFETCH", "SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU"]
print("%-18s" % "kernel" + "".join("%18s" % x[3:] for x in names))
for k in sorted(acc, key=lambda k: -acc[k]["SQ_WAVE_CYCLES"])[:8]:
    print("%-18s" % k[:17] + "".join("%18.4g" % (acc[k][x] / n[k]) for x in names))
PY
