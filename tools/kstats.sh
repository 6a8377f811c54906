#!/bin/bash
# Per-kernel rocprofv3 statistics of one or more builds of libft8rx.so (GPU box):  tools/kstats.sh <tag> lib1.so [lib2.so ...]
# -> gpurun_out/<tag>_<name>_kstats.txt ; prints the kernels of each build side by side (average us per launch).
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
for L in "$@"; do
  N=$(basename "$L" .so)
  rm -rf "$OUT/${TAG}_${N}_st"
  FT8RX_LIB=$PWD/$L timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/${TAG}_${N}_st" -o s -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-entry --streams 1 --subbatch 0 --min-seconds 0 --no-other-configs > "$OUT/${TAG}_${N}_st.log" 2>&1
  DB=$(find "$OUT/${TAG}_${N}_st" -name '*.db' | head -1)
  [ -n "$DB" ] && python3 tools/rocprof_summary.py "$DB" "$OUT/${TAG}_${N}_kstats.txt" > /dev/null
  rm -rf "$OUT/${TAG}_${N}_st"
  echo "== $N"; head -24 "$OUT/${TAG}_${N}_kstats.txt"
done
