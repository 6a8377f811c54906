#!/bin/bash
# SQ-counter passes per kernel (rocprofv3 --pmc, counters only; kernel trace needed for kernel names):
#   tools/pmc_sq.sh <tag> [libft8rx variant .so]   -> gpurun_out/<tag>_sq.txt (+ raw csv dirs)
# Three passes of <= 8 SQ counters each over `bench.py --streams 1` (whole-batch launches, B = 256).
set -u
TAG=$1; LIBV=${2:-}
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp
[ -n "$LIBV" ] && export FT8RX_LIB=$PWD/$LIBV
BENCH="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-entry --streams 1 --subbatch 0 --min-seconds 0 --no-other-configs"
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS"
P3="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_INSTS_SENDMSG SQ_INSTS_FLAT SQ_IFETCH"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1)); rm -rf "$OUT/${TAG}_sq$i"
  timeout 900 rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$OUT/${TAG}_sq$i" -o p -- $BENCH > "$OUT/${TAG}_sq$i.log" 2>&1
done
python3 tools/pmc_sq_summary.py "$OUT/${TAG}_sq.txt" $(find "$OUT/${TAG}_sq1" "$OUT/${TAG}_sq2" "$OUT/${TAG}_sq3" -name '*counter_collection.csv')
