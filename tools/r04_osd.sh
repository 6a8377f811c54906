#!/bin/bash
set -u
export TMPDIR=/tmp
cp pyft8_amd/libft8rx.so build/ab/osd0.so
# correctness of every variant first (the OSD parity tests through FT8RX_LIB)
for v in 1 2 3; do
  FT8RX_LIB=$PWD/build/ab/osd$v.so timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "osd or golden or sweep" 2>&1 | tail -1
done
tools/ab_variants.sh r04osd build/ab/osd0.so build/ab/osd1.so build/ab/osd2.so build/ab/osd3.so build/ab/osd0.so build/ab/osd3.so
