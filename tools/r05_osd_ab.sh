#!/bin/bash
# round 5: k_osd elimination A/B -- systematic columns not visited (product) vs every position visited (-DOSD_VISIT_ALL), with phase timing
mkdir -p build/ab gpurun_out/r05
python3 -c "
from pyft8_amd import _lib
_lib.build_variant('build/ab/osd_t_new.so', ['-DOSD_TIMING'])
_lib.build_variant('build/ab/osd_t_old.so', ['-DOSD_TIMING', '-DOSD_VISIT_ALL'])
_lib.build_variant('build/ab/osd_old.so', ['-DOSD_VISIT_ALL'])
"
for v in new old; do echo "== $v"; FT8RX_LIB=build/ab/osd_t_$v.so python3 tools/osd_timing.py; done | tee gpurun_out/r05/osd_timing_ab.txt
bash tools/ab_variants.sh r05osd pyft8_amd/libft8rx.so build/ab/osd_old.so | tee gpurun_out/r05/osd_ab.txt
