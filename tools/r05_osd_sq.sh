#!/bin/bash
# SQ counters of k_osd for the product build and the -DOSD_VISIT_ALL build
mkdir -p build/ab
python3 -c "
from pyft8_amd import _lib
_lib.build_variant('build/ab/osd_old.so', ['-DOSD_VISIT_ALL'])
"
bash tools/pmc_sq.sh r05osdnew > /dev/null 2>&1
bash tools/pmc_sq.sh r05osdold build/ab/osd_old.so > /dev/null 2>&1
for t in new old; do echo "== $t"; grep -E "^kernel|^k_osd |^k_bp |^k_fine " gpurun_out/r05osd${t}_sq.txt; done
