#!/bin/bash
# how much of k_osd's DURATION each phase really accounts for: timing-only builds that skip one phase (results are wrong, times are not)
mkdir -p build/ab
python3 -c "
from pyft8_amd import _lib
_lib.build_variant('build/ab/osd_noelim.so', ['-DOSD_TIMING_SKIP_ELIM'])
_lib.build_variant('build/ab/osd_nosort.so', ['-DOSD_TIMING_SKIP_SORT'])
_lib.build_variant('build/ab/osd_notrials.so', ['-DOSD_TIMING_SKIP_TRIALS'])
_lib.build_variant('build/ab/osd_old.so', ['-DOSD_VISIT_ALL'])
_lib.build_variant('build/ab/osd_old_noelim.so', ['-DOSD_VISIT_ALL', '-DOSD_TIMING_SKIP_ELIM'])
"
bash tools/ab_variants.sh r05skip pyft8_amd/libft8rx.so build/ab/osd_noelim.so build/ab/osd_nosort.so build/ab/osd_notrials.so build/ab/osd_old.so build/ab/osd_old_noelim.so 2>&1 | grep -E "stage|frames|osd|bp_fine"
