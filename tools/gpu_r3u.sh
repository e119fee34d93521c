#!/bin/bash
# round 3, call U: the gather back-projection's XCD patch shape (GPX x GPY tiles of 8 x 8 columns), 1024 angles per launch
R=${GRAFT_REPO_ROOT:-$(pwd)}
rm -f $R/gpurun_out/r3u.log
for lib in "" $R/build/ab2/libtomo_gp_8x16.so $R/build/ab2/libtomo_gp_16x8.so $R/build/ab2/libtomo_gp_4x32.so $R/build/ab2/libtomo_gp_12x12.so $R/build/ab2/libtomo_gp_6x16.so; do
  echo "== library: ${lib:-default (8 x 12)}" | tee -a $R/gpurun_out/r3u.log
  TOMO_AB_LIB=$lib timeout -k 10 300 python3 tools/quick_bench.py adj:1024:1024:tilt=0 adj:1024:128:tilt=0 2>&1 | tee -a $R/gpurun_out/r3u.log
done
