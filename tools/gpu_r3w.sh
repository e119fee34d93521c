#!/bin/bash
# round 3, call W: waves per work-group of the gather back-projection (they share the per-projection weight table), 1024 angles per launch
R=${GRAFT_REPO_ROOT:-$(pwd)}
rm -f $R/gpurun_out/r3w.log
for lib in "" $R/build/ab2/libtomo_gw8.so $R/build/ab2/libtomo_gw16.so $R/build/ab2/libtomo_gw2.so; do
  echo "== library: ${lib:-default (4 waves)}" | tee -a $R/gpurun_out/r3w.log
  TOMO_AB_LIB=$lib timeout -k 10 300 python3 tools/quick_bench.py adj:1024:1024:tilt=0 adj:1024:128:tilt=0 adj:512:128:tilt=0 2>&1 | tee -a $R/gpurun_out/r3w.log
done
