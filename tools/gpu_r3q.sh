#!/bin/bash
# round 3, call Q: where an entry's weights come from (LDS table / v_readlane / half and half) on the 128-plane flat forward, and the
# kernel without its atomics, at 1024 and 128 angles per launch
R=${GRAFT_REPO_ROOT:-$(pwd)}
rm -f $R/gpurun_out/r3q.log
for lib in "" "$R/build/ab2/libtomo_tabw0.so" "$R/build/ab2/libtomo_tabw2.so" "$R/build/ab2/libtomo_noatomic1.so"; do
  echo "== library: ${lib:-default}" | tee -a $R/gpurun_out/r3q.log
  TOMO_AB_LIB=$lib timeout -k 10 300 python3 tools/quick_bench.py fwd:1024:1024:tilt=0 fwd:1024:128:tilt=0 fwd:1024:128:tilt=0:shepp=1 fwd:512:128:tilt=0 2>&1 | tee -a $R/gpurun_out/r3q.log
done
