#!/bin/bash
# ablation: the round-3 flat forward without its float atomics (1) / with plain stores in their place (2: wrong sums, right traffic)
R=${GRAFT_REPO_ROOT:-$(pwd)}
rm -f $R/gpurun_out/r3l.log
for lib in "" "$R/build/ab2/libtomo_noatomic1.so" "$R/build/ab2/libtomo_noatomic2.so"; do
  echo "== library: ${lib:-default}" | tee -a $R/gpurun_out/r3l.log
  TOMO_AB_LIB=$lib timeout -k 10 300 python3 tools/quick_bench.py fwd:1024:128:tilt=0 fwd:1024:128:tilt=0:shepp=1 2>&1 | tee -a $R/gpurun_out/r3l.log
done
