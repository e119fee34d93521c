#!/bin/bash
# round 3, call P: the flat forward with / without its atomics at 1024 angles per launch (the SIRT step's regime: a 4.3 GB sinogram)
R=${GRAFT_REPO_ROOT:-$(pwd)}
rm -f $R/gpurun_out/r3p.log
for lib in "" "$R/build/ab2/libtomo_noatomic1.so"; do
  echo "== library: ${lib:-default}" | tee -a $R/gpurun_out/r3p.log
  TOMO_AB_LIB=$lib timeout -k 10 300 python3 tools/quick_bench.py fwd:1024:1024:tilt=0 fwd:1024:512:tilt=0 fwd:1024:256:tilt=0 fwd:1024:128:tilt=0 2>&1 | tee -a $R/gpurun_out/r3p.log
done
