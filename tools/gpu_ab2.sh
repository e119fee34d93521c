#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
J="fwd:1024:64:tilt=1 adj:1024:64:tilt=1"
for lib in "$@"; do
  echo "== $lib" >> $R/gpurun_out/ab2.log
  TOMO_HIP_LIB=$R/build/$lib timeout -k 10 300 python3 tools/quick_bench.py $J >> $R/gpurun_out/ab2.log 2>&1 || exit 1
done
cat $R/gpurun_out/ab2.log
