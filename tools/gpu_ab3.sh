#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
J="$1"; shift
rm -f $R/gpurun_out/ab3.log
for lib in "$@"; do
  echo "== $lib" >> $R/gpurun_out/ab3.log
  TOMO_HIP_LIB=$R/build/$lib timeout -k 10 300 python3 tools/quick_bench.py $J >> $R/gpurun_out/ab3.log 2>&1 || { cat $R/gpurun_out/ab3.log; exit 1; }
done
cat $R/gpurun_out/ab3.log
