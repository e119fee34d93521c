#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
J="cg:512:240:grad_variant=3:tilt=2 cg:512:240:grad_variant=3:tilt=2:shepp=1 cg:512:720:grad_variant=3:tilt=2:shepp=1 cg:512:240:grad_variant=2:tilt=0"
echo "== old lib (commit 8493464)" > $R/gpurun_out/r2d_ab.log
TOMO_HIP_LIB=$R/build/libtomo_hip_r2b.so timeout -k 10 300 python3 tools/quick_bench.py $J >> $R/gpurun_out/r2d_ab.log 2>&1 || exit 1
echo "== new lib" >> $R/gpurun_out/r2d_ab.log
timeout -k 10 300 python3 tools/quick_bench.py $J cg:512:720:grad_variant=5:tilt=2:shepp=1 cg:512:240:grad_variant=5:tilt=2:shepp=1 >> $R/gpurun_out/r2d_ab.log 2>&1
cat $R/gpurun_out/r2d_ab.log
