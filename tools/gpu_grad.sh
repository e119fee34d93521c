#!/bin/bash
# usage: bash tools/gpu_grad.sh <tag> -- GPU tests of the gradient / ray-driven kernels, then their timings at 512^3
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py -m gpu -q -rA > $R/gpurun_out/${tag}_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; grep -E "passed|failed|FAILED" $R/gpurun_out/${tag}_pytest.log | tail -20
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 600 python3 tools/quick_bench.py cg:512:240:grad_variant=2:tilt=0 cg:512:240:grad_variant=3:tilt=0 \
  cg:512:240:grad_variant=2:tilt=1 cg:512:240:grad_variant=3:tilt=1 cg:512:240:grad_variant=2:tilt=2 cg:512:240:grad_variant=3:tilt=2 cg:512:240:grad_variant=3:tilt=3 \
  cg:512:720:grad_variant=4:tilt=2:shepp=1 cg:512:720:grad_variant=4:tilt=0:shepp=1 fwd:512:64:fwd_variant=2 > $R/gpurun_out/${tag}_quick.log 2>&1
echo "quick rc=$?"; cat $R/gpurun_out/${tag}_quick.log
