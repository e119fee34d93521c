#!/bin/bash
# round 3, GPU call H: the LDS-table + image-pair flat forward (option fwd_flat_tab) against the round-2 kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout -k 10 90 python3 tools/quick_bench.py fwd:128:8:tilt=0:fwd_flat_tab=1 > $R/gpurun_out/r3h_smoke.log 2>&1 || { echo "smoke failed/timed out"; tail -5 $R/gpurun_out/r3h_smoke.log; exit 1; }
cat $R/gpurun_out/r3h_smoke.log
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "flat" > $R/gpurun_out/r3h_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; grep -E "passed|failed|FAILED|Error" $R/gpurun_out/r3h_pytest.log | tail -5
if [ $rc -ne 0 ]; then tail -40 $R/gpurun_out/r3h_pytest.log; exit $rc; fi
timeout -k 10 300 python3 tools/quick_bench.py fwd:1024:128:tilt=0:fwd_flat_tab=0 fwd:1024:128:tilt=0:fwd_flat_tab=1 fwd:1024:128:tilt=0:fwd_flat_tab=0:shepp=1 fwd:1024:128:tilt=0:fwd_flat_tab=1:shepp=1 fwd:512:128:tilt=0:fwd_flat_tab=0 fwd:512:128:tilt=0:fwd_flat_tab=1 fwd:1024:1024:tilt=0 fwd:1024:1024:tilt=0:shift=1 fwd:1024:1024:tilt=0:shepp=2 2>&1 | tee $R/gpurun_out/r3h_time.log
