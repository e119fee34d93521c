#!/bin/bash
# round 3, call S: live-tile lists in both tile forwards -- parity + fuzz tests, then timings on dense / slab / phantom volumes
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout -k 10 90 python3 tools/quick_bench.py fwd:128:8:tilt=0 fwd:128:8:tilt=1 > $R/gpurun_out/r3s_smoke.log 2>&1 || { echo "smoke failed/timed out"; tail -5 $R/gpurun_out/r3s_smoke.log; exit 1; }
cat $R/gpurun_out/r3s_smoke.log
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q > $R/gpurun_out/r3s_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; grep -E "passed|failed|FAILED|Error" $R/gpurun_out/r3s_pytest.log | tail -5
if [ $rc -ne 0 ]; then tail -40 $R/gpurun_out/r3s_pytest.log; exit $rc; fi
timeout -k 10 400 python3 tools/quick_bench.py fwd:1024:128:tilt=1 fwd:1024:128:tilt=1:shepp=2 fwd:1024:128:tilt=1:shepp=1 fwd:1024:1024:tilt=0 fwd:1024:1024:tilt=0:shepp=2 2>&1 | tee $R/gpurun_out/r3s_time.log
