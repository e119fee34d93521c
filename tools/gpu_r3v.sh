#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout -k 10 90 python3 tools/quick_bench.py adj:128:8:tilt=0 > $R/gpurun_out/r3v_smoke.log 2>&1 || { echo "smoke failed"; tail -5 $R/gpurun_out/r3v_smoke.log; exit 1; }
timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -m gpu -q > $R/gpurun_out/r3v_pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; grep -E "passed|failed|FAILED|Error" $R/gpurun_out/r3v_pytest.log | tail -5
if [ $rc -ne 0 ]; then tail -40 $R/gpurun_out/r3v_pytest.log; exit $rc; fi
timeout -k 10 300 python3 tools/quick_bench.py adj:1024:1024:tilt=0 adj:1024:128:tilt=0 2>&1 | tee $R/gpurun_out/r3v_time.log
