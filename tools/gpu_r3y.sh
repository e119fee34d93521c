#!/bin/bash
# round 3, call Y: the whole GPU suite, then a fuzz soak of the tile kernels (seeds 4 .. 27) on the final kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout -k 10 600 python3 -m pytest tests -m gpu -q -rA > $R/gpurun_out/r3y_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; grep -E "passed|failed|FAILED|Error" $R/gpurun_out/r3y_pytest.log | tail -5
if [ $rc -ne 0 ]; then tail -60 $R/gpurun_out/r3y_pytest.log; exit $rc; fi
timeout -k 10 900 python3 tools/fuzz_soak.py 4 28 > $R/gpurun_out/r3y_soak.log 2>&1
echo "soak rc=$?"; tail -5 $R/gpurun_out/r3y_soak.log
