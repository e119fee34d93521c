#!/bin/bash
# round 3, GPU call E: the whole GPU suite on the final kernels, then the profile passes behind bench.py's roofline
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout -k 10 600 python3 -m pytest tests -m gpu -q -rA > $R/gpurun_out/r3e_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; grep -E "passed|failed|FAILED|Error" $R/gpurun_out/r3e_pytest.log | tail -5
if [ $rc -ne 0 ]; then tail -60 $R/gpurun_out/r3e_pytest.log; exit $rc; fi
bash tools/profile_round.sh round3z 2>&1 | tail -80
