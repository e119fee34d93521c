#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/gpu_check.sh <tag> [pytest args...]
# 1. issue micro-benchmark (if built)  2. the GPU test suite  3. the default bench.py line -- each step only if the previous one
# ended normally (a killed / aborted GPU step is never followed by another one).  Logs under gpurun_out/<tag>_*.
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
if [ -x $R/build/tools/issue_bench ]; then
  timeout -k 10 300 $R/build/tools/issue_bench > $R/gpurun_out/${tag}_issue_bench.log 2>&1 || { echo "issue_bench failed rc=$?"; exit 1; }
  echo "issue_bench done"
fi
timeout -k 10 1500 python3 -m pytest tests -m gpu -q -rA "$@" > $R/gpurun_out/${tag}_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -5 $R/gpurun_out/${tag}_pytest.log
if [ $rc -gt 1 ]; then echo "pytest did not end normally: stopping"; exit $rc; fi
timeout -k 10 900 python3 bench.py > $R/gpurun_out/${tag}_bench.json 2> $R/gpurun_out/${tag}_bench.err
brc=$?
echo "bench rc=$brc"; head -c 3000 $R/gpurun_out/${tag}_bench.json; tail -3 $R/gpurun_out/${tag}_bench.err
exit $(( rc > brc ? rc : brc ))
