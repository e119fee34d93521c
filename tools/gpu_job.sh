#!/bin/bash
# One parametrised GPU job runner (replaces the per-call scripts of rounds 1-3).
#   usage (on the GPU box, from the repo root):  bash tools/gpu_job.sh <tag> <step> [<step> ...]
# A step is  name:seconds:command...  (the command may contain colons; it runs under `timeout -k 10 <seconds>`); its stdout+stderr go to
# gpurun_out/<tag>_<name>.log and the last lines are echoed.  Steps run in order and the job stops at the first failure or
# timeout (no further GPU step after a killed one).  Example:
#   bash tools/gpu_job.sh r4a "tests:900:python3 -m pytest tests -m gpu -x -q" "bench:400:python3 bench.py"
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
for step in "$@"; do
    name=${step%%:*}; rest=${step#*:}; secs=${rest%%:*}; cmd=${rest#*:}
    log=$R/gpurun_out/${tag}_${name}.log
    echo "== $name (limit ${secs}s): $cmd"
    t0=$(date +%s)
    timeout -k 10 $secs bash -c "$cmd" > $log 2>&1
    rc=$?
    echo "== $name rc=$rc after $(( $(date +%s) - t0 ))s"; tail -n 6 $log
    if [ $rc -ne 0 ]; then echo "stopping after failed step $name"; exit $rc; fi
done
