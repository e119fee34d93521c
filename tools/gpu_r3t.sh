#!/bin/bash
# round 3, call T: the sharded solver's pipelined iteration on ONE rank (1-rank RCCL communicator) by number of x slabs
R=${GRAFT_REPO_ROOT:-$(pwd)}
A="--force-sharded --no-align --no-tilted --no-dense --no-cpu-baseline --no-e2e"
for n in 8 6 4; do
  timeout -k 10 300 python3 bench.py $A --slabs $n > $R/gpurun_out/r3t_$n.json 2> $R/gpurun_out/r3t_$n.err || { echo "slabs=$n failed"; tail -5 $R/gpurun_out/r3t_$n.err; exit 1; }
  python3 - <<PY
import json
d = json.loads(open("$R/gpurun_out/r3t_$n.json").read().strip().splitlines()[-1])
print("slabs=$n", d["value"], "it/s", {k: round(v["ms_per_step"], 1) for k, v in d["kernels"].items()})
PY
done
