#!/bin/bash
# round 3, call O: the SIRT step with and without the flat forward's atomics (what any work on the atomics can gain at most)
R=${GRAFT_REPO_ROOT:-$(pwd)}
A="--steps 3 --warmup 1 --no-align --no-tilted --no-dense --no-cpu-baseline --no-e2e"
timeout -k 10 300 python3 tools/ab_bench.py $A > $R/gpurun_out/r3o_default.json 2> $R/gpurun_out/r3o_default.err || { echo "default failed"; tail -5 $R/gpurun_out/r3o_default.err; exit 1; }
TOMO_AB_LIB=$R/build/ab2/libtomo_noatomic1.so timeout -k 10 300 python3 tools/ab_bench.py $A > $R/gpurun_out/r3o_noatomic.json 2> $R/gpurun_out/r3o_noatomic.err || { echo "noatomic failed"; tail -5 $R/gpurun_out/r3o_noatomic.err; exit 1; }
python3 - <<PY
import json
for n in ("default", "noatomic"):
    d = json.loads(open("$R/gpurun_out/r3o_%s.json" % n).read().strip().splitlines()[-1])
    print(n, d["value"], {k: round(v["ms_per_step"], 1) for k, v in d["kernels"].items()})
PY
