#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_solvers.py -m gpu -q -k "flat or sirt or pipelined" > $R/gpurun_out/r3k_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; grep -E "passed|failed|FAILED|Error" $R/gpurun_out/r3k_pytest.log | tail -5
if [ $rc -ne 0 ]; then tail -40 $R/gpurun_out/r3k_pytest.log; exit $rc; fi
timeout -k 10 400 python3 $R/bench.py --no-cpu-baseline --no-align --no-tilted > $R/gpurun_out/r3k_bench.json 2> $R/gpurun_out/r3k_bench.err || { tail -30 $R/gpurun_out/r3k_bench.err; exit 1; }
python3 - <<'PY'
import json
j = json.loads(open("gpurun_out/r3k_bench.json").read().strip().splitlines()[-1])
print("=====", j["value"], "it/s", j["ms_per_step"], "ms/step", {k: round(v["ms_per_step"], 2) for k, v in j["kernels"].items()})
print("dense", json.dumps(j.get("dense_volume"))[:300])
PY
