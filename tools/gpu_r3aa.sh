#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout -k 10 600 python3 -m pytest tests/test_gpu_solvers.py tests/test_gpu_dist.py tests/test_gpu_configs.py -m gpu -q > $R/gpurun_out/r3aa_pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; grep -E "passed|failed|FAILED|Error" $R/gpurun_out/r3aa_pytest.log | tail -5
if [ $rc -ne 0 ]; then tail -40 $R/gpurun_out/r3aa_pytest.log; exit $rc; fi
A="--steps 3 --warmup 1 --no-align --no-tilted --no-dense --no-cpu-baseline --no-e2e"
for extra in "" "--force-sharded"; do
timeout -k 10 300 python3 bench.py $A $extra > $R/gpurun_out/r3aa_bench.json 2> $R/gpurun_out/r3aa_bench.err || { echo "bench failed"; tail -5 $R/gpurun_out/r3aa_bench.err; exit 1; }
python3 - <<PY
import json
d = json.loads(open("$R/gpurun_out/r3aa_bench.json").read().strip().splitlines()[-1])
print("$extra", d["value"], {k: round(v["ms_per_step"], 1) for k, v in d["kernels"].items()}, d["config"].get("rms_error_last"))
PY
done
