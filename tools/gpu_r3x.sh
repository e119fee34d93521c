#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout -k 10 90 python3 tools/quick_bench.py adj:128:8:tilt=1 > $R/gpurun_out/r3x_smoke.log 2>&1 || { echo "smoke failed"; tail -5 $R/gpurun_out/r3x_smoke.log; exit 1; }
timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q > $R/gpurun_out/r3x_pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; grep -E "passed|failed|FAILED|Error" $R/gpurun_out/r3x_pytest.log | tail -5
if [ $rc -ne 0 ]; then tail -40 $R/gpurun_out/r3x_pytest.log; exit $rc; fi
timeout -k 10 300 python3 tools/quick_bench.py adj:1024:64:tilt=1 2>&1 | tee $R/gpurun_out/r3x_time.log
A="--steps 2 --warmup 1 --no-align --no-dense --no-cpu-baseline --no-e2e"
timeout -k 10 400 python3 bench.py $A > $R/gpurun_out/r3x_bench.json 2> $R/gpurun_out/r3x_bench.err || { echo "bench failed"; tail -5 $R/gpurun_out/r3x_bench.err; exit 1; }
python3 - <<PY
import json
d = json.loads(open("$R/gpurun_out/r3x_bench.json").read().strip().splitlines()[-1])
print(d["value"], {k: round(v["ms_per_step"], 1) for k, v in d["kernels"].items()})
print(d["tilted_poses"])
PY
