#!/bin/bash
# round 3, GPU call I: whole GPU suite with the new flat forward as default, a fuzz soak, then the bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout -k 10 600 python3 -m pytest tests -m gpu -q > $R/gpurun_out/r3i_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; grep -E "passed|failed|FAILED|Error" $R/gpurun_out/r3i_pytest.log | tail -5
if [ $rc -ne 0 ]; then tail -60 $R/gpurun_out/r3i_pytest.log; exit $rc; fi
timeout -k 10 400 python3 tools/fuzz_soak.py 4 60 > $R/gpurun_out/r3i_soak.log 2>&1; tail -2 $R/gpurun_out/r3i_soak.log
timeout -k 10 600 python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/r3i_bench_1024.json 2> $R/gpurun_out/r3i_bench_1024.err || { tail -30 $R/gpurun_out/r3i_bench_1024.err; exit 1; }
python3 - <<'PY'
import json
j = json.loads(open("gpurun_out/r3i_bench_1024.json").read().strip().splitlines()[-1])
print("=====", j["value"], "it/s", j["ms_per_step"], "ms/step")
print({k: round(v["ms_per_step"], 2) for k, v in j["kernels"].items()})
for k in ("tilted_poses", "dense_volume"):
    print(k, json.dumps(j.get(k))[:400])
print("e2e", j["align_rigid_e2e"]["wall_s"], j["align_rigid_e2e"]["sirt_kernel_s"])
PY
