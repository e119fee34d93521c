#!/bin/bash
# round 3, GPU call B: the new tests, then bench.py as the driver runs it (all side measurements) and at config 2's size
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_fuzz.py tests/test_gpu_solvers.py tests/test_gpu_regularized.py -m gpu -q -rA > $R/gpurun_out/r3b_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; grep -E "passed|failed|FAILED|Error|alignment API" $R/gpurun_out/r3b_pytest.log | tail -8
if [ $rc -ne 0 ]; then tail -60 $R/gpurun_out/r3b_pytest.log; exit $rc; fi
ulimit -a | grep -i stack
t0=$(date +%s)
timeout -k 10 900 python3 $R/bench.py > $R/gpurun_out/r3b_bench_1024.json 2> $R/gpurun_out/r3b_bench_1024.err || { tail -30 $R/gpurun_out/r3b_bench_1024.err; exit 1; }
echo "bench default: $(( $(date +%s) - t0 )) s"
timeout -k 10 600 python3 $R/bench.py --size 256 --angles 256 --steps 5 --warmup 1 > $R/gpurun_out/r3b_bench_256.json 2> $R/gpurun_out/r3b_bench_256.err || { tail -30 $R/gpurun_out/r3b_bench_256.err; exit 1; }
python3 - <<'PY'
import json
for t in ("1024", "256"):
    j = json.loads(open("gpurun_out/r3b_bench_%s.json" % t).read().strip().splitlines()[-1])
    print("=====", t, j["value"], "it/s", j["ms_per_step"], "ms/step")
    print({k: round(v["ms_per_step"], 2) for k, v in j["kernels"].items()})
    r = j["roofline"]
    print("roofline:", {k: r.get(k) for k in ("kernel", "bound", "frac", "useful_flop_frac", "atomics_frac", "stale_counters_refused")})
    for k in ("tilted_poses", "dense_volume", "alignment_gradient", "align_rigid_e2e", "cpu_baseline"):
        print(k, json.dumps(j.get(k))[:1800])
PY
