#!/bin/bash
# round 3, GPU call A: full GPU test suite on the round's first changes; micro-benchmarks (masked LDS atomics, float-atomic
# scopes); counters of the CURRENT gradient kernels on config 5 (dense 512^3, +-2 deg); the sharded path on a 1-rank communicator.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
timeout -k 10 600 python3 -m pytest tests -m gpu -q -rA > $R/gpurun_out/r3a_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; grep -E "passed|failed|FAILED|Error" $R/gpurun_out/r3a_pytest.log | tail -8
if [ $rc -ne 0 ]; then tail -40 $R/gpurun_out/r3a_pytest.log; exit $rc; fi
timeout -k 10 120 $R/build/tools/lds_mask_bench > $R/gpurun_out/r3a_lds_mask.log 2>&1 && cat $R/gpurun_out/r3a_lds_mask.log || exit 1
timeout -k 10 120 $R/build/tools/gatomic_scope_bench > $R/gpurun_out/r3a_gatomic.log 2>&1 && cat $R/gpurun_out/r3a_gatomic.log || exit 1
timeout -k 10 300 python3 $R/bench.py --size 512 --angles 256 --steps 3 --warmup 1 --force-sharded --no-align --no-dense --no-cpu-baseline > $R/gpurun_out/r3a_bench_sharded512.json 2> $R/gpurun_out/r3a_bench_sharded512.err || { tail -20 $R/gpurun_out/r3a_bench_sharded512.err; exit 1; }
timeout -k 10 300 python3 $R/bench.py --size 512 --angles 256 --steps 3 --warmup 1 --no-align --no-dense --no-cpu-baseline > $R/gpurun_out/r3a_bench_plain512.json 2> $R/gpurun_out/r3a_bench_plain512.err || exit 1
python3 - <<'PY'
import json
for t in ("sharded512", "plain512"):
    j = json.loads(open("gpurun_out/r3a_bench_%s.json" % t).read().strip().splitlines()[-1])
    print(t, j["value"], "it/s", {k: (round(v["ms_per_step"], 2), v["launches_per_step"]) for k, v in j["kernels"].items()}, "tilted", j.get("tilted_poses", {}).get("value"))
PY
bash $R/tools/pmc_passes.sh r3a_pmc_grad3 cg:512:240:grad_variant=3:tilt=2 || exit 1
bash $R/tools/pmc_passes.sh r3a_pmc_grad2 cg:512:240:grad_variant=2:tilt=2 || exit 1
cd $R
for v in 3 2; do
  echo "== k_proj_grad_v$v (fused) dense 512^3, 240 poses +-2 deg"
  python3 tools/pmc_table.py "k_proj_grad_v$v" gpurun_out/r3a_pmc_grad$v/p* > gpurun_out/r3a_pmc_grad$v.txt
  cat gpurun_out/r3a_pmc_grad$v.txt; grep "ms " gpurun_out/r3a_pmc_grad$v.p1.log
done
