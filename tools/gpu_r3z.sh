#!/bin/bash
# round 3, call Z: 4 against 8 waves per work-group in the gather back-projection, with the live-wave table sharing (SIRT step and 512^3)
R=${GRAFT_REPO_ROOT:-$(pwd)}
A="--steps 3 --warmup 1 --no-align --no-tilted --no-dense --no-cpu-baseline --no-e2e"
for lib in "" $R/build/ab2/libtomo_gw8.so; do
  echo "== library: ${lib:-default (4 waves)}"
  TOMO_AB_LIB=$lib timeout -k 10 300 python3 tools/ab_bench.py $A > $R/gpurun_out/r3z.json 2> $R/gpurun_out/r3z.err || { echo "bench failed"; tail -5 $R/gpurun_out/r3z.err; exit 1; }
  python3 - <<PY
import json
d = json.loads(open("$R/gpurun_out/r3z.json").read().strip().splitlines()[-1])
print(d["value"], {k: round(v["ms_per_step"], 1) for k, v in d["kernels"].items()})
PY
  TOMO_AB_LIB=$lib timeout -k 10 300 python3 tools/quick_bench.py adj:1024:1024:tilt=0 adj:512:512:tilt=0 adj:256:256:tilt=0 2>&1 | tail -3
done
