#!/bin/bash
# round 3, call N: what does WRITE_SIZE count for float atomics -- lanes or 64-B sectors?  (the microbenchmark under --pmc WRITE_SIZE)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/tools && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/gab gatomic_scope_bench.hip && cd /tmp && export TMPDIR=/tmp || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r3n_write -o run -- /tmp/gab > $R/gpurun_out/r3n_gab.log 2>&1 || { echo "pmc run failed"; tail -5 $R/gpurun_out/r3n_gab.log; exit 1; }
ls $R/gpurun_out/r3n_write
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$R/gpurun_out/r3n_write/run_counter_collection.csv")))
print(rows[0].keys())
for r in rows:
    print(r.get("Kernel_Name")[:60], r.get("Counter_Name"), r.get("Counter_Value"))
PY
