#!/bin/bash
# round 3, call R: SQ counters of the flat forward on a DENSE volume at 1024 angles per launch (is its loop LDS-bound?), then the round's profile passes
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r3r_dense_sq -o run -- python3 $R/tools/quick_bench.py fwd:1024:1024:tilt=0 > $R/gpurun_out/r3r_dense.log 2>&1 || { echo "dense sq pass failed"; tail -5 $R/gpurun_out/r3r_dense.log; exit 1; }
cat $R/gpurun_out/r3r_dense.log | tail -2
cd $R
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$R/gpurun_out/r3r_dense_sq/run_counter_collection.csv")))
acc = collections.OrderedDict()
for r in rows:
    if "k_fwd_flat_tab" in r["Kernel_Name"]:
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, ["%.4g" % x for x in v])
PY
bash tools/profile_round.sh round3z 2>&1 | tail -30
