#!/bin/bash
# round 3, GPU call D: the 32 x 16-footprint flat forward (option fwd_flat_wide) against the default two-z-tile kernel:
# parity test, ms per angle on a dense volume and on the phantom, and the bytes of float atomics (rocprofv3 --pmc WRITE_SIZE)
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "wide or flat_tile" > $R/gpurun_out/r3d_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; grep -E "passed|failed|FAILED|Error" $R/gpurun_out/r3d_pytest.log | tail -5
if [ $rc -ne 0 ]; then tail -40 $R/gpurun_out/r3d_pytest.log; exit $rc; fi
JOBS="fwd:1024:128:tilt=0:fwd_flat_wide=0 fwd:1024:128:tilt=0:fwd_flat_wide=1 fwd:1024:128:tilt=0:fwd_flat_wide=0:shepp=1 fwd:1024:128:tilt=0:fwd_flat_wide=1:shepp=1"
timeout -k 10 300 python3 tools/quick_bench.py $JOBS 2>&1 | tee $R/gpurun_out/r3d_time.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r3d_write -o run -- python3 $R/tools/quick_bench.py $JOBS > $R/gpurun_out/r3d_write.log 2>&1 || { echo "write pass failed"; tail -5 $R/gpurun_out/r3d_write.log; exit 1; }
cd $R
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob("gpurun_out/r3d_write/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "WRITE_SIZE" and "k_fwd_flat_z" in r["Kernel_Name"]:
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0], float(r["Counter_Value"])))
agg = {}
for d, k, v in rows:
    agg[(d, k)] = agg.get((d, k), 0.0) + v
for (d, k), v in sorted(agg.items()):
    print("dispatch %4d %-40s WRITE_SIZE %.4g KB = %.1f GB per 128 angles -> %.4g KB per 1024 angles" % (d, k, v, v * 1024 / 1e9, v * 8))
PY
