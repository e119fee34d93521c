#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
# 1. rocprofv3 --kernel-trace --stats of the default bench.py run
# 2./3. separate --pmc FETCH_SIZE / WRITE_SIZE passes and 4./5. two --pmc SQ passes of `bench.py --steps 1 --warmup 1` (the last
#    dispatch of each kernel is the timed step)   6. plain bench.py line.
# Raw output under gpurun_out/<tag>_*, summaries into gpurun_out/<tag>_profiles (copy what is judged into profiles/).
# Every rocprofv3 run has the program itself after `--` (no shell / env hop), counters never combined with tracing.
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
SIDE="--steps 1 --warmup 1 --no-align --no-tilted --no-dense --no-cpu-baseline"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -o run -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/${tag}_bench_under_rocprof.json 2> $R/gpurun_out/${tag}_stats.err || { echo "stats pass failed"; tail -5 $R/gpurun_out/${tag}_stats.err; exit 1; }
echo "stats pass done"
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${tag}_fetch -o run -- python3 $R/bench.py $SIDE > $R/gpurun_out/${tag}_fetch.json 2> $R/gpurun_out/${tag}_fetch.err || { echo "fetch pass failed"; exit 1; }
echo "fetch pass done"
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${tag}_write -o run -- python3 $R/bench.py $SIDE > $R/gpurun_out/${tag}_write.json 2> $R/gpurun_out/${tag}_write.err || { echo "write pass failed"; exit 1; }
echo "write pass done"
timeout -k 10 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${tag}_sq1 -o run -- python3 $R/bench.py $SIDE > $R/gpurun_out/${tag}_sq1.json 2> $R/gpurun_out/${tag}_sq1.err || { echo "sq1 pass failed"; tail -5 $R/gpurun_out/${tag}_sq1.err; exit 1; }
echo "sq1 pass done"
timeout -k 10 600 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/${tag}_sq2 -o run -- python3 $R/bench.py $SIDE > $R/gpurun_out/${tag}_sq2.json 2> $R/gpurun_out/${tag}_sq2.err || { echo "sq2 pass failed"; tail -5 $R/gpurun_out/${tag}_sq2.err; exit 1; }
echo "sq2 pass done"
cd $R
python3 tools/summarise_rocprof.py $tag gpurun_out/${tag}_stats gpurun_out/${tag}_fetch gpurun_out/${tag}_write --which last --workload "N=1024 n_proj=1024 n_gpus=1 (bench.py default; PMC passes: bench.py $SIDE, last dispatch of each kernel)" --key N1024_A1024_G1 --out gpurun_out/${tag}_profiles
python3 tools/summarise_sq.py $tag N1024_A1024_G1 gpurun_out/${tag}_sq1 gpurun_out/${tag}_sq2 --workload "bench.py $SIDE" --out gpurun_out/${tag}_profiles > /dev/null
cp gpurun_out/${tag}_bench_under_rocprof.json gpurun_out/${tag}_profiles/${tag}_bench_under_rocprof.json
echo "summary done"
