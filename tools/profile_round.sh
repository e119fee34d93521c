#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
# 1. rocprofv3 --kernel-trace --stats of the default bench.py run  2./3. separate --pmc FETCH_SIZE / WRITE_SIZE passes of a
# one-step run  4. plain bench.py line.  Raw output under gpurun_out/<tag>_*, summaries into profiles/ (tools/summarise_rocprof.py).
set -e
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -o run -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/${tag}_bench_under_rocprof.json 2> $R/gpurun_out/${tag}_stats.err
echo "stats pass done"
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${tag}_fetch -o run -- python3 $R/bench.py --steps 1 --warmup 0 --no-align --no-tilted --no-cpu-baseline > $R/gpurun_out/${tag}_fetch.json 2> $R/gpurun_out/${tag}_fetch.err
echo "fetch pass done"
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${tag}_write -o run -- python3 $R/bench.py --steps 1 --warmup 0 --no-align --no-tilted --no-cpu-baseline > $R/gpurun_out/${tag}_write.json 2> $R/gpurun_out/${tag}_write.err
echo "write pass done"
cd $R
python3 tools/summarise_rocprof.py $tag gpurun_out/${tag}_stats gpurun_out/${tag}_fetch gpurun_out/${tag}_write --workload "N=1024 n_proj=1024 n_gpus=1 (bench.py default; PMC passes with --steps 1 --warmup 0 --no-align)" --key N1024_A1024_G1 --out gpurun_out/${tag}_profiles
cp gpurun_out/${tag}_bench_under_rocprof.json gpurun_out/${tag}_profiles/${tag}_bench_under_rocprof.json
echo "summary done"
