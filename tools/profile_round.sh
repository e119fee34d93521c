#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag> [legs]      legs: subset of "H D P G" (default "H D P G")
#   H  the headline workload            bench.py                    key N1024_A1024_G1
#   D  its dense-volume leg             bench.py --dense            key N1024_A1024_G1_D   (Shepp-Logan + 0.05: no zero voxel, nothing skipped)
#   P  its tilted-pose leg              bench.py --perturbed        key N1024_A1024_G1_P   (general tile kernels)
#   G  the alignment-gradient kernels   bench.py --only-align near|dense   keys C5_N512_P720_G1_near / _dense  (config 5: k_cost_grad(v2|v3); + a TA pass)
# 1. rocprofv3 --kernel-trace --stats of the default bench.py run (all legs in one process: the per-dispatch durations of every kernel)
# per leg: 2./3. separate --pmc FETCH_SIZE / WRITE_SIZE passes and 4./5. two --pmc SQ passes of `bench.py --steps 1 --warmup 1 <leg>`
#    (the last dispatch of each kernel is the timed step)
# Raw output under gpurun_out/<tag>_*, summaries into gpurun_out/<tag>_profiles (copy what is judged into profiles/; the two JSON
# files hold all legs, keyed by workload, and the kernel-source hash they were taken on).
# Every rocprofv3 run has the program itself after `--` (no shell / env hop), counters never combined with tracing.
tag=$1
legs=${2:-"H D P G"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
SIDE="--steps 1 --warmup 1 --no-align --no-tilted --no-dense --no-cpu-baseline"
OUT=gpurun_out/${tag}_profiles
cd /tmp && export TMPDIR=/tmp
# counters of an earlier call of the same round (same kernel sources) are carried along: a round's passes do not fit one 20-minute GPU call
mkdir -p $R/$OUT
for f in sq_counters.json pmc_traffic.json; do [ -f $R/profiles/$f ] && cp $R/profiles/$f $R/$OUT/$f; done
if [ -z "$SKIP_STATS" ]; then
    timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -o run -- python3 $R/bench.py --no-cpu-baseline --detail $R/gpurun_out/${tag}_bench_under_rocprof_detail.json > $R/gpurun_out/${tag}_bench_under_rocprof.json 2> $R/gpurun_out/${tag}_stats.err || { echo "stats pass failed"; tail -5 $R/gpurun_out/${tag}_stats.err; exit 1; }
    echo "stats pass done"
fi
SQ1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
SQ2="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_BUSY_CYCLES"
TA="TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"
for leg in $legs; do
    if [ "$leg" = "G" ]; then
        for sub in near dense; do
            t=${tag}_G${sub}
            for pass in fetch write sq1 sq2 ta; do
                case $pass in
                    fetch) ctrs="FETCH_SIZE";; write) ctrs="WRITE_SIZE";; sq1) ctrs="$SQ1";; sq2) ctrs="$SQ2";; ta) ctrs="$TA";;
                esac
                cd /tmp
                timeout -k 10 400 rocprofv3 --pmc $ctrs --output-format csv -d $R/gpurun_out/${t}_${pass} -o run -- python3 $R/bench.py --only-align $sub > $R/gpurun_out/${t}_${pass}.json 2> $R/gpurun_out/${t}_${pass}.err || { echo "gradient $sub: $pass pass failed"; tail -5 $R/gpurun_out/${t}_${pass}.err; exit 1; }
                echo "gradient $sub: $pass pass done"
            done
            cd $R
            python3 tools/summarise_rocprof.py $t gpurun_out/${tag}_stats gpurun_out/${t}_fetch gpurun_out/${t}_write --which last --workload "config 5: 512^3 x 720 poses, alignment gradient, $sub (stats: bench.py default; PMC passes: bench.py --only-align $sub, last dispatch of each kernel)" --key C5_N512_P720_G1_$sub --out $OUT > /dev/null
            python3 tools/summarise_sq.py $t C5_N512_P720_G1_$sub gpurun_out/${t}_sq1 gpurun_out/${t}_sq2 gpurun_out/${t}_ta --workload "bench.py --only-align $sub" --out $OUT > /dev/null
        done
        continue
    fi
    case $leg in
        H) flag=""; sfx=""; what="headline";;
        D) flag="--dense"; sfx="_D"; what="dense volume";;
        P) flag="--perturbed"; sfx="_P"; what="tilted poses";;
        *) echo "unknown leg $leg"; exit 1;;
    esac
    t=${tag}${sfx}
    for pass in fetch write sq1 sq2; do
        case $pass in
            fetch) ctrs="FETCH_SIZE";; write) ctrs="WRITE_SIZE";; sq1) ctrs="$SQ1";; sq2) ctrs="$SQ2";;
        esac
        cd /tmp
        timeout -k 10 600 rocprofv3 --pmc $ctrs --output-format csv -d $R/gpurun_out/${t}_${pass} -o run -- python3 $R/bench.py $SIDE $flag > $R/gpurun_out/${t}_${pass}.json 2> $R/gpurun_out/${t}_${pass}.err || { echo "$what: $pass pass failed"; tail -5 $R/gpurun_out/${t}_${pass}.err; exit 1; }
        echo "$what: $pass pass done"
    done
    cd $R
    python3 tools/summarise_rocprof.py $t gpurun_out/${tag}_stats gpurun_out/${t}_fetch gpurun_out/${t}_write --which last --workload "N=1024 n_proj=1024 n_gpus=1, $what (stats: bench.py default; PMC passes: bench.py $SIDE $flag, last dispatch of each kernel)" --key N1024_A1024_G1$sfx --out $OUT > /dev/null
    python3 tools/summarise_sq.py $t N1024_A1024_G1$sfx gpurun_out/${t}_sq1 gpurun_out/${t}_sq2 --workload "bench.py $SIDE $flag" --out $OUT > /dev/null
done
cd $R
if [ -z "$SKIP_STATS" ]; then
    cp gpurun_out/${tag}_bench_under_rocprof.json $OUT/${tag}_bench_under_rocprof.json
    cp gpurun_out/${tag}_bench_under_rocprof_detail.json $OUT/${tag}_bench_under_rocprof_detail.json
    find gpurun_out/${tag}_stats -name "*kernel_stats.csv" -exec cp {} $OUT/${tag}_kernel_stats.csv \;
fi
echo "summary done: $(ls $OUT)"
