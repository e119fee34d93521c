#!/bin/bash
# usage (GPU box, repo root): bash tools/ab_sweep.sh "<quick_bench jobs>" lib1.so lib2.so ...   -- the same quick_bench jobs on the shipped library, then on each A/B build
jobs=$1; shift
echo "== shipped"; python3 tools/quick_bench.py $jobs 2>&1 | grep -E "^(fwd|adj|cg|pg|bpv)"
for lib in "$@"; do
    echo "== $lib"; TOMO_AB_LIB=$lib python3 tools/quick_bench.py $jobs 2>&1 | grep -E "^(fwd|adj|cg|pg|bpv)"
done
