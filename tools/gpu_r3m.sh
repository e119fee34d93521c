#!/bin/bash
# round 3, call M: (1) does the ALIGNMENT of a 63-float row atomic matter at the memory side?  (2) the final flat forward without its atomics
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/tools && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/gab gatomic_scope_bench.hip && cd $R || exit 1
timeout -k 10 300 /tmp/gab > $R/gpurun_out/r3m_gatomic.log 2>&1 || { echo "gatomic failed"; tail -5 $R/gpurun_out/r3m_gatomic.log; exit 1; }
grep -E "63 lanes|64 lanes|32 lanes|agent scope" $R/gpurun_out/r3m_gatomic.log
bash tools/gpu_r3l.sh
