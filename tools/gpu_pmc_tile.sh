#!/bin/bash
# SQ counters of the general (tilted) tile kernels on 1024^3 x 64 perturbed angles (dense volume)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_WR"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_tile/p$i -o run -- python3 $R/tools/quick_bench.py fwd:1024:64:tilt=1 adj:1024:64:tilt=1 > $R/gpurun_out/pmc_tile.p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $R/gpurun_out/pmc_tile.p$i.log; exit 1; }
done
cd $R
cat gpurun_out/pmc_tile.p1.log
python3 tools/pmc_table.py "k_tile<true>" gpurun_out/pmc_tile/p1 gpurun_out/pmc_tile/p2
python3 tools/pmc_table.py "k_tile<false>" gpurun_out/pmc_tile/p1 gpurun_out/pmc_tile/p2
