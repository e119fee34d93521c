#!/bin/bash
# SQ counters of the two flat forward kernels on a dense 1024^3 volume, 64 angles
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_WR"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/r3j_pmc/p$i -o run -- python3 $R/tools/quick_bench.py fwd:1024:64:tilt=0:fwd_flat_tab=0 fwd:1024:64:tilt=0:fwd_flat_tab=1 > $R/gpurun_out/r3j.p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $R/gpurun_out/r3j.p$i.log; exit 1; }
done
cd $R
cat gpurun_out/r3j.p1.log
echo "== k_fwd_flat_z<2,16> (round 2)"; python3 tools/pmc_table.py "k_fwd_flat_z" gpurun_out/r3j_pmc/p1 gpurun_out/r3j_pmc/p2
echo "== k_fwd_flat_tab (round 3)"; python3 tools/pmc_table.py "k_fwd_flat_tab" gpurun_out/r3j_pmc/p1 gpurun_out/r3j_pmc/p2
