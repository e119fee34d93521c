#!/bin/bash
# usage: tools/pmc_passes.sh <outdir under gpurun_out> <quick_bench job> -- runs one rocprofv3 --pmc pass per counter group
out=$1; shift
job="$@"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/$out/p$i -o run -- python3 $R/tools/quick_bench.py $job > $R/gpurun_out/$out.p$i.log 2>&1 || { echo "pass $i failed: $grp"; tail -3 $R/gpurun_out/$out.p$i.log; exit 1; }
done <<'GRP'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD
SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
TCP_GATE_EN1_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
TD_TD_BUSY_sum TD_TC_STALL_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
GRBM_GUI_ACTIVE GRBM_COUNT
GRP
echo done $i passes
