#!/bin/bash
# SQ / LDS counters of the structured step's kernel (tools/timeline_structured_run.py <workload>), each counter group its own run:
#   tools/pmc_structured.sh C5 [outdir]
set -e -o pipefail
export TMPDIR=/tmp
WL=${1:-C3}; OUT=${2:-gpurun_out/pmc_struct_$WL}; mkdir -p $OUT
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -o run -- python3 tools/timeline_structured_run.py $WL > $OUT/g$i.log 2>&1 || echo "group $i failed" >> $OUT/progress.log
  echo "group $i done" >> $OUT/progress.log
done
python3 tools/pmc_reduce.py $OUT k_step_fd_structured > $OUT/summary.txt
