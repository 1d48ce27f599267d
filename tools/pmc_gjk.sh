#!/bin/bash
# PMC passes over the GJK probe (each counter group in its own run); output under gpurun_out/pmc_gjk/
set -e -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/pmc_gjk; mkdir -p $OUT
i=0
for grp in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -o run -- python3 tools/gjk_probe.py C3 4 > $OUT/g$i.log 2>&1 || echo "group $i failed" >> $OUT/progress.log
  echo "group $i done" >> $OUT/progress.log
done
