#!/bin/bash
# SQ / LDS counters of the default bench step's kernels (each counter group its own run); output under gpurun_out/pmc_ps/
set -e -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/pmc_ps; mkdir -p $OUT
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $OUT/g$i.log 2>&1 || echo "group $i failed" >> $OUT/progress.log
  echo "group $i done" >> $OUT/progress.log
done
