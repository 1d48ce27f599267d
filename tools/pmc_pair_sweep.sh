#!/bin/bash
# SQ / LDS counters of the default bench step's kernels (each counter group its own run); output under gpurun_out/pmc_ps/
# extra environment for the runs: PMC_ENV="VAR=val VAR=val"; extra bench flags: BENCH_ARGS="--workload C4 --no-variants"
set -e -o pipefail
export TMPDIR=/tmp
OUT=${1:-gpurun_out/pmc_ps}; mkdir -p $OUT
if [ -n "$PMC_ENV" ]; then export $PMC_ENV; fi
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu $BENCH_ARGS > $OUT/g$i.log 2>&1 || echo "group $i failed" >> $OUT/progress.log
  echo "group $i done" >> $OUT/progress.log
done
