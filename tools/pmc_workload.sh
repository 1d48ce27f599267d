#!/bin/bash
# SQ / LDS counters of every kernel of a bench workload (each counter group its own run): tools/pmc_workload.sh C5 r02f_pmc
set -e -o pipefail
export TMPDIR=/tmp
W=${1:-C5}; OUT=gpurun_out/${2:-pmc_$W}; EXTRA=${3:-}; mkdir -p $OUT
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -o run -- python3 bench.py --workload $W --steps 3 --warmup 1 --no-cpu --streams 1 $EXTRA > $OUT/g$i.log 2>&1 || echo "group $i failed" >> $OUT/progress.log
  echo "group $i done" >> $OUT/progress.log
done
python3 tools/pmc_reduce.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
