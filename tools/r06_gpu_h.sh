#!/bin/bash
# round 6, call h: C4 / C3 with coarser history bins (OBTG_HIST_SHIFT), interleaved with baselines; then the whole suite
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r06_h; mkdir -p $OUT
one() {   # workload, steps, shift
  OBTG_HIST_SHIFT=$3 timeout -k 10 200 python bench.py --workload $1 --steps $2 --warmup 3 --no-cpu --no-variants --no-configs > $OUT/$1_s$3_$4.json 2> $OUT/$1_s$3_$4.err || { tail -5 $OUT/$1_s$3_$4.err; return 1; }
  python3 -c "
import json,sys
d=json.loads(open('$OUT/$1_s$3_$4.json').read().strip().splitlines()[-1])
k=[k for k in d['kernels'] if k['kernel']=='pair_sweep'][0]
print('$1 shift $3 run $4: ms/step %.4f kernel %.5f ms' % (d['ms_per_step'], k['avg_ms']))"
}
for rep in 1 2; do
  for sh in 0 1 2 3; do one C4 12 $sh $rep || exit 1; done
done
for rep in 1 2; do
  for sh in 0 1 2; do one C3 400 $sh $rep || exit 1; done
done
# LDS conflict counters at C4 for shift 0 and 2
for sh in 0 2; do
  OBTG_HIST_SHIFT=$sh rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_c4_s$sh -o run -- python3 bench.py --workload C4 --steps 3 --warmup 1 --no-cpu --no-variants --no-configs > $OUT/pmc_c4_s$sh.json 2> $OUT/pmc_c4_s$sh.err
  echo "== C4 counters, shift $sh"; python3 tools/pmc_reduce.py $OUT/pmc_c4_s$sh k_pair_sweep_tiled
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/suite.log 2>&1; echo "suite rc=$?"; tail -3 $OUT/suite.log
