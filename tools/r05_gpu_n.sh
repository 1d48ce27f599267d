#!/bin/bash
# round 5: small batches -- gjkNew's phase 1 on fewer of a workgroup's waves (OBTG_SWEEP_GJK_WAVES) x workgroups per row (OBTG_SWEEP_WGS)
set -o pipefail
OUT=gpurun_out/r05_n; mkdir -p $OUT
run() { # B W G
  OBTG_SWEEP_WGS=$2 OBTG_SWEEP_GJK_WAVES=$3 timeout -k 5 100 python3 bench.py --batch $1 --steps 200 --warmup 30 --no-cpu --no-variants --no-proxy > $OUT/b$1_w$2_g$3.json 2> $OUT/b$1_w$2_g$3.err
  python3 -c "
import json
d=json.loads(open('$OUT/b$1_w$2_g$3.json').read().strip().splitlines()[-1]); print('B $1 W $2 gjk_waves $3', d['ms_per_step'], [(k['kernel'],k['avg_ms']) for k in d['kernels']])" | tee -a $OUT/summary.txt
}
for B in 145 289; do
  for W in 3 4 6 8; do for G in 4 2 1; do run $B $W $G; done; done
done
run 1153 2 4; run 1153 2 2; run 1153 3 2
