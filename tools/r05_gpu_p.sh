#!/bin/bash
# round 5: the one-launch sweep compiled for FIVE waves per SIMD (96 VGPRs) at a rank's share of the rows: more, smaller workgroups in one round
set -o pipefail
OUT=gpurun_out/r05_p; mkdir -p $OUT
P=optimalbeziertrajectorygeneration_amd
run() { # name lib B W
  OBTG_LIB=$2 OBTG_SWEEP_WGS=$4 timeout -k 5 100 python3 bench.py --batch $3 --steps 200 --warmup 30 --no-cpu --no-variants --no-proxy > $OUT/$1.json 2> $OUT/$1.err || tail -2 $OUT/$1.err
  python3 -c "
import json
d=json.loads(open('$OUT/$1.json').read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], [(k['kernel'],k['avg_ms']) for k in d['kernels']], d['parity_check'])" | tee -a $OUT/summary.txt
}
for B in 145 289; do
  run base_b${B}_auto $P/libobtg_hip.so $B 0
  for W in 4 6 7 8 10; do run ps5_b${B}_w$W $P/exp_ps5.so $B $W; done
  run ps5_b${B}_auto $P/exp_ps5.so $B 0
done
run base_b1153 $P/libobtg_hip.so 1153 0
run ps5_b1153 $P/exp_ps5.so 1153 0
