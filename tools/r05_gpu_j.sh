#!/bin/bash
# round 5: C4 brute force (k_pair_sweep_tiled<16>) A/B: tile height, refill threshold, history order off; interleaved with the baseline
set -o pipefail
OUT=gpurun_out/r05_j; mkdir -p $OUT
run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 200 python3 bench.py --workload C4 --steps 12 --warmup 3 --no-cpu --no-variants --no-proxy > $OUT/$name.json 2> $OUT/$name.err || tail -2 $OUT/$name.err
  python3 -c "
import json,sys
d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'))" | tee -a $OUT/summary.txt
}
run base_1 OBTG_NOP=1
run tile13 OBTG_TILE_A=13
run tile21 OBTG_TILE_A=21
run tile25 OBTG_TILE_A=25
run base_2 OBTG_NOP=1
run refill16 OBTG_REFILL_MIN=16
run refill48 OBTG_REFILL_MIN=48
run refill64 OBTG_REFILL_MIN=64
run base_3 OBTG_NOP=1
timeout -k 10 300 python3 bench.py --workload C4 --steps 12 --warmup 3 --no-cpu --no-proxy > $OUT/with_variants.json 2> $OUT/with_variants.err
python3 -c "
import json
d=json.loads(open('$OUT/with_variants.json').read().strip().splitlines()[-1])
print('with variants', d['ms_per_step'])
for k,v in (d.get('variants') or {}).items(): print('  ',k, v.get('ms_per_step'), v.get('spread'))" | tee -a $OUT/summary.txt
