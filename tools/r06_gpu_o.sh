#!/bin/bash
set -o pipefail
OUT=gpurun_out/r06_o; mkdir -p $OUT
timeout -k 10 300 python bench.py --mode mindist --steps 100 --warmup 20 --no-cpu --mindist-legs provider_end_to_end,jacobian_list_robust > $OUT/md.json 2> $OUT/md.err; echo "rc=$?"; tail -3 $OUT/md.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_o/md.json').read().strip().splitlines()[-1])
for k,v in d['variants'].items(): print(k, {q:v.get(q) for q in ('ms_per_eval','kernel_avg_ms','ms_per_jacobian','first_ms','ms_per_constraint_evaluation','jacobian_shape','nonzeros','finite')})
PY
