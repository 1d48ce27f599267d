#!/bin/bash
set -o pipefail
OUT=gpurun_out/r06_p; mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_gpu_dropin.py -m gpu -x -q -k "spatial or complex or track" > $OUT/t.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/t.log
timeout -k 10 300 python bench.py --mode mindist --steps 100 --warmup 20 --no-cpu --mindist-legs provider_end_to_end > $OUT/md.json 2> $OUT/md.err; echo "rc=$?"
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_p/md.json').read().strip().splitlines()[-1])
for k,v in d['variants'].items(): print(k, {q:v.get(q) for q in ('ms_per_jacobian','first_ms','ms_per_constraint_evaluation','jacobian_shape','nonzeros','finite')})
PY
timeout -k 5 120 python examples/example4_complex_obstacles.py 2>&1 | grep -v amdgpu | tail -2
