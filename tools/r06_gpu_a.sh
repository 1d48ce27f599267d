#!/bin/bash
# round 6, first GPU call: the _minDist family at SLSQP-iteration scale (jacobian_list), the one-line bench with `configs`, the suite
set -o pipefail
OUT=gpurun_out/r06_a; mkdir -p $OUT
timeout -k 10 120 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_min_dist_golden" > $OUT/first.log 2>&1 || { tail -30 $OUT/first.log; exit 1; }
tail -1 $OUT/first.log
timeout -k 10 400 python bench.py --mode mindist --steps 100 --warmup 20 > $OUT/mindist.json 2> $OUT/mindist.err || { tail -20 $OUT/mindist.err; tail -c 2000 $OUT/mindist.json; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_a/mindist.json').read().strip().splitlines()[-1])
for k,v in d['variants'].items():
    print(k, {q:v.get(q) for q in ('ms_per_eval','first_eval_ms','kernel_avg_ms','pairs','nodes_per_s','gjk_calls_per_s','status_counts')})
    if v.get('parity_check'): print('   parity', v['parity_check'])
    if v.get('cpu_baseline'): print('   cpu', v['cpu_baseline'])
PY
timeout -k 10 300 python -m pytest tests/test_gpu_dropin.py -m gpu -x -q -k "any_degree_kernels" > $OUT/fdserve.log 2>&1; echo "fd-serving test rc=$?"; tail -5 $OUT/fdserve.log
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python3 - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r06_a/bench.json').read().strip().splitlines()[-1])
    print('value', d['value'], 'ms', d['ms_per_step'], 'roofline', {k:d['roofline'][k] for k in ('bound','frac','traffic')})
    for k,v in (d.get('configs') or {}).items():
        print(k, json.dumps(v)[:700])
except Exception as e:
    print('no bench line', e)
PY
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/suite.log 2>&1; echo "suite rc=$?"; tail -3 $OUT/suite.log
