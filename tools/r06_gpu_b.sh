#!/bin/bash
# round 6, call b: the planar per-row gjkNew machine in k_min_dist_quad -- tests first, then A/B of the forms and of the worker count
set -o pipefail
OUT=gpurun_out/r06_b; mkdir -p $OUT
timeout -k 10 120 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_min_dist_golden" > $OUT/first.log 2>&1 || { tail -30 $OUT/first.log; exit 1; }
tail -1 $OUT/first.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "min_dist or minDist or mindist or spatial" > $OUT/md.log 2>&1 || { tail -40 $OUT/md.log; exit 1; }
tail -1 $OUT/md.log
summ() { python3 - "$1" <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ('reference_algorithm','jacobian_list'):
    v=d['variants'][k]
    print('  ', k, {q:v.get(q) for q in ('ms_per_eval','first_eval_ms','kernel_avg_ms','nodes_per_s','status_counts','result_checksum')}, (v.get('parity_check') or {}).get('ok'))
PY
}
echo "== default (planar machine, 3 workers per SIMD)"
timeout -k 10 300 python bench.py --mode mindist --steps 100 --warmup 20 > $OUT/md_default.json 2> $OUT/md_default.err || { tail -20 $OUT/md_default.err; exit 1; }
summ $OUT/md_default.json
echo "== OBTG_MD_PLANAR=0 (3-D machine, 2 workers per SIMD)"
OBTG_MD_PLANAR=0 timeout -k 10 300 python bench.py --mode mindist --steps 100 --warmup 20 --no-cpu > $OUT/md_3d.json 2> $OUT/md_3d.err || { tail -20 $OUT/md_3d.err; exit 1; }
summ $OUT/md_3d.json
for v in mdp2 mdp4; do
  echo "== $v"
  OBTG_LIB=optimalbeziertrajectorygeneration_amd/exp_$v.so timeout -k 10 300 python bench.py --mode mindist --steps 100 --warmup 20 --no-cpu > $OUT/md_$v.json 2> $OUT/md_$v.err || { tail -20 $OUT/md_$v.err; exit 1; }
  summ $OUT/md_$v.json
done
echo "== default with OBTG_MD_WAVES_PER_SIMD=2"
OBTG_MD_WAVES_PER_SIMD=2 timeout -k 10 300 python bench.py --mode mindist --steps 100 --warmup 20 --no-cpu > $OUT/md_w2.json 2> $OUT/md_w2.err || { tail -20 $OUT/md_w2.err; exit 1; }
summ $OUT/md_w2.json
timeout -k 10 300 python -m pytest tests/test_gpu_dropin.py -m gpu -x -q -k "any_degree_kernels" > $OUT/fdserve.log 2>&1; echo "fd-serving test rc=$?"; tail -5 $OUT/fdserve.log
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python3 - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r06_b/bench.json').read().strip().splitlines()[-1])
    print('value', d['value'], 'ms', d['ms_per_step'], 'roofline', {k:d['roofline'][k] for k in ('bound','frac','traffic')})
    for k,v in (d.get('configs') or {}).items():
        print(k, json.dumps(v)[:600])
except Exception as e:
    print('no bench line', e)
PY
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/suite.log 2>&1; echo "suite rc=$?"; tail -3 $OUT/suite.log
