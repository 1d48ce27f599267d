#!/bin/bash
# round 6, call e: the quad kernels built per control-point count (no spills at K = 11), planar closest points; then the suite
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r06_e; mkdir -p $OUT
timeout -k 10 120 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_min_dist_golden" > $OUT/first.log 2>&1 || { tail -30 $OUT/first.log; exit 1; }
tail -1 $OUT/first.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "min_dist or minDist or mindist or spatial or smoke" > $OUT/md.log 2>&1 || { tail -40 $OUT/md.log; exit 1; }
tail -1 $OUT/md.log
summ() { python3 - "$1" <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k,v in d['variants'].items():
    print('  ', k, {q:v.get(q) for q in ('ms_per_eval','first_eval_ms','kernel_avg_ms','nodes_per_s','status_counts')}, (v.get('parity_check') or {}).get('ok'))
PY
}
timeout -k 10 300 python bench.py --mode mindist --steps 100 --warmup 20 > $OUT/md_default.json 2> $OUT/md_default.err || { tail -20 $OUT/md_default.err; exit 1; }
summ $OUT/md_default.json
echo "== 4 workers per SIMD (116 B of scratch)"
OBTG_LIB=optimalbeziertrajectorygeneration_amd/exp_mdp4.so timeout -k 10 300 python bench.py --mode mindist --steps 100 --warmup 20 --no-cpu --mindist-legs reference_algorithm,jacobian_list > $OUT/md_mdp4.json 2> $OUT/md_mdp4.err || { tail -20 $OUT/md_mdp4.err; exit 1; }
summ $OUT/md_mdp4.json
echo "== OBTG_MD_MANY=1 (3 workers per SIMD for the 4560-pair call too)"
OBTG_MD_MANY=1 timeout -k 10 300 python bench.py --mode mindist --steps 100 --warmup 20 --no-cpu --mindist-legs reference_algorithm > $OUT/md_many1.json 2> $OUT/md_many1.err || { tail -20 $OUT/md_many1.err; exit 1; }
summ $OUT/md_many1.json
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/suite.log 2>&1; echo "suite rc=$?"; tail -3 $OUT/suite.log
