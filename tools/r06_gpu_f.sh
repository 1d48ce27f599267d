#!/bin/bash
# round 6, call f: split-parameter quotients from hoisted reciprocals (A/B against the divisions as written), 4 workers per SIMD
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r06_f; mkdir -p $OUT
timeout -k 10 120 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_min_dist_golden" > $OUT/first.log 2>&1 || { tail -30 $OUT/first.log; exit 1; }
tail -1 $OUT/first.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "min_dist or minDist or mindist or spatial or smoke" > $OUT/md.log 2>&1 || { tail -40 $OUT/md.log; exit 1; }
tail -1 $OUT/md.log
summ() { python3 - "$1" <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k,v in d['variants'].items():
    print('  ', k, {q:v.get(q) for q in ('ms_per_eval','first_eval_ms','kernel_avg_ms','nodes_per_s','result_checksum')}, (v.get('parity_check') or {}).get('ok'))
PY
}
LEGS=reference_algorithm,jacobian_list,curve_polygon_reference_algorithm
for rep in 1 2; do
echo "== default, run $rep"
timeout -k 10 300 python bench.py --mode mindist --steps 100 --warmup 20 --mindist-legs $LEGS > $OUT/md_default$rep.json 2> $OUT/md_default$rep.err || { tail -20 $OUT/md_default$rep.err; exit 1; }
summ $OUT/md_default$rep.json
for v in pdiv mdp3; do
  echo "== $v, run $rep"
  OBTG_LIB=optimalbeziertrajectorygeneration_amd/exp_$v.so timeout -k 10 300 python bench.py --mode mindist --steps 100 --warmup 20 --no-cpu --mindist-legs $LEGS > $OUT/md_$v$rep.json 2> $OUT/md_$v$rep.err || { tail -20 $OUT/md_$v$rep.err; exit 1; }
  summ $OUT/md_$v$rep.json
done
done
# a campaign beyond the suite's slice: random curve sets, both dimensions, against the oracle (identity)
timeout -k 10 500 python tools/mindist_campaign.py 3000 > $OUT/campaign.log 2>&1; echo "campaign rc=$?"; tail -4 $OUT/campaign.log
timeout -k 10 300 python tools/mindist2poly_campaign.py 1500 > $OUT/campaign2.log 2>&1; echo "campaign2 rc=$?"; tail -4 $OUT/campaign2.log
