#!/bin/bash
# round 6, call d: merged first two gjkNew steps + max-only row reduction (tests, timing); C4's HBM traffic attributed
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r06_d; mkdir -p $OUT
timeout -k 10 120 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_min_dist_golden" > $OUT/first.log 2>&1 || { tail -30 $OUT/first.log; exit 1; }
tail -1 $OUT/first.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "min_dist or minDist or mindist or spatial or smoke or teacher_forced" > $OUT/md.log 2>&1 || { tail -40 $OUT/md.log; exit 1; }
tail -1 $OUT/md.log
timeout -k 10 300 python bench.py --mode mindist --steps 100 --warmup 20 > $OUT/md_default.json 2> $OUT/md_default.err || { tail -20 $OUT/md_default.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_d/md_default.json').read().strip().splitlines()[-1])
for k,v in d['variants'].items():
    print('  ', k, {q:v.get(q) for q in ('ms_per_eval','first_eval_ms','kernel_avg_ms','nodes_per_s','status_counts')}, (v.get('parity_check') or {}).get('ok'))
PY
# C4: where do 80.6 GB of writes for 74.3 GB of results come from?  (a) the one-launch step, (b) the dynamics groups as a launch of
# their own (OBTG_FOLD_DYNAMICS=0), (c) separation rows and the gjkNew sweep as two launches (--separate)
for cfg in "a:" "b:OBTG_FOLD_DYNAMICS=0" "c:--separate"; do
  tag=${cfg%%:*}; what=${cfg#*:}
  envs=""; flags=""
  case "$what" in OBTG_*) envs="$what";; --*) flags="$what";; esac
  for ctr in FETCH_SIZE WRITE_SIZE; do
    ( [ -n "$envs" ] && export $envs; rocprofv3 --pmc $ctr --output-format csv -d $OUT/c4_${tag}_$ctr -o run -- python3 bench.py --workload C4 --steps 3 --warmup 1 --no-cpu --no-variants $flags > $OUT/c4_${tag}_$ctr.json 2> $OUT/c4_${tag}_$ctr.err ) || { tail -5 $OUT/c4_${tag}_$ctr.err; exit 1; }
  done
  echo "== C4 traffic, setting $tag ($what)"
  python3 tools/pmc_reduce.py $OUT/c4_${tag}_FETCH_SIZE | grep -v "fill\|copy\|elementwise" | tail -6
  python3 tools/pmc_reduce.py $OUT/c4_${tag}_WRITE_SIZE | grep -v "fill\|copy\|elementwise" | tail -6
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/suite.log 2>&1; echo "suite rc=$?"; tail -3 $OUT/suite.log
