#!/bin/bash
# round 6, call g: three-instruction exchanges in the row reductions; VALU counters of the mindist kernels
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r06_g; mkdir -p $OUT
timeout -k 10 120 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_min_dist_golden" > $OUT/first.log 2>&1 || { tail -30 $OUT/first.log; exit 1; }
tail -1 $OUT/first.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "min_dist or minDist or mindist or spatial or smoke" > $OUT/md.log 2>&1 || { tail -40 $OUT/md.log; exit 1; }
tail -1 $OUT/md.log
LEGS=reference_algorithm,jacobian_list,curve_polygon_reference_algorithm
timeout -k 10 300 python bench.py --mode mindist --steps 100 --warmup 20 --mindist-legs $LEGS > $OUT/md_default.json 2> $OUT/md_default.err || { tail -20 $OUT/md_default.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_g/md_default.json').read().strip().splitlines()[-1])
for k,v in d['variants'].items():
    print('  ', k, {q:v.get(q) for q in ('ms_per_eval','first_eval_ms','kernel_avg_ms','nodes_per_s','nodes_per_eval')}, (v.get('parity_check') or {}).get('ok'))
PY
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc1 -o run -- python3 bench.py --mode mindist --steps 100 --warmup 20 --no-cpu --mindist-legs $LEGS > $OUT/md_pmc1.json 2> $OUT/md_pmc1.err || { tail -5 $OUT/md_pmc1.err; exit 1; }
python3 tools/pmc_reduce.py $OUT/pmc1 min_dist | grep "INSTS_VALU\|INSTS_SALU\|INSTS_LDS"
timeout -k 10 500 python tools/mindist_campaign.py 1500 > $OUT/campaign.log 2>&1; echo "campaign rc=$?"; tail -2 $OUT/campaign.log | cut -c1-400
