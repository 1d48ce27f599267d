#!/bin/bash
# round 6, call c: planar curve <-> polygon kernel (worker counts A/B), the W = 2 / 3 choice of k_min_dist_quad, counters of the mindist kernels
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r06_c; mkdir -p $OUT
timeout -k 10 120 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_min_dist_golden" > $OUT/first.log 2>&1 || { tail -30 $OUT/first.log; exit 1; }
tail -1 $OUT/first.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "min_dist or minDist or mindist or spatial or smoke" > $OUT/md.log 2>&1 || { tail -40 $OUT/md.log; exit 1; }
tail -1 $OUT/md.log
summ() { python3 - "$1" <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k,v in d['variants'].items():
    print('  ', k, {q:v.get(q) for q in ('ms_per_eval','first_eval_ms','kernel_avg_ms','nodes_per_s','status_counts','result_checksum')}, (v.get('parity_check') or {}).get('ok'))
PY
}
echo "== default"
timeout -k 10 300 python bench.py --mode mindist --steps 100 --warmup 20 > $OUT/md_default.json 2> $OUT/md_default.err || { tail -20 $OUT/md_default.err; exit 1; }
summ $OUT/md_default.json
for v in m2p2 m2p3; do
  echo "== $v (curve <-> polygon legs)"
  OBTG_LIB=optimalbeziertrajectorygeneration_amd/exp_$v.so timeout -k 10 300 python bench.py --mode mindist --steps 100 --warmup 20 --no-cpu --mindist-legs curve_polygon_reference_algorithm > $OUT/md_$v.json 2> $OUT/md_$v.err || { tail -20 $OUT/md_$v.err; exit 1; }
  summ $OUT/md_$v.json
done
echo "== OBTG_MD_PLANAR=0 (curve <-> polygon legs)"
OBTG_MD_PLANAR=0 timeout -k 10 300 python bench.py --mode mindist --steps 100 --warmup 20 --no-cpu --mindist-legs curve_polygon_reference_algorithm > $OUT/md_2p3d.json 2> $OUT/md_2p3d.err || { tail -20 $OUT/md_2p3d.err; exit 1; }
summ $OUT/md_2p3d.json
# counters: the reference-algorithm kernels
LEGS="reference_algorithm,jacobian_list,curve_polygon_reference_algorithm"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 bench.py --mode mindist --steps 100 --warmup 20 --no-cpu --mindist-legs $LEGS > $OUT/md_stats.json 2> $OUT/md_stats.err || { tail -5 $OUT/md_stats.err; exit 1; }
echo "stats done"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc$i -o run -- python3 bench.py --mode mindist --steps 100 --warmup 20 --no-cpu --mindist-legs $LEGS > $OUT/md_pmc$i.json 2> $OUT/md_pmc$i.err || { tail -5 $OUT/md_pmc$i.err; exit 1; }
  echo "pmc group $i done"
done
python3 tools/pmc_reduce.py $OUT/pmc1 min_dist | tail -20
python3 tools/pmc_reduce.py $OUT/pmc2 min_dist | tail -20
