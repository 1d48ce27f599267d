#!/bin/bash
# round 6, final call: mindist tests, every counter on the final tree, then the driver's own command
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r06_l; mkdir -p $OUT
timeout -k 10 120 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_min_dist_golden" > $OUT/first.log 2>&1 || { tail -30 $OUT/first.log; exit 1; }
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "min_dist or minDist or mindist or spatial or planar_builds" > $OUT/md.log 2>&1 || { tail -40 $OUT/md.log; exit 1; }
tail -1 $OUT/md.log
bash tools/r06_collect_counters.sh r06_counters > $OUT/collect.log 2>&1; echo "collect rc=$?"; tail -2 $OUT/collect.log
timeout -k 10 300 python bench.py --mode mindist --steps 100 --warmup 20 > $OUT/mindist.json 2> $OUT/mindist.err; echo "mindist rc=$?"
t0=$(date +%s)
timeout -k 10 580 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; rc=$?
echo "bench rc=$rc in $(( $(date +%s) - t0 )) s"
