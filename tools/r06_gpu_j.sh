#!/bin/bash
# round 6, call j: campaigns beyond the suite on the final tree (new seeds): sweeps against the oracle, the _minDist kernels, structured step
set -o pipefail
OUT=gpurun_out/r06_j; mkdir -p $OUT
timeout -k 10 120 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_min_dist_golden" > $OUT/first.log 2>&1 || { tail -30 $OUT/first.log; exit 1; }
timeout -k 10 700 python tools/stress_sweeps.py 100 606 > $OUT/planar.log 2>&1; echo "planar sweeps rc=$?"; tail -2 $OUT/planar.log | cut -c1-300
timeout -k 10 500 python tools/stress_sweeps.py gjk3d 300 607 > $OUT/gjk3d.log 2>&1; echo "3-D sweeps rc=$?"; tail -2 $OUT/gjk3d.log | cut -c1-300
timeout -k 10 500 python tools/stress_sweeps.py families 150 608 > $OUT/families.log 2>&1; echo "families rc=$?"; tail -2 $OUT/families.log | cut -c1-300
timeout -k 10 900 python tools/mindist_campaign.py 10000 > $OUT/campaign.log 2>&1; echo "mindist campaign rc=$?"; tail -1 $OUT/campaign.log | cut -c1-700
timeout -k 10 600 python tools/mindist2poly_campaign.py 6000 > $OUT/campaign2.log 2>&1; echo "mindist2poly campaign rc=$?"; tail -1 $OUT/campaign2.log | cut -c1-500
timeout -k 10 600 python tools/stress_structured.py 60 > $OUT/structured.log 2>&1; echo "structured rc=$?"; tail -2 $OUT/structured.log | cut -c1-300
