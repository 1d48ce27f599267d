#!/bin/bash
set -o pipefail
OUT=gpurun_out/r06_m; mkdir -p $OUT
timeout -k 10 120 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_min_dist_golden" > $OUT/first.log 2>&1 || { tail -30 $OUT/first.log; exit 1; }
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "extreme_scales" > $OUT/ext.log 2>&1; echo "extreme-scale test rc=$?"; tail -15 $OUT/ext.log
