#!/bin/bash
set -o pipefail
OUT=gpurun_out/r05_i; mkdir -p $OUT
timeout -k 5 300 python -m pytest tests/test_gpu_multirank.py -m gpu -q -k "collective_behind" > $OUT/comm_test.log 2>&1; rc=$?
tail -25 $OUT/comm_test.log
timeout -k 5 400 python -m pytest tests/test_gpu_dropin.py -m gpu -q -k "active_separation" > $OUT/active_test.log 2>&1; rc2=$?
tail -8 $OUT/active_test.log
exit $((rc + rc2))
