#!/bin/bash
set -o pipefail
OUT=gpurun_out/r05_m; mkdir -p $OUT
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "deg8 or golden_and_oracle or constraints_golden" > $OUT/deg8_tests.log 2>&1; rc=$?
tail -25 $OUT/deg8_tests.log
exit $rc
