#!/bin/bash
set -o pipefail
OUT=gpurun_out/r05_x; mkdir -p $OUT
timeout -k 5 300 python -m pytest tests/test_gpu_dropin.py -m gpu -q -k "latency" > $OUT/t.log 2>&1; rc=$?
tail -5 $OUT/t.log
exit $rc
