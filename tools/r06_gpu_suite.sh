#!/bin/bash
set -o pipefail
OUT=gpurun_out/r06_suite; mkdir -p $OUT
timeout -k 10 120 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_min_dist_golden" > $OUT/first.log 2>&1 || { tail -30 $OUT/first.log; exit 1; }
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/suite.log 2>&1; echo "suite rc=$?"; tail -3 $OUT/suite.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
