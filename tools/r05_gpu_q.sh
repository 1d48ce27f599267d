#!/bin/bash
set -o pipefail
OUT=gpurun_out/r05_q; mkdir -p $OUT
timeout -k 5 600 python -m pytest tests -m gpu -q -k "bezier_methods or gjk_pairs_bit_exact or min_dist2poly or mindist_known" > $OUT/tests.log 2>&1; rc=$?
tail -12 $OUT/tests.log
exit $rc
