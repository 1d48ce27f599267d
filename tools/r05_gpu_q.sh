#!/bin/bash
set -o pipefail
OUT=gpurun_out/r05_q; mkdir -p $OUT
timeout -k 5 600 python -m pytest tests -m gpu -q -k "bezier_methods or module_level or closures_match" > $OUT/tests.log 2>&1; rc=$?
tail -12 $OUT/tests.log
exit $rc
