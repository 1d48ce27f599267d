#!/bin/bash
# round 6, call k: the new planar-vs-3-D test, the suite, smoke, then every counter again on this tree
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r06_k; mkdir -p $OUT
timeout -k 10 120 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_min_dist_golden" > $OUT/first.log 2>&1 || { tail -30 $OUT/first.log; exit 1; }
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "planar_builds" > $OUT/planar.log 2>&1; echo "planar test rc=$?"; tail -3 $OUT/planar.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/suite.log 2>&1; echo "suite rc=$?"; tail -3 $OUT/suite.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
bash tools/r06_collect_counters.sh r06_counters2 > $OUT/collect.log 2>&1; echo "collect rc=$?"; tail -3 $OUT/collect.log
