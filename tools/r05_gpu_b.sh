#!/bin/bash
# round 5, second GPU call: the whole GPU suite on the tree with 9 control points, the full-size fixtures and the driver flows
set -o pipefail
mkdir -p gpurun_out/r05_b
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r05_b/gpu_tests.log 2>&1; rc=$?
tail -60 gpurun_out/r05_b/gpu_tests.log
echo "pytest rc=$rc"
[ $rc -eq 0 ] && timeout -k 10 120 python examples/example7_dubins_degree8.py time_optimal > gpurun_out/r05_b/ex7_tt.log 2>&1 && tail -8 gpurun_out/r05_b/ex7_tt.log
[ $rc -eq 0 ] && timeout -k 10 120 python examples/example7_dubins_degree8.py example2 10 > gpurun_out/r05_b/ex7_e2.log 2>&1 && tail -5 gpurun_out/r05_b/ex7_e2.log
[ $rc -eq 0 ] && timeout -k 10 120 python examples/example8_driving_on_a_track.py > gpurun_out/r05_b/ex8.log 2>&1 && tail -6 gpurun_out/r05_b/ex8.log
exit $rc
