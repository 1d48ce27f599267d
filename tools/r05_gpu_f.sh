#!/bin/bash
# round 5: separationRows='active' (tests + the swarm example), workgroup timelines of the C3 step at a rank's share of the rows
set -o pipefail
OUT=gpurun_out/r05_f; mkdir -p $OUT
timeout -k 5 400 python -m pytest tests/test_gpu_dropin.py -m gpu -q -k "active_separation or reduced_separation" > $OUT/active_tests.log 2>&1; rc=$?
tail -30 $OUT/active_tests.log
timeout -k 5 300 python examples/example2_swarm_3d.py 5 > $OUT/example2_5.log 2>&1; cat $OUT/example2_5.log
timeout -k 5 300 python examples/example2_swarm_3d.py 8 > $OUT/example2_8.log 2>&1; cat $OUT/example2_8.log
for B in 145 289 1153; do
  OBTG_TIMELINE=$OUT/tl_b$B.txt timeout -k 5 120 python3 bench.py --batch $B --steps 30 --warmup 5 --no-cpu --no-variants --no-proxy > $OUT/bench_b$B.json 2> $OUT/bench_b$B.err
  python3 tools/timeline_report.py $OUT/tl_b$B.txt 2.5 > $OUT/tl_b${B}_report.txt 2>&1
  head -6 $OUT/tl_b${B}_report.txt
done
exit $rc
