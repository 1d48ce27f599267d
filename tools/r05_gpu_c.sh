#!/bin/bash
# round 5, third GPU call: the GPU suite again (fixed tests) + the _minDist kernel as queue-fed worker waves
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r05_c; mkdir -p $OUT
# the queue-fed _minDist kernel first, alone and under a short limit (its first build hung: see the kernel's comment)
timeout -k 5 90 python -m pytest tests/test_gpu_dropin.py -m gpu -q -k "mindist_known_answers" > $OUT/mindist_first.log 2>&1; rc0=$?
tail -3 $OUT/mindist_first.log
if [ $rc0 -ne 0 ]; then echo "minDist test failed or timed out (rc=$rc0): stopping here"; exit $rc0; fi
timeout -k 10 1000 python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1; rc=$?
tail -25 $OUT/gpu_tests.log
echo "pytest rc=$rc"
for w in 1 2 3; do
  OBTG_MD_WAVES_PER_SIMD=$w timeout -k 10 200 python3 bench.py --mode mindist > $OUT/mindist_w$w.json 2> $OUT/mindist_w$w.err || tail -3 $OUT/mindist_w$w.err
done
OBTG_MD_HISTORY=0 timeout -k 10 200 python3 bench.py --mode mindist > $OUT/mindist_nohist.json 2> $OUT/mindist_nohist.err
python3 - <<'PY'
import json
for n in ("w1","w2","w3","nohist"):
    try:
        d=json.loads(open("gpurun_out/r05_c/mindist_%s.json"%n).read().strip().splitlines()[-1])
        v=d["variants"]["reference_algorithm"]
        print(n, v["ms_per_eval"], v["first_eval_ms"], v["nodes_per_s"], v["gjk_calls_per_s"], v["status_counts"], v["result_checksum"])
    except Exception as e: print(n, "failed", e)
PY
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mindist_stats -o run -- python3 bench.py --mode mindist > $OUT/mindist_stats.json 2> $OUT/mindist_stats.err
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/mindist_pmc -o run -- python3 bench.py --mode mindist --steps 50 --warmup 10 > $OUT/mindist_pmc.json 2> $OUT/mindist_pmc.err
for ex in "example7_dubins_degree8.py time_optimal" "example7_dubins_degree8.py example2 10" "example8_driving_on_a_track.py --raw"; do
  timeout -k 10 120 python examples/$ex > "$OUT/$(echo $ex | tr ' ./' '___').log" 2>&1; tail -4 "$OUT/$(echo $ex | tr ' ./' '___').log"
done
exit $rc
