#!/bin/bash
# round 5: level-parallel de Casteljau splits in k_min_dist_wave -- the _minDist tests (values, node and call counts) first, then its bench line
set -o pipefail
OUT=gpurun_out/r05_e; mkdir -p $OUT
timeout -k 5 300 python -m pytest tests -m gpu -q -k "min_dist or mindist or spatial or complex" > $OUT/md_tests.log 2>&1; rc=$?
tail -5 $OUT/md_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for i in 1 2; do timeout -k 10 120 python3 bench.py --mode mindist > $OUT/mindist_$i.json 2> $OUT/mindist_$i.err || tail -3 $OUT/mindist_$i.err; done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_e/mindist_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); v=d["variants"]["reference_algorithm"]
    print(f.split("/")[-1], v["ms_per_eval"], v["first_eval_ms"], v["nodes_per_s"], v["status_counts"], v["result_checksum"])
PY
