#!/bin/bash
# round 5: the robust _minDist search with a node's rows in registers -- tests (incl. equality with the LDS form), then timing
set -o pipefail
OUT=gpurun_out/r05_q; mkdir -p $OUT
timeout -k 5 600 python -m pytest tests -m gpu -q -k "robust or spatial or complex or track" > $OUT/tests.log 2>&1; rc=$?
tail -6 $OUT/tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 120 python3 bench.py --mode mindist > $OUT/mindist.json 2> $OUT/mindist.err
python3 -c "
import json
d=json.loads(open('gpurun_out/r05_q/mindist.json').read().strip().splitlines()[-1])
for k,v in d['variants'].items(): print(k, v['ms_per_eval'], v['nodes_per_eval'], v['status_counts'], v['result_checksum'])"
python3 tools/robust_stats_probe.py | tail -4
OBTG_MDR_GENERIC=1 python3 tools/robust_stats_probe.py | tail -4
