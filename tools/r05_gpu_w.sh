#!/bin/bash
set -o pipefail
OUT=gpurun_out/r05_w; mkdir -p $OUT
timeout -k 5 300 python -m pytest tests -m gpu -q -k "min_dist or mindist or spatial or complex" > $OUT/md_tests.log 2>&1; rc=$?
tail -4 $OUT/md_tests.log
[ $rc -ne 0 ] && exit $rc
OBTG_LIB=optimalbeziertrajectorygeneration_amd/exp_mdtm.so python tools/mindist_phase_probe.py | tail -2
for i in 1 2; do timeout -k 10 120 python3 bench.py --mode mindist > $OUT/mindist_$i.json 2> $OUT/mindist_$i.err; done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_w/mindist_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], {k:(v["ms_per_eval"], v["nodes_per_eval"], v["result_checksum"]) for k,v in d["variants"].items()})
PY
