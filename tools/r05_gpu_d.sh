#!/bin/bash
# round 5: worker waves per SIMD of k_min_dist_wave against the register budget they need (2 / 3 / 4 waves: 194 / 168+52 B scratch / 128+216 B)
set -o pipefail
OUT=gpurun_out/r05_d; mkdir -p $OUT
P=optimalbeziertrajectorygeneration_amd
run() { # name lib waves
  OBTG_LIB=$2 OBTG_MD_WAVES_PER_SIMD=$3 timeout -k 10 120 python3 bench.py --mode mindist > $OUT/$1.json 2> $OUT/$1.err || tail -3 $OUT/$1.err
}
run base_w2 $P/libobtg_hip.so 2
run mdw3_w3 $P/exp_mdw3.so 3
run mdw4_w4 $P/exp_mdw4.so 4
run mdw4_w3 $P/exp_mdw4.so 3
run mdw3_w2 $P/exp_mdw3.so 2
run base_w2_again $P/libobtg_hip.so 2
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_d/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); v=d["variants"]["reference_algorithm"]
        print(f.split("/")[-1], v["ms_per_eval"], v["first_eval_ms"], v["nodes_per_s"], v["status_counts"], v["result_checksum"])
    except Exception as e: print(f, "failed", e)
PY
