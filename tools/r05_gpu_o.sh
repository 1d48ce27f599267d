#!/bin/bash
# round 5: VALU instruction counts of the C3 step at B = 145 / 1153, with all four or two of a workgroup's waves on gjkNew
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r05_o; mkdir -p $OUT
run() { # name B G
  OBTG_SWEEP_GJK_WAVES=$3 timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/$1 -o run -- python3 bench.py --batch $2 --steps 5 --warmup 2 --no-cpu --no-variants --no-proxy > $OUT/$1.json 2> $OUT/$1.err
}
run b145_g4 145 4
run b145_g2 145 2
run b145_g1 145 1
run b1153_g4 1153 4
run b1153_g2 1153 2
python3 - <<'PY'
import csv, collections, glob
for d in ("b145_g4","b145_g2","b145_g1","b1153_g4","b1153_g2"):
    acc = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/r05_o/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_pair_sweep" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(d, {k: "%.4g" % (sum(v)/len(v)) for k, v in sorted(acc.items())}, "launches", len(acc.get("SQ_WAVES", [])))
PY
