#!/bin/bash
# round 5: _minDist with k_min_dist_quad -- kernel stats and the two counter passes (profiles/r05_mindist_*_quad)
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r05_l2; mkdir -p $OUT
timeout -k 5 90 python -m pytest tests/test_gpu_dropin.py -m gpu -q -k "mindist_known_answers" > $OUT/mindist_first.log 2>&1 || { tail -3 $OUT/mindist_first.log; exit 1; }
timeout -k 10 120 python3 bench.py --mode mindist > $OUT/mindist.json 2> $OUT/mindist.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mindist_stats -o run -- python3 bench.py --mode mindist > $OUT/mindist_stats.json 2> $OUT/mindist_stats.err || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/mindist_pmc -o run -- python3 bench.py --mode mindist --steps 50 --warmup 10 > $OUT/mindist_pmc.json 2> $OUT/mindist_pmc.err || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/mindist_pmc2 -o run -- python3 bench.py --mode mindist --steps 50 --warmup 10 > $OUT/mindist_pmc2.json 2> $OUT/mindist_pmc2.err || exit 1
cut -d, -f1-4 $OUT/mindist_stats/run_kernel_stats.csv | head -4
python3 - <<'PY'
import csv, glob, collections
for d in ("mindist_pmc", "mindist_pmc2"):
    acc = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/r05_l2/%s/*counter_collection.csv" % d):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].split("::")[-1]
            if "min_dist" in k: acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print("%-20s %-22s launches=%d mean=%.5g" % (k, c, len(v), sum(v) / len(v)))
PY
