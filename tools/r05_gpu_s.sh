#!/bin/bash
# round 5: the whole GPU suite + the bench lines of C3 (default), C5, C4 and the mindist mode on the round's final tree
set -o pipefail
OUT=gpurun_out/r05_s; mkdir -p $OUT
timeout -k 5 90 python -m pytest tests/test_gpu_dropin.py -m gpu -q -k "mindist_known_answers" > $OUT/mindist_first.log 2>&1 || { tail -3 $OUT/mindist_first.log; exit 1; }
timeout -k 10 1000 python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1; rc=$?
tail -6 $OUT/gpu_tests.log
echo "pytest rc=$rc"
timeout -k 10 300 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout -k 10 300 python3 bench.py --workload C5 --steps 100 --warmup 10 > $OUT/bench_C5.json 2> $OUT/bench_C5.err
timeout -k 10 300 python3 bench.py --workload C4 --steps 10 --warmup 2 --cpu-seconds 4 > $OUT/bench_C4.json 2> $OUT/bench_C4.err
timeout -k 10 120 python3 bench.py --mode mindist > $OUT/bench_mindist.json 2> $OUT/bench_mindist.err
timeout -k 10 200 python3 bench.py --mode pairs --workload C4 --steps 20 --warmup 5 --no-cpu > $OUT/bench_C4_pairs_B1.json 2> $OUT/bench_C4_pairs_B1.err
timeout -k 10 300 python3 bench.py --mode rows --workload C4 --steps 5 --warmup 2 --no-cpu --no-variants --gather-minima --force-dist --backend nccl > $OUT/bench_C4_rows_sparse_rccl1.json 2> $OUT/bench_C4_rows_sparse_rccl1.err
python3 - <<'PY'
import json
for n in ("default","C5","C4"):
    d=json.loads(open("gpurun_out/r05_s/bench_%s.json"%n).read().strip().splitlines()[-1])
    print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], (d.get("parity_check") or {}).get("ok"), d["cpu_baseline"]["value"] if d.get("cpu_baseline") else None)
d=json.loads(open("gpurun_out/r05_s/bench_mindist.json").read().strip().splitlines()[-1])
print("mindist", {k:(v["ms_per_eval"], v.get("valu_busy")) for k,v in d["variants"].items()})
for n in ("C4_pairs_B1","C4_rows_sparse_rccl1"):
    try:
        d=json.loads(open("gpurun_out/r05_s/bench_%s.json"%n).read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d["config"].get("allgather_bytes"), d["config"].get("gather_check"), d["config"].get("backend"))
    except Exception as e: print(n, "failed", e)
PY
exit $rc
