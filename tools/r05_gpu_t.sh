#!/bin/bash
set -o pipefail
OUT=gpurun_out/r05_t; mkdir -p $OUT
timeout -k 5 900 python -m pytest tests/test_gpu_multirank.py -m gpu -q -k "rows_mode or rccl_path or collective_behind" > $OUT/tests.log 2>&1; rc=$?
tail -6 $OUT/tests.log
timeout -k 10 300 python3 bench.py --mode rows --workload C4 --steps 5 --warmup 2 --no-cpu --no-variants --gather-minima --force-dist --backend nccl > $OUT/bench_C4_rows_sparse_rccl1.json 2> $OUT/bench_C4_rows_sparse_rccl1.err
timeout -k 10 300 python3 bench.py --mode rows --workload C4 --steps 5 --warmup 2 --no-cpu --no-variants > $OUT/bench_C4_rows_plain.json 2> $OUT/bench_C4_rows_plain.err
timeout -k 10 300 python3 bench.py --mode rows --workload C3 --steps 200 --warmup 20 --no-cpu --no-variants --gather-minima --force-dist --backend nccl > $OUT/bench_C3_rows_sparse_rccl1.json 2> $OUT/bench_C3_rows_sparse_rccl1.err
timeout -k 10 300 python3 bench.py --mode rows --workload C3 --steps 200 --warmup 20 --no-cpu --no-variants > $OUT/bench_C3_rows_plain.json 2> $OUT/bench_C3_rows_plain.err
python3 - <<'PY'
import json
for n in ("C4_rows_sparse_rccl1","C4_rows_plain","C3_rows_sparse_rccl1","C3_rows_plain"):
    d=json.loads(open("gpurun_out/r05_t/bench_%s.json"%n).read().strip().splitlines()[-1])
    print(n, d["value"], d["ms_per_step"], d["config"].get("allgather_bytes"), d["config"].get("gather_check"), d["config"].get("backend"))
PY
exit $rc
