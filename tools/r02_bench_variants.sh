#!/bin/bash
# C3 bench in its three step forms + the C5 line (one gpurun call): tools/r02_bench_variants.sh <outdir>
OUT=gpurun_out/${1:-r02e}; mkdir -p $OUT
for v in "--streams 2" "--streams 1" "--materialise"; do
  tag=$(echo $v | tr -d ' -')
  timeout -k 10 120 python bench.py --no-cpu $v > $OUT/c3_$tag.json 2> $OUT/c3_$tag.err || exit 1
  python - "$v" $OUT/c3_$tag.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(sys.argv[1], d["value"], d["ms_per_step"], [(k["kernel"], k["avg_ms"]) for k in d["kernels"]], d["roofline"]["frac"])
PY
done
timeout -k 10 150 python bench.py --workload C5 --steps 50 --warmup 5 --no-cpu > $OUT/bench_c5.json 2> $OUT/bench_c5.err || exit 1
python - $OUT/bench_c5.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("C5", d["value"], d["ms_per_step"], [(k["kernel"], k["avg_ms"], k["frac"]) for k in d["kernels"]])
PY
