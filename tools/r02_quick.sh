#!/bin/bash
# quick A/B numbers: C3 default, C3 --separate, C4 (short)
OUT=gpurun_out/${1:-quick}; mkdir -p $OUT
for v in "" "--separate"; do
  timeout -k 10 120 python bench.py --no-cpu $v > $OUT/c3.json 2> $OUT/c3.err || exit 1
  python - "C3 $v" $OUT/c3.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(sys.argv[1], d["value"], d["ms_per_step"], [(k["kernel"], k["avg_ms"], k["frac"]) for k in d["kernels"]])
PY
done
timeout -k 10 120 python bench.py --no-cpu --workload C4 --steps 10 --warmup 3 > $OUT/c4.json 2> $OUT/c4.err || exit 1
python - "C4" $OUT/c4.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(sys.argv[1], d["value"], d["ms_per_step"], [(k["kernel"], k["avg_ms"], k["frac"]) for k in d["kernels"]])
PY
