#!/bin/bash
# interleaved A/B runs of bench.py variants on one box: tools/r03_ab.sh OUTDIR "ENV1" "ENV2" ...   (each ENV: VAR=val,VAR=val or "-")
OUT=$1; shift; mkdir -p $OUT
for rep in 1 2 3; do
  i=0
  for e in "$@"; do
    i=$((i+1))
    ( if [ "$e" != "-" ]; then export $(echo $e | tr ',' ' '); fi
      python bench.py --no-cpu --steps 300 --warmup 30 > $OUT/v${i}_r$rep.json 2> $OUT/v${i}_r$rep.err ) || exit 1
    python - <<PY
import json
d = json.load(open("$OUT/v${i}_r$rep.json"))
print("rep $rep [$e]:", d["ms_per_step"], [ (k["kernel"], k["avg_ms"]) for k in d["kernels"]])
PY
  done
done
