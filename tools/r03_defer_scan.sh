#!/bin/bash
# C3 step time against the separation work left to the grid's tail (OBTG_TS_DEFER_ROWS rows keep OBTG_TS_DEFER_KEEP groups per wave)
mkdir -p gpurun_out/r03_defer
for cfg in "0 2" "-1 -1" "1024 0" "1024 1" "1024 2" "1024 3" "896 1" "896 2" "1153 1" "1153 2" "768 1" "512 0"; do
  set -- $cfg
  if [ "$1" = "-1" ]; then unset OBTG_TS_DEFER_ROWS OBTG_TS_DEFER_KEEP; else export OBTG_TS_DEFER_ROWS=$1 OBTG_TS_DEFER_KEEP=$2; fi
  python bench.py --no-cpu --steps 300 --warmup 30 > gpurun_out/r03_defer/t$1_$2.json 2> gpurun_out/r03_defer/t$1_$2.err || exit 1
  python - <<PY
import json
d = json.load(open("gpurun_out/r03_defer/t$1_$2.json"))
print("defer_rows $1 keep $2:", d["ms_per_step"], d["roofline"]["frac"])
PY
done
