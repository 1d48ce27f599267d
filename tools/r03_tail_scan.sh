#!/bin/bash
# C3 step time against the tail split of the one-launch sweep's grid (OBTG_TAIL_ROWS rows at OBTG_TAIL_W workgroups per row)
mkdir -p gpurun_out/r03_tail
for cfg in "0 2" "-1 -1" "128 4" "128 8" "192 8" "256 8" "256 16" "320 8" "384 8" "256 4" "512 4"; do
  set -- $cfg
  if [ "$1" = "-1" ]; then unset OBTG_TAIL_ROWS OBTG_TAIL_W; else export OBTG_TAIL_ROWS=$1 OBTG_TAIL_W=$2; fi
  python bench.py --no-cpu --steps 300 --warmup 30 > gpurun_out/r03_tail/t$1_$2.json 2> gpurun_out/r03_tail/t$1_$2.err || exit 1
  python - <<PY
import json
d = json.load(open("gpurun_out/r03_tail/t$1_$2.json"))
print("tail_rows $1 W $2:", d["ms_per_step"], d["roofline"]["frac"], d.get("parity_check"))
PY
done
