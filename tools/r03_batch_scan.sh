#!/bin/bash
# ms per step of the C3 one-launch sweep against the number of rows: where the workgroup rounds of the chip show
mkdir -p gpurun_out/r03_scan
for B in 256 384 512 640 768 896 1024 1153 1280 1536 2048; do
  python bench.py --no-cpu --steps 200 --warmup 20 --batch $B > gpurun_out/r03_scan/b$B.json 2> gpurun_out/r03_scan/b$B.err || exit 1
  python - <<PY
import json
d = json.load(open("gpurun_out/r03_scan/b$B.json"))
print($B, d["ms_per_step"], d["value"])
PY
done
