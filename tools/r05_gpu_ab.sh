#!/bin/bash
# A/B on one box, interleaved: the library as built (a**2 as libm's pow) against the a * a variant
mkdir -p gpurun_out/ab
for i in 1 2 3; do
  python bench.py --no-cpu --no-variants --steps 200 --warmup 20 > gpurun_out/ab/pow_$i.json 2>/dev/null
  OBTG_LIB=optimalbeziertrajectorygeneration_amd/exp_sqprod.so python bench.py --no-cpu --no-variants --steps 200 --warmup 20 > gpurun_out/ab/prod_$i.json 2>/dev/null
done
python3 - <<'PY'
import json, glob
for k in ("pow", "prod"):
    v = [json.loads(open(f).read().strip().splitlines()[-1])["ms_per_step"] for f in sorted(glob.glob("gpurun_out/ab/%s_*.json" % k))]
    print(k, v)
PY
