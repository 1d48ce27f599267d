#!/bin/bash
# scheduler-strategy variants of the two kernel files against the library as built, interleaved on one box: the headline step and the _minDist sweep
mkdir -p gpurun_out/sched
for i in 1 2; do
  for v in base ilp bias0; do
    if [ $v = base ]; then unset OBTG_LIB; else export OBTG_LIB=optimalbeziertrajectorygeneration_amd/exp_$v.so; fi
    python bench.py --no-cpu --no-variants --steps 200 --warmup 20 > gpurun_out/sched/c3_${v}_$i.json 2>/dev/null
    python bench.py --mode mindist --steps 10 --warmup 3 > gpurun_out/sched/md_${v}_$i.json 2>/dev/null
  done
done
unset OBTG_LIB
python3 - <<'PY'
import json, glob
for v in ("base", "ilp", "bias0"):
    c3 = [json.loads(open(f).read().strip().splitlines()[-1])["ms_per_step"] for f in sorted(glob.glob("gpurun_out/sched/c3_%s_*.json" % v))]
    md = [json.loads(open(f).read().strip().splitlines()[-1])["variants"]["reference_algorithm"]["ms_per_eval"] for f in sorted(glob.glob("gpurun_out/sched/md_%s_*.json" % v))]
    print(v, "C3 step ms", c3, " _minDist sweep ms", md)
PY
