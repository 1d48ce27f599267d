#!/bin/bash
# (the one-launch DEG_ELEV > 0 step of profiles/r04_experiments/step_elev_one_launch_mfma.patch must be applied: OBTG_STEP_ELEV_PER16 exists only there)
mkdir -p gpurun_out; out=gpurun_out/r04_step_elev_scan.txt; : > $out
for sh in default 4,2,10 6,2,8 7,3,6 8,3,5 3,2,11 5,3,8; do
  if [ $sh = default ]; then unset OBTG_STEP_ELEV_PER16; else export OBTG_STEP_ELEV_PER16=$sh; fi
  for wg in 0 3 6; do
    if [ $wg -eq 0 ]; then unset OBTG_SWEEP_WGS; else export OBTG_SWEEP_WGS=$wg; fi
    line=$(timeout -k 10 120 python bench.py --workload C5 --steps 60 --warmup 10 --no-cpu --no-variants 2>/dev/null | tail -1)
    echo "per16 $sh sweep_wgs $wg $(python -c "import json,sys; d=json.loads(sys.argv[1]); print(d['ms_per_step'], [(k['kernel'],k['avg_ms']) for k in d['kernels']])" "$line")" >> $out
  done
done
cat $out
