#!/bin/bash
# the C3 step on the first B rows of the view, workgroups per row forced (OBTG_SWEEP_WGS) -- what a rank of `--mode rows` runs
mkdir -p gpurun_out
out=gpurun_out/r04_small_batch_scan.txt
: > $out
for B in 145 289 577; do
  for W in 0 2 3 4 6 8 12 16; do
    if [ $W -eq 0 ]; then unset OBTG_SWEEP_WGS; else export OBTG_SWEEP_WGS=$W; fi
    line=$(timeout -k 10 120 python bench.py --batch $B --steps 400 --warmup 50 --no-cpu --no-variants 2>/dev/null | tail -1)
    echo "B $B W $W $(python -c "import json,sys; d=json.loads(sys.argv[1]); print(d['ms_per_step'], d['value'], [(k['kernel'],k['avg_ms']) for k in d['kernels']])" "$line")" >> $out
  done
done
cat $out
