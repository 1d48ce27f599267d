#!/bin/bash
# the structured step at small batches: workgroups of the S (separation streams) and G (gjkNew streams) kinds forced
mkdir -p gpurun_out
out=gpurun_out/r04_struct_small_batch_scan.txt
: > $out
for B in 145 289 577; do
  for cfg in "0 0" "1024 512" "512 384" "384 256" "256 256" "128 128" "64 64"; do
    set -- $cfg
    if [ $1 -eq 0 ]; then unset OBTG_STRUCT_SEP_WGS OBTG_STRUCT_GJK_WGS; else export OBTG_STRUCT_SEP_WGS=$1 OBTG_STRUCT_GJK_WGS=$2; fi
    line=$(timeout -k 10 120 python bench.py --batch $B --steps 200 --warmup 20 --no-cpu --no-proxy 2>/dev/null | tail -1)
    echo "B $B S_wgs $1 G_wgs $2 $(python -c "import json,sys; d=json.loads(sys.argv[1]); v=d['variants']['fd_structured']; print(v.get('ms_per_step'), v.get('spread'))" "$line")" >> $out
  done
done
cat $out
