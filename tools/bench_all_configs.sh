#!/bin/bash
# one bench line per BASELINE.json configuration (1 GPU), written to gpurun_out/configs/
mkdir -p gpurun_out/configs
for w in C2 C3 C4 C5; do
  steps=300; [ $w = C4 ] && steps=10; [ $w = C5 ] && steps=50
  echo "start $w $(date +%T)" >> gpurun_out/configs/progress.log
  timeout -k 5 240 python bench.py --workload $w --steps $steps --warmup 5 --cpu-seconds 4 > gpurun_out/configs/$w.json 2> gpurun_out/configs/$w.err
  echo "$w rc=$? $(date +%T)" >> gpurun_out/configs/progress.log
done
cat gpurun_out/configs/progress.log
