#!/bin/bash
# one bench line per BASELINE.json configuration (1 GPU) + the variants reported beside them, written to gpurun_out/<dir>/
OUT=gpurun_out/${1:-configs}; mkdir -p $OUT
for w in C2 C2_file C3 C4 C5; do
  steps=300; [ $w = C4 ] && steps=10; [ $w = C5 ] && steps=50
  echo "start $w $(date +%T)" >> $OUT/progress.log
  timeout -k 5 240 python bench.py --workload $w --steps $steps --warmup 5 --cpu-seconds 4 > $OUT/$w.json 2> $OUT/$w.err
  echo "$w rc=$? $(date +%T)" >> $OUT/progress.log
done
timeout -k 5 120 python bench.py --mode mindist > $OUT/C5_mindist.json 2> $OUT/C5_mindist.err; echo "mindist rc=$?" >> $OUT/progress.log
timeout -k 5 120 python bench.py --mode pairs --workload C4 --steps 5 --warmup 2 --no-cpu > $OUT/C4_pairs_1rank.json 2> $OUT/C4_pairs.err; echo "pairs rc=$?" >> $OUT/progress.log
timeout -k 5 120 python bench.py --gpus 2 --backend gloo --one-device --mode pairs --workload C4 --steps 5 --warmup 2 --no-cpu > $OUT/C4_pairs_2ranks_one_device.json 2>> $OUT/C4_pairs.err; echo "pairs2 rc=$?" >> $OUT/progress.log
python tools/host_path_probe.py > $OUT/host_path.txt 2>&1; echo "host path rc=$?" >> $OUT/progress.log
cat $OUT/progress.log
