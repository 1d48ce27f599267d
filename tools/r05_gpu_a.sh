#!/bin/bash
# round 5, first GPU call: measurement only, on the tree as round 4 left it
#  (a) _minDist kernels: kernel stats + one SQ counter pass
#  (b) C4 brute force: SQ counters + HBM traffic of k_pair_sweep_tiled<16> (item 6)
#  (c) the default bench line
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r05_a; mkdir -p $OUT
timeout -k 10 300 python3 bench.py --steps 20 --warmup 10 > $OUT/bench20.json 2> $OUT/bench20.err || tail -5 $OUT/bench20.err
echo "bench done" >> $OUT/progress.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mindist_stats -o run -- python3 bench.py --mode mindist > $OUT/mindist_stats.json 2> $OUT/mindist_stats.err
echo "mindist stats done" >> $OUT/progress.log
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/mindist_pmc -o run -- python3 bench.py --mode mindist --steps 50 --warmup 10 > $OUT/mindist_pmc.json 2> $OUT/mindist_pmc.err
echo "mindist pmc done" >> $OUT/progress.log
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/mindist_pmc2 -o run -- python3 bench.py --mode mindist --steps 50 --warmup 10 > $OUT/mindist_pmc2.json 2> $OUT/mindist_pmc2.err
echo "mindist pmc2 done" >> $OUT/progress.log
timeout -k 10 600 bash tools/pmc_workload.sh C4 r05_a/pmc_C4 "--no-variants --no-proxy" > $OUT/pmc_C4.log 2>&1
echo "C4 pmc done" >> $OUT/progress.log
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_c4 -o run -- python3 bench.py --workload C4 --steps 3 --warmup 1 --no-cpu --no-variants --no-proxy > $OUT/fetch_c4.json 2> $OUT/fetch_c4.err
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_c4 -o run -- python3 bench.py --workload C4 --steps 3 --warmup 1 --no-cpu --no-variants --no-proxy > $OUT/write_c4.json 2> $OUT/write_c4.err
echo "C4 traffic done" >> $OUT/progress.log
cat $OUT/progress.log
find $OUT -name "*.csv" -size +20M -delete
du -sh $OUT
