#!/bin/bash
# round 5: _minDist kernels on the final tree -- kernel stats and the two counter passes (profiles/r05_mindist_*_after)
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r05_l; mkdir -p $OUT
timeout -k 5 90 python -m pytest tests/test_gpu_dropin.py -m gpu -q -k "mindist_known_answers" > $OUT/mindist_first.log 2>&1 || { tail -3 $OUT/mindist_first.log; exit 1; }
timeout -k 10 120 python3 bench.py --mode mindist > $OUT/mindist.json 2> $OUT/mindist.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mindist_stats -o run -- python3 bench.py --mode mindist > $OUT/mindist_stats.json 2> $OUT/mindist_stats.err
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/mindist_pmc -o run -- python3 bench.py --mode mindist --steps 50 --warmup 10 > $OUT/mindist_pmc.json 2> $OUT/mindist_pmc.err
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/mindist_pmc2 -o run -- python3 bench.py --mode mindist --steps 50 --warmup 10 > $OUT/mindist_pmc2.json 2> $OUT/mindist_pmc2.err
cat $OUT/mindist.json | head -c 1500; echo
cut -d, -f1-4 $OUT/mindist_stats/run_kernel_stats.csv | head -4
