#!/bin/bash
# parity of the sweeps + quick numbers
OUT=gpurun_out/${1:-ab}; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "gjk or pair_sweep or c5 or constraint_sweep or fd_forms" > $OUT/pytest.log 2>&1 || { tail -15 $OUT/pytest.log; exit 1; }
tail -1 $OUT/pytest.log
bash tools/r02_quick.sh $1
