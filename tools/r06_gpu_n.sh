#!/bin/bash
set -o pipefail
OUT=gpurun_out/r06_n; mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_gpu_dropin.py -m gpu -x -q -k "private_separation or served_from_one_batch or example1 or teacher" > $OUT/t.log 2>&1; echo "tests rc=$?"; tail -6 $OUT/t.log
for v in 1 0; do echo "== OBTG_FD_BATCHING=$v"; OBTG_FD_BATCHING=$v timeout -k 5 120 python examples/example1_dubins_time_optimal.py 2>&1 | grep -v amdgpu.ids | grep "DEG_ELEV"; done
