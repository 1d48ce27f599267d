#!/bin/bash
# quad form of _minDist: the first call alone under a short limit, then the minDist tests, then the bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 120 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_min_dist_golden" > gpurun_out/z_first.log 2>&1 || { tail -30 gpurun_out/z_first.log; exit 1; }
tail -3 gpurun_out/z_first.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "min_dist or minDist or mindist or 3d_sweep_as_one_launch" > gpurun_out/z_md.log 2>&1 || { tail -40 gpurun_out/z_md.log; exit 1; }
tail -3 gpurun_out/z_md.log
timeout -k 10 300 python bench.py --mode mindist --steps 10 --warmup 3 > gpurun_out/z_bench_quad.log 2>&1 || { tail -20 gpurun_out/z_bench_quad.log; exit 1; }
tail -1 gpurun_out/z_bench_quad.log
OBTG_MD_FORM=wave timeout -k 10 300 python bench.py --mode mindist --steps 10 --warmup 3 > gpurun_out/z_bench_wave.log 2>&1 || { tail -20 gpurun_out/z_bench_wave.log; exit 1; }
tail -1 gpurun_out/z_bench_wave.log
