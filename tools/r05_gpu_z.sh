#!/bin/bash
# quad form of _minDist: the first call alone under a short limit, then the minDist tests, the phase probe, the bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 120 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_min_dist_golden" > gpurun_out/z_first.log 2>&1 || { tail -30 gpurun_out/z_first.log; exit 1; }
tail -1 gpurun_out/z_first.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "min_dist or minDist or mindist or 3d_sweep_as_one_launch" > gpurun_out/z_md.log 2>&1 || { tail -40 gpurun_out/z_md.log; exit 1; }
tail -1 gpurun_out/z_md.log
true
true
timeout -k 10 300 python bench.py --mode mindist --steps 10 --warmup 3 > gpurun_out/z_bench_quad.log 2>&1 || { tail -20 gpurun_out/z_bench_quad.log; exit 1; }
python3 -c "
import json
d=json.loads(open('gpurun_out/z_bench_quad.log').read().strip().splitlines()[-1])
v=d['variants']['reference_algorithm']; print({k:v[k] for k in ('ms_per_eval','first_eval_ms','nodes_per_eval','result_checksum','status_counts')})"
