#!/bin/bash
# k_min_dist2poly_quad: the first call alone under a short limit, then the 2poly tests, then the bench's curve-polygon lines (quad, wave)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 120 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_min_dist2poly_golden" > gpurun_out/2p_first.log 2>&1 || { tail -30 gpurun_out/2p_first.log; exit 1; }
tail -1 gpurun_out/2p_first.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "min_dist or minDist or mindist or dist2poly" > gpurun_out/2p_md.log 2>&1 || { tail -40 gpurun_out/2p_md.log; exit 1; }
tail -1 gpurun_out/2p_md.log
for f in quad wave; do
  [ $f = wave ] && export OBTG_MD_FORM=wave
  timeout -k 10 300 python bench.py --mode mindist --steps 10 --warmup 3 > gpurun_out/2p_bench_$f.log 2>&1 || { tail -20 gpurun_out/2p_bench_$f.log; exit 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/2p_bench_$f.log').read().strip().splitlines()[-1])
for k in ('reference_algorithm','curve_polygon_reference_algorithm'):
    v=d['variants'][k]; print('$f', k, {x:v[x] for x in ('ms_per_eval','nodes_per_eval','result_checksum','status_counts')})"
done
