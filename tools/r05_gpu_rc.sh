#!/bin/bash
# after a change in the _minDist kernels: first call alone, the campaign (identity with the oracle), the minDist tests, the bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 120 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_min_dist_golden" > gpurun_out/rc_first.log 2>&1 || { tail -30 gpurun_out/rc_first.log; exit 1; }
timeout -k 10 700 python tools/mindist_campaign.py 4000 2>/dev/null | tail -6 > gpurun_out/rc_campaign.txt; echo "campaign rc=$?"; cut -c1-600 gpurun_out/rc_campaign.txt
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "min_dist or minDist or mindist or dist2poly" > gpurun_out/rc_md.log 2>&1 || { tail -40 gpurun_out/rc_md.log; exit 1; }
tail -1 gpurun_out/rc_md.log
timeout -k 10 300 python bench.py --mode mindist --steps 10 --warmup 3 > gpurun_out/rc_bench.log 2>&1 || { tail -20 gpurun_out/rc_bench.log; exit 1; }
python3 -c "
import json
d=json.loads(open('gpurun_out/rc_bench.log').read().strip().splitlines()[-1])
for k in ('reference_algorithm','curve_polygon_reference_algorithm'):
    v=d['variants'][k]; print(k, {x:v[x] for x in ('ms_per_eval','nodes_per_eval','status_counts')})"
