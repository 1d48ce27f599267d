#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "row_ranges or set_stream" > gpurun_out/r04d_t1.log 2>&1; tail -12 gpurun_out/r04d_t1.log
timeout -k 10 900 python -m pytest tests/test_gpu_multirank.py -m gpu -q -x > gpurun_out/r04d_t2.log 2>&1; tail -15 gpurun_out/r04d_t2.log
timeout -k 10 400 python bench.py --steps 20 --warmup 10 > gpurun_out/r04d_bench20.json 2> gpurun_out/r04d_bench20.err || tail -5 gpurun_out/r04d_bench20.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04d_bench20.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
for k,v in d['variants'].items(): print(k, {a:b for a,b in v.items() if a not in ('what',)})
for e in d['strong_scaling_proxy']['rows']: print(e)
PY
