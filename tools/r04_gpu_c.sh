#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_multirank.py > gpurun_out/r04c_tests.log 2>&1
rc=$?
tail -25 gpurun_out/r04c_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests timed out"; exit 1; fi
OBTG_SEP_DYN_ELEV=0 timeout -k 10 300 python bench.py --workload C5 --steps 100 --warmup 20 --no-cpu --no-variants > gpurun_out/r04c_c5_sep.json 2> gpurun_out/r04c_c5_sep.err || { echo "bench sep failed"; tail -5 gpurun_out/r04c_c5_sep.err; }
timeout -k 10 300 python bench.py --workload C5 --steps 100 --warmup 20 > gpurun_out/r04c_c5.json 2> gpurun_out/r04c_c5.err || { echo "bench failed"; tail -5 gpurun_out/r04c_c5.err; }
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04c_c5*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'unparsable', e); continue
    print(f, d['value'], d['ms_per_step'], [(k['kernel'], k.get('avg_ms')) for k in d.get('kernels',[])], (d.get('variants') or {}).get('fd_structured',{}).get('ms_per_step'), d.get('parity_check'))
PY
