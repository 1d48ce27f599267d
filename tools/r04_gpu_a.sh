#!/bin/bash
# round 4, first look at the matrix-instruction elevations: GPU suite, then C5 with the separation rows as their own launch
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_multirank.py > gpurun_out/r04a_tests.log 2>&1
rc=$?
tail -15 gpurun_out/r04a_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests timed out"; exit 1; fi
for v in "coop1" "coop0"; do
  nt=1; [ $v = coop0 ] && nt=0
  OBTG_ELEV_COOP=$nt OBTG_SEP_DYN_ELEV=0 timeout -k 10 300 python bench.py --workload C5 --steps 100 --warmup 20 --no-cpu --no-variants > gpurun_out/r04a_c5_sep_$v.json 2> gpurun_out/r04a_c5_sep_$v.err || { echo "bench $v failed"; tail -5 gpurun_out/r04a_c5_sep_$v.err; exit 1; }
  OBTG_ELEV_COOP=$nt timeout -k 10 300 python bench.py --workload C5 --steps 100 --warmup 20 --no-cpu > gpurun_out/r04a_c5_$v.json 2> gpurun_out/r04a_c5_$v.err || { echo "bench $v failed"; tail -5 gpurun_out/r04a_c5_$v.err; exit 1; }
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04a_c5*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'unparsable', e); continue
    print(f, d['value'], d['ms_per_step'], [(k['kernel'], k.get('avg_ms')) for k in d.get('kernels',[])], (d.get('variants') or {}).get('fd_structured'), d.get('parity_check'))
PY
