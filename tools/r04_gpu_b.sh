#!/bin/bash
mkdir -p gpurun_out
for v in 0 1 2; do
  OBTG_ELEV_DBG=$v OBTG_SEP_DYN_ELEV=0 timeout -k 10 300 python bench.py --workload C5 --steps 100 --warmup 20 --no-cpu --no-variants > gpurun_out/r04b_dbg$v.json 2> gpurun_out/r04b_dbg$v.err || { echo "bench $v failed"; tail -5 gpurun_out/r04b_dbg$v.err; }
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04b_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'unparsable', e); continue
    print(f, d['value'], d['ms_per_step'], [(k['kernel'], k.get('avg_ms')) for k in d.get('kernels',[])], d['roofline'].get('peak_measured_write_only'))
PY
