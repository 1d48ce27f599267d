#!/bin/bash
# round 6, call i: the driver's own command on the final tree; the line kept under profiles/
set -o pipefail
OUT=gpurun_out/r06_i; mkdir -p $OUT
t0=$(date +%s)
timeout -k 10 580 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; rc=$?
echo "bench rc=$rc in $(( $(date +%s) - t0 )) s"; tail -2 $OUT/bench.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_i/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'])
r=d['roofline']; print('roofline', {k:r.get(k) for k in ('bound','kernel','achieved','frac','traffic','traffic_source','issue')})
print('fd_structured', {k:d['variants']['fd_structured'].get(k) for k in ('ms_per_step','frac','bound','counters')})
for k,v in d['configs'].items():
    if isinstance(v,dict): print(k, {q:v.get(q) for q in ('ms_per_step','value','wall_s')}, (v.get('kernel') or {}).get('frac'), (v.get('parity') or {}).get('ok'))
print('jacobian', json.dumps(d['configs']['C5_mindist'].get('jacobian_list'))[:1500])
PY
