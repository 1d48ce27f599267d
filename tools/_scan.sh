set -e
for sh in 2 3 5 8 11; do
OBTG_SEP_DYN_SHARE=$sh python bench.py --workload C5 --steps 100 --warmup 20 --no-cpu --no-variants > gpurun_out/b.json
python - $sh <<'P'
import json,sys
j=json.loads(open('gpurun_out/b.json').read().strip().splitlines()[-1])
print('share', sys.argv[1], j['ms_per_step'], [(k['kernel'],k['avg_ms']) for k in j['kernels']])
P
done
