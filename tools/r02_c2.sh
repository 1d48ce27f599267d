#!/bin/bash
OUT=gpurun_out/${1:-c2}; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -q -x -m gpu -k "gjk or min_dist or mindist or swarm_3d or spatial or c5" > $OUT/pytest.log 2>&1 || { tail -25 $OUT/pytest.log; exit 1; }
tail -1 $OUT/pytest.log
for w in C2 C2_file; do
  timeout -k 10 120 python bench.py --no-cpu --workload $w --steps 300 --warmup 20 > $OUT/$w.json 2> $OUT/$w.err || { tail -3 $OUT/$w.err; exit 1; }
  python - "$w" $OUT/$w.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(sys.argv[1], d["value"], d["ms_per_step"], [(k["kernel"], k["avg_ms"], k["frac"]) for k in d["kernels"]])
PY
done
timeout -k 10 120 python bench.py --mode mindist > $OUT/mindist.json 2> $OUT/mindist.err; python -c "
import json; d=json.load(open('$OUT/mindist.json')); print({k:v['ms_per_eval'] for k,v in d['variants'].items()})"
