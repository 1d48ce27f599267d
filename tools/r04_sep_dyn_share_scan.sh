#!/bin/bash
# C5: share of the dynamics groups among every 16 block ids of k_sep_dynamics_elev (OBTG_SEP_DYN_SHARE)
mkdir -p gpurun_out; out=gpurun_out/r04_sep_dyn_share.txt; : > $out
for sh in 0 2 3 4 5 6 7 8; do
  if [ $sh -eq 0 ]; then unset OBTG_SEP_DYN_SHARE; else export OBTG_SEP_DYN_SHARE=$sh; fi
  line=$(timeout -k 10 120 python bench.py --workload C5 --steps 100 --warmup 20 --no-cpu --no-variants 2>/dev/null | tail -1)
  echo "share $sh $(python -c "import json,sys; d=json.loads(sys.argv[1]); print(d['ms_per_step'], [(k['kernel'],k['avg_ms']) for k in d['kernels']])" "$line")" >> $out
done
cat $out
