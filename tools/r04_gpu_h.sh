#!/bin/bash
# PMC of the final tree's headline launch and of the C5 launches; the 3-D sweep at one row against the full batch
export TMPDIR=/tmp
mkdir -p gpurun_out
BENCH_ARGS="--no-variants" bash tools/pmc_pair_sweep.sh gpurun_out/r04_pmc_c3 > /dev/null 2>&1
python tools/pmc_reduce.py gpurun_out/r04_pmc_c3 k_pair_sweep > gpurun_out/r04_pmc_pair_sweep.txt
BENCH_ARGS="--no-variants --workload C5" bash tools/pmc_pair_sweep.sh gpurun_out/r04_pmc_c5 > /dev/null 2>&1
python tools/pmc_reduce.py gpurun_out/r04_pmc_c5 k_sep_dynamics_elev > gpurun_out/r04_pmc_c5.txt
python tools/pmc_reduce.py gpurun_out/r04_pmc_c5 k_gjk_swarm_planar >> gpurun_out/r04_pmc_c5.txt
for wl in C2 C2_file; do
  for b in 0 1 8; do
    args="--workload $wl --steps 300 --warmup 30 --no-cpu --no-variants"; [ $b -ne 0 ] && args="$args --batch $b"
    echo "$wl batch $b: $(python bench.py $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'ms/step', d['config']['evals_per_step_per_gpu'], 'rows')")" >> gpurun_out/r04_3d_one_row.txt
  done
done
cat gpurun_out/r04_3d_one_row.txt; head -30 gpurun_out/r04_pmc_pair_sweep.txt; cat gpurun_out/r04_pmc_c5.txt | head -50
