#!/bin/bash
# Every counter figure bench.py reports, taken on ONE tree in one call (run through gpurun; tools/r06_counters.py then writes
# profiles/counters.json with the source hashes of the library that ran):
#   gpurun_out/<tag>/<workload>__<pass>/   pass = stats (rocprofv3 --kernel-trace --stats), fetch / write (FETCH_SIZE / WRITE_SIZE,
#   separate passes as MI355X_MICROARCH.md prescribes), issue (SQ_INSTS_* / waves), lds (LDS index cycles, conflicts, waits)
set -o pipefail
export TMPDIR=/tmp
TAG=${1:-r06_counters}
OUT=gpurun_out/$TAG; mkdir -p $OUT
python3 - > $OUT/meta.json <<'PY'
import json
from optimalbeziertrajectorygeneration_amd import _capi
print(json.dumps({u: _capi.source_hash(u) for u in ("gjk_kernels", "bern_kernels", "capi", "all")}))
PY
ISSUE="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES"
LDS="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"
run_passes() {   # name, then the program and its arguments
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${name}__stats -o run -- "$@" > $OUT/${name}__stats.log 2>&1 || { echo "$name stats failed"; tail -3 $OUT/${name}__stats.log; return 1; }
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${name}__fetch -o run -- "$@" > $OUT/${name}__fetch.log 2>&1 || { echo "$name fetch failed"; return 1; }
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${name}__write -o run -- "$@" > $OUT/${name}__write.log 2>&1 || { echo "$name write failed"; return 1; }
  rocprofv3 --pmc $ISSUE --output-format csv -d $OUT/${name}__issue -o run -- "$@" > $OUT/${name}__issue.log 2>&1 || { echo "$name issue failed"; return 1; }
  rocprofv3 --pmc $LDS --output-format csv -d $OUT/${name}__lds -o run -- "$@" > $OUT/${name}__lds.log 2>&1 || { echo "$name lds failed"; return 1; }
  echo "$name done" | tee -a $OUT/progress.log
}
B="--no-cpu --no-variants --no-configs"
run_passes C3 python3 bench.py --workload C3 --steps 20 --warmup 3 $B &&
run_passes C5 python3 bench.py --workload C5 --steps 10 --warmup 2 $B &&
run_passes C2 python3 bench.py --workload C2 --steps 20 --warmup 3 $B &&
run_passes C2_file python3 bench.py --workload C2_file --steps 20 --warmup 3 $B &&
run_passes C3_fd_structured python3 tools/timeline_structured_run.py C3 &&
run_passes C5_fd_structured python3 tools/timeline_structured_run.py C5 &&
run_passes C5_mindist python3 bench.py --mode mindist --steps 100 --warmup 20 --no-cpu --mindist-legs reference_algorithm,jacobian_list,curve_polygon_reference_algorithm &&
run_passes C4 python3 bench.py --workload C4 --steps 3 --warmup 1 $B &&
run_passes C4_fd_structured python3 tools/timeline_structured_run.py C4
cat $OUT/progress.log
