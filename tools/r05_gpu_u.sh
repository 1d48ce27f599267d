#!/bin/bash
# diagnostic: the bench invocations of test_bench_rows_mode_two_ranks_tile_the_iteration through the launcher, one by one, each
# watched for a stall; stalled ranks are sent SIGABRT so that faulthandler prints where they are
OUT=gpurun_out/r05_u; mkdir -p $OUT; rm -f $OUT/summary.txt
export PYTHONFAULTHANDLER=1
A="--mode rows --workload C3 --batch 301 --steps 3 --warmup 1 --no-cpu"
run() { # name args...
  name=$1; shift
  python3 bench.py "$@" > $OUT/$name.out 2> $OUT/$name.err &
  lp=$!
  for i in $(seq 1 70); do kill -0 $lp 2>/dev/null || break; sleep 1; done
  if kill -0 $lp 2>/dev/null; then
    echo "$name: STALLED after 70 s" | tee -a $OUT/summary.txt
    for p in $(pgrep -P $lp); do kill -ABRT $p 2>/dev/null; done
    sleep 3
    kill -TERM $lp 2>/dev/null; sleep 2; kill -9 $lp 2>/dev/null
    grep -v "^\[Gloo\]" $OUT/$name.err | tail -60
    return 1
  fi
  echo "$name: finished ($(date +%T))" | tee -a $OUT/summary.txt
}
run one --gpus 1 $A --no-variants || exit 1
run two --gpus 2 --backend gloo --one-device $A --no-variants || exit 1
run three_sparse --gpus 3 --backend gloo --one-device --gather-minima $A --no-variants || exit 1
run two_dense --gpus 2 --backend gloo --one-device --gather-minima dense $A --no-variants || exit 1
run one_v --gpus 1 $A || exit 1
run two_v --gpus 2 --backend gloo --one-device $A || exit 1
run three_v --gpus 3 --backend gloo --one-device $A || exit 1
