#!/bin/bash
# all-rows C3-size SLSQP iterations on the GPU box (minutes of single-core SciPy work): keeps gpurun_out/ ticking
OUT=gpurun_out/${1:-r02d}; mkdir -p $OUT
( while true; do sleep 50; date +%T >> $OUT/tick.log; done ) &
TICK=$!
timeout -k 10 ${3:-1000} python tools/slsqp_c3_probe.py --variant jac --iters ${2:-3} > $OUT/slsqp_jac.json 2> $OUT/slsqp_jac.err
rc=$?
kill $TICK
cat $OUT/slsqp_jac.json
exit $rc
