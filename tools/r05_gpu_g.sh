#!/bin/bash
set -o pipefail
OUT=gpurun_out/r05_g; mkdir -p $OUT
timeout -k 5 400 python -m pytest tests/test_gpu_dropin.py -m gpu -q -k "active_separation" > $OUT/active_tests.log 2>&1; rc=$?
tail -12 $OUT/active_tests.log
python3 - > $OUT/active_scan.log 2>&1 <<'PY'
import importlib.util, os
spec = importlib.util.spec_from_file_location("ex2", "examples/example2_swarm_3d.py"); ex = importlib.util.module_from_spec(spec); spec.loader.exec_module(ex)
for nveh in (5, 8, 12):
    for rows, k in (("all", 0), ("min", 1), ("active", 2), ("active", 3), ("active", 4)):
        bo, r, dt = ex.solve(nveh, with_jac=True, separationRows=rows, activeRows=max(k, 1), maxiter=400)
        full = ex.solve.__globals__["BezOptimization"]
        chk = ex.solve(nveh, with_jac=True, maxiter=1)[0].temporalSeparationConstraints(r.x).min()
        print("veh %2d rows %-6s k %d: success %s nit %3d fun %.6f full-set margin %+.2e rows %d  %.2fs" % (nveh, rows, k, r.success, r.nit, r.fun, chk, bo.temporalSeparationConstraints(r.x).size, dt), flush=True)
PY
cat $OUT/active_scan.log
exit $rc
