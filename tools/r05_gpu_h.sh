#!/bin/bash
set -o pipefail
OUT=gpurun_out/r05_h; mkdir -p $OUT
python3 - > $OUT/active_scan_elev.log 2>&1 <<'PY'
import importlib.util, os
spec = importlib.util.spec_from_file_location("ex2", "examples/example2_swarm_3d.py"); ex = importlib.util.module_from_spec(spec); spec.loader.exec_module(ex)
import optimalbeziertrajectorygeneration_amd.optimization as opt
for R in (10, 4):
    opt.DEG_ELEV = R
    for nveh in (5, 8, 12):
        for rows, k in (("all", 0), ("min", 1), ("active", 2), ("active", 3), ("active", 4)):
            bo, r, dt = ex.solve(nveh, with_jac=True, separationRows=rows, activeRows=max(k, 1), maxiter=400)
            chk = ex.solve(nveh, with_jac=True, maxiter=1)[0].temporalSeparationConstraints(r.x).min()
            print("DEG_ELEV %2d veh %2d rows %-6s k %d: success %s nit %3d fun %.6f full-set margin %+.2e rows %d  %.2fs" % (R, nveh, rows, k, r.success, r.nit, r.fun, chk, bo.temporalSeparationConstraints(r.x).size, dt), flush=True)
PY
cat $OUT/active_scan_elev.log
