#!/usr/bin/env python3
"""End-to-end SLSQP iterations of a C3-size problem (64 vehicles, 2-D, degree 10: n_x = 1152) on the drop-in path.

What a user of the reference sees: seconds per SLSQP major iteration, split into the time spent inside our
constraint / Jacobian callbacks and the rest (SciPy's own least-squares step, Fortran on one core).  The
reference's callbacks alone cost ~115 s per iteration at this size.

    python tools/slsqp_c3_probe.py [--iters 3] [--variant jac|jac_min|fd_callbacks]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import scipy.optimize as sop

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimalbeziertrajectorygeneration_amd import synth                                  # noqa: E402
from optimalbeziertrajectorygeneration_amd.optimization import BezOptimization          # noqa: E402


class Timed(object):
    def __init__(self, f):
        self.f, self.t, self.n = f, 0.0, 0

    def __call__(self, x):
        t0 = time.perf_counter()
        r = self.f(x)
        self.t += time.perf_counter() - t0
        self.n += 1
        return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--variant", default="jac", choices=["jac", "jac_min", "fd_callbacks"])
    ap.add_argument("--nveh", type=int, default=64)
    args = ap.parse_args()
    N, d, n = args.nveh, 2, 10
    init, final = synth.swarm_points(N, d, seed=1234)
    bo = BezOptimization(numVeh=N, dimension=d, degree=n, minimizeGoal='Euclidean', maxSep=0.9, maxSpeed=30.0,
                         maxAngRate=5.0, initPoints=init, finalPoints=final, tf=10.0,
                         separationRows='min' if args.variant == "jac_min" else 'all')
    x0 = bo.generateGuess(std=1.0, seed=1)
    fams = [("temporal_sep", bo.temporalSeparationConstraints, bo.temporalSeparationJacobian),
            ("max_speed", bo.maxSpeedConstraints, bo.maxSpeedJacobian),
            ("max_ang_rate", bo.maxAngularRateConstraints, bo.maxAngularRateJacobian)]
    funs = {k: Timed(f) for k, f, _ in fams}
    jacs = {k: Timed(j) for k, _, j in fams}
    obj = Timed(bo.objectiveFunction)
    for k, f, j in fams:                 # warm-up: contexts, tables, pinned pools
        f(x0); j(x0)
    rows = {k: int(np.size(f(x0))) for k, f, _ in fams}
    out = {"variant": args.variant, "n_x": int(x0.size), "rows": rows, "rows_total": int(sum(rows.values()))}
    if args.variant == "fd_callbacks":
        # what SciPy's own 2-point differences cost through the plain callbacks: n_x + 1 calls per constraint
        from scipy.optimize._numdiff import approx_derivative
        t0 = time.perf_counter()
        for k, f, _ in fams:
            approx_derivative(funs[k], x0, method='2-point', abs_step=1.4901161193847656e-08)
        out["one_jacobian_set_by_callbacks_s"] = time.perf_counter() - t0
        out["callback_calls"] = {k: funs[k].n for k in funs}
        print(json.dumps(out))
        return
    cons = [{'type': 'ineq', 'fun': funs[k], 'jac': jacs[k]} for k, _, _ in fams]
    stamps = [time.perf_counter()]
    grad = Timed(bo.objectiveGradient)
    res = sop.minimize(obj, x0=x0, jac=grad, method='SLSQP', constraints=cons, callback=lambda xk: stamps.append(time.perf_counter()),
                       options={'maxiter': args.iters, 'disp': False})
    total = time.perf_counter() - stamps[0]
    t_cb = sum(t.t for t in funs.values()) + sum(t.t for t in jacs.values()) + obj.t + grad.t
    its = max(1, len(stamps) - 1)
    out.update({"iterations": int(res.nit), "status": int(res.status), "message": str(res.message),
                "total_s": total, "s_per_iteration": total / its,
                "iteration_wall_s": [stamps[i + 1] - stamps[i] for i in range(len(stamps) - 1)],
                "in_callbacks_s": t_cb, "in_callbacks_s_per_iteration": t_cb / its,
                "in_scipy_s_per_iteration": (total - t_cb) / its,
                "calls": {"fun": {k: funs[k].n for k in funs}, "jac": {k: jacs[k].n for k in jacs}, "objective": obj.n},
                "fun_s": {k: funs[k].t for k in funs}, "jac_s": {k: jacs[k].t for k in jacs},
                "reference_callbacks_s_per_iteration": 115.0})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
