#!/usr/bin/env python3
"""The flow of Examples/ComplexObstacles.py:19-63 on the MI355X path: one Dubins-like vehicle, degree 10, time optimal,
two Bezier "tracks" as shapeObstacles, SLSQP with max-speed, max-angular-rate and the SPATIAL separation constraint
(`_minDist` on every pair of vehicle and obstacles).

    python examples/example4_complex_obstacles.py [--robust]

Without --robust the reference's own search is tried first; on this very problem it hits a pair on which the
reference's gjkNew never returns (the reference's script did not finish in 120 s in the survey container either,
SURVEY.md 8(a) G4), the closure raises, and the script goes on with the robust search (tf* = 5.415 in 31 iterations).

What differs from the reference's script, and why:
  * the constraint's distance column (spatialSeparationConstraints(x)[:, 0]) is what goes to SLSQP: the reference hands
    over the whole (P, 3) array, which also subtracts maxSep from the two curve parameters and is not a 1-D constraint
    vector;
  * every constraint carries a `jac`: SciPy would otherwise call each closure n_x + 1 times per iteration; the spatial one
    is ONE obtg_min_dist launch with only the pairs each variable touches (BezOptimization.spatialSeparationJacobian);
  * the reference's lower bounds are +inf (ComplexObstacles.py:43-45), which no x satisfies; here only tf is bounded below.
Caps and statuses: a pair on which the reference's branch & bound does not end (depth 128 / 4 000 000 nodes here; the
reference recurses until Python's stack gives out) raises RuntimeError from the closure, as an exception would propagate
out of `minimize` there.  --robust uses the true-minimum search (obtg_min_dist_robust) instead, which always returns.
"""
import os
import sys
import time

import numpy as np
import scipy.optimize as sop

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import optimalbeziertrajectorygeneration_amd.bezier as bez  # was: import bezier as bez
from optimalbeziertrajectorygeneration_amd.optimization import BezOptimization  # was: from optimization import ...


def solve(robust):
    track1 = bez.Bezier(np.array([[8, 9, 10, 11, 12, 13, 12, 11, 10, 9, 8],
                                  [8, 10, 12, 14, 20, 14, 12, 10, 10, 9, 8]], dtype=float))
    track2 = bez.Bezier(np.array([[18, 13, 9, 6, 4, 3, 4, 6, 9, 13, 18],
                                  [3, 3, 4, 4, 4, 5, 5, 5, 7, 8, 3]], dtype=float))
    bezopt = BezOptimization(numVeh=1, dimension=2, degree=10, minimizeGoal='TimeOpt', maxSep=0.5, maxSpeed=5,
                             maxAngRate=0.5, initPoints=(2, 1), finalPoints=(15, 15), initSpeeds=1, finalSpeeds=1,
                             initAngs=np.pi / 2, finalAngs=np.pi / 2, shapeObstacles=[track1, track2])
    xGuess = bezopt.generateGuess()
    xGuess[-1] = 10
    lb = np.full(xGuess.size, -np.inf)
    lb[-1] = 1e-3
    bounds = sop.Bounds(lb, np.inf)
    ineqCons = [{'type': 'ineq', 'fun': bezopt.maxSpeedConstraints, 'jac': bezopt.maxSpeedJacobian},
                {'type': 'ineq', 'fun': bezopt.maxAngularRateConstraints, 'jac': bezopt.maxAngularRateJacobian},
                {'type': 'ineq', 'fun': lambda x: bezopt.spatialSeparationConstraints(x, robust=robust)[:, 0],
                 'jac': lambda x: bezopt.spatialSeparationJacobian(x, robust=robust, column=0)}]
    t0 = time.time()
    res = sop.minimize(bezopt.objectiveFunction, x0=xGuess, method='SLSQP', constraints=ineqCons, bounds=bounds,
                       options={'maxiter': 250, 'disp': False})
    dt = time.time() - t0
    d = bezopt.spatialSeparationConstraints(res.x, robust=robust)[:, 0]
    print('%s search: tf* = %.6f, %d iterations, success %s, %.2f s; distance margins to (track1, track2, track1-track2): %s'
          % ('robust' if robust else "reference's", res.x[-1], res.nit, res.success, dt, np.round(d, 4)))


def main():
    if "--robust" not in sys.argv:
        try:
            return solve(False)
        except RuntimeError as e:
            print("reference's search: %s -> switching to the robust search" % e)
    solve(True)


if __name__ == '__main__':
    main()
