#!/usr/bin/env python3
"""Examples/DrivingOnATrack.py:18-60 on the MI355X path: one car, degree 10, time optimal, driving between two Bezier
"tracks" (shapeObstacles built from plain lists), constructor arguments as scalars / bare tuples, and the SPATIAL
separation constraint (`_minDist` on every pair of vehicle and tracks) next to max speed and max angular rate.

    python examples/example8_driving_on_a_track.py [--raw]

The reference's script cannot run at its HEAD, for four independent reasons; tests/golden/drivers.npz records the first two
from the reference itself, tests/test_gpu_dropin.py::test_driving_on_a_track_flow holds this script to all of them:
  * `_minDist` overflows Python's stack on EVERY pair of this problem (vehicle-track1, vehicle-track2, track1-track2):
    RecursionError at the first constraint evaluation.  Here the same call raises the same exception (depth cap ->
    RecursionError), and the script goes on with the robust search (obtg_min_dist_robust);
  * `bezopt.spatialSeparationConstraints` returns the (P, 3) array of (distance, t1, t2) - maxSep, and SLSQP's wrapper
    concatenates it with the 1-D speed / angular-rate vectors: SciPy raises "all the input arrays must have same number of
    dimensions" (SciPy 1.15, scipy/optimize/_slsqp_py.py `_eval_constraint`).  `--raw` shows it; what goes to SLSQP
    otherwise is the distance column;
  * its lower bounds are +inf for every control point (DrivingOnATrack.py:44-46): SciPy treats lb == ub as "fixed" and
    pins every control point at inf -- with these bounds and the two constraints that do evaluate, the reference's own
    closures end SLSQP after one iteration with "Inequality constraints incompatible" (SciPy 1.15, run in the build
    container).  Here only tf is bounded (>= 1e-3), the evident intent;
  * `bezier.py` never imports gjkNew (bezier.py:21-22), so `minDist` raises NameError before any of the above.
"""
import os
import sys
import time

import numpy as np
import scipy.optimize as sop

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import optimalbeziertrajectorygeneration_amd.bezier as bez  # was: import bezier as bez
from optimalbeziertrajectorygeneration_amd.optimization import BezOptimization  # was: from optimization import ...


def problem():
    track1 = bez.Bezier([[0, 0, 0, 3, 4, 5, 6, 7, 10, 10, 10],
                         [0, 3, 4, 5, 6, 6, 6, 6, 7, 8, 10]])
    track2 = bez.Bezier([[4,  4,  4,  7,  8,  9, 10, 11, 14, 14, 14],
                         [0, 3, 4, 4, 4, 5, 5, 5, 7, 8, 10]])
    tracks = [track1, track2]
    bezopt = BezOptimization(numVeh=1, dimension=2, degree=10, minimizeGoal='TimeOpt', maxSep=0.5, maxSpeed=5,
                             maxAngRate=0.5, initPoints=(2, 1), finalPoints=(12, 9), initSpeeds=1, finalSpeeds=1,
                             initAngs=np.pi / 2, finalAngs=np.pi / 2, shapeObstacles=tracks)
    xGuess = bezopt.generateGuess()
    xGuess[-1] = 10
    return bezopt, xGuess


def solve(robust=True, raw=False, maxiter=250):
    bezopt, xGuess = problem()
    lb = np.full(xGuess.size, -np.inf)
    lb[-1] = 1e-3
    bounds = sop.Bounds(lb, np.inf)
    if raw:                      # the reference's wiring: the (P, 3) array as it is -- SciPy's concatenate refuses it
        spatial = lambda x: bezopt.spatialSeparationConstraints(x, robust=robust)         # noqa: E731
    else:
        spatial = lambda x: bezopt.spatialSeparationConstraints(x, robust=robust)[:, 0]   # noqa: E731
    ineqCons = [{'type': 'ineq', 'fun': bezopt.maxSpeedConstraints},
                {'type': 'ineq', 'fun': bezopt.maxAngularRateConstraints},
                {'type': 'ineq', 'fun': spatial}]
    startTime = time.time()
    results = sop.minimize(bezopt.objectiveFunction, x0=xGuess, method='SLSQP', constraints=ineqCons, bounds=bounds,
                           options={'maxiter': maxiter, 'disp': False})
    return bezopt, results, time.time() - startTime


def main():
    try:
        solve(robust=False, maxiter=1)
        robust = False
    except (RecursionError, RuntimeError) as e:
        print("the reference's search: %s: %s -> the robust search" % (type(e).__name__, e))
        robust = True
    if '--raw' in sys.argv:
        try:
            solve(robust=robust, raw=True, maxiter=1)
        except ValueError as e:
            print("the (P, 3) array handed to SLSQP as it is: ValueError: %s -> the distance column" % e)
    bezopt, results, dt = solve(robust=robust)
    print('---\nComputation Time: {}\n---'.format(dt))
    print('success %s (%s), %d iterations, tf = %s' % (results.success, results.message, results.nit, results.x[-1]))
    d = bezopt.spatialSeparationConstraints(results.x, robust=True)
    print('distance margins to (track1, track2) and between the tracks: %s; speed / angular-rate margins %.2e / %.2e'
          % (np.round(d[:, 0], 4), bezopt.maxSpeedConstraints(results.x).min(), bezopt.maxAngularRateConstraints(results.x).min()))


if __name__ == '__main__':
    main()
