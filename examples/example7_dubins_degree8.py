#!/usr/bin/env python3
"""The two degree-8 Dubins-car drivers of the reference on the MI355X path, flow for flow:

  time_optimal   Examples/DubinsCarTimeOptimal.py:60-137 -- one car, (3, 0) -> (7, 10), 2 point obstacles, unit speeds at
                 both ends, minimise tf; if SLSQP gives up, start again from a noisier straight line (std 1, 2, ...)
  example2       Examples/DubinsCarExample2.py:60-140   -- (0, 0) -> (12, 8), 7 point obstacles, bounds on every
                 variable (tf in [1e-4, 50]), DEG_ELEV read from the optimization module, same retry loop

    python examples/example7_dubins_degree8.py [time_optimal|example2] [DEG_ELEV]

Only the import lines differ from the reference's scripts, plus: the retries draw SEEDED guesses (seed 100 + std; the
reference draws unseeded, so its runs differ from call to call), and a retry that raises TypeError -- SLSQP stepped to
tf <= 0, where the reference dies in optimization.py:604 -- is reported and followed by the next one instead of ending
the script.  tests/test_gpu_dropin.py::test_degree8_driver_flows holds every attempt to the reference's own outcome
(tests/golden/drivers.npz).
"""
import os
import sys
import time

import numpy as np
import scipy.optimize as sop

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import optimalbeziertrajectorygeneration_amd.bezier as bez                      # was: import bezier as bez
import optimalbeziertrajectorygeneration_amd.optimization as opt_mod            # was: from optimization import DEG_ELEV, ...
from optimalbeziertrajectorygeneration_amd.optimization import BezOptimization

OBS = [(3, 2), (7, 6), (9, 9), (4, 5), (5, 8), (3, 7), (7, 3)]                   # DubinsCarExample2.py:60-66


def problem(which):
    numVeh, dim, deg = 1, 2, 8
    if which == 'time_optimal':
        bezopt = BezOptimization(numVeh=numVeh, dimension=dim, degree=deg, minimizeGoal='TimeOpt', maxSep=1, maxSpeed=5,
                                 maxAngRate=1, initPoints=[(3, 0)], finalPoints=[(7, 10)], initSpeeds=[1] * numVeh,
                                 finalSpeeds=[1] * numVeh, initAngs=[np.pi / 2], finalAngs=[np.pi / 2],
                                 pointObstacles=[[3, 2], [6, 7]])
        return bezopt, None
    bezopt = BezOptimization(numVeh=numVeh, dimension=dim, degree=deg, minimizeGoal='TimeOpt', maxSep=1, maxSpeed=3,
                             maxAngRate=np.pi / 2, initPoints=[(0, 0)], finalPoints=[(12, 8)], initSpeeds=[1] * numVeh,
                             finalSpeeds=[1] * numVeh, tf=8, initAngs=[np.pi / 2], finalAngs=[0], pointObstacles=OBS)
    return bezopt, sop.Bounds([-100] * 10 + [0.0001], [100] * 10 + [50], [False] * 10 + [True])


def solve(which='time_optimal', starts=None, max_retries=100, verbose=False):
    """-> (bezopt, attempts): attempts = [(x0, result or the TypeError raised)], the last one successful unless the
    retries ran out.  starts: explicit list of start vectors (the tests replay the reference's)."""
    bezopt, bounds = problem(which)
    xGuess = bezopt.generateGuess(std=0)
    ineqCons = [{'type': 'ineq', 'fun': bezopt.temporalSeparationConstraints},
                {'type': 'ineq', 'fun': bezopt.maxSpeedConstraints},
                {'type': 'ineq', 'fun': bezopt.maxAngularRateConstraints},
                {'type': 'ineq', 'fun': lambda x: x[-1]}]
    _ = bez.Bezier(bezopt.reshapeVector(xGuess))          # the scripts' warm-up of the elevation / product tables
    _.elev(max(int(opt_mod.DEG_ELEV), 1))
    _ = _ * _
    kw = dict(method='SLSQP', constraints=ineqCons, options={'maxiter': 250, 'disp': verbose, 'iprint': 1})
    if bounds is not None:
        kw['bounds'] = bounds
    attempts, std = [], 0
    while True:
        x0 = starts[std] if starts is not None else xGuess
        try:
            results = sop.minimize(bezopt.objectiveFunction, x0=x0, **kw)
        except TypeError as e:                            # the reference's script ends here
            results = e
        attempts.append((x0, results))
        done = not isinstance(results, TypeError) and results.success
        std += 1
        if done or std > max_retries or (starts is not None and std >= len(starts)):
            return bezopt, attempts
        xGuess = bezopt.generateGuess(std=std, seed=100 + std)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else 'time_optimal'
    if len(sys.argv) > 2:
        opt_mod.DEG_ELEV = int(sys.argv[2])
    t0 = time.time()
    bezopt, attempts = solve(which)
    dt = time.time() - t0
    for k, (x0, r) in enumerate(attempts):
        if isinstance(r, TypeError):
            print('attempt %d (std %d): TypeError -- SLSQP stepped to tf <= 0 (the reference script dies here)' % (k, k))
        else:
            print('attempt %d (std %d): success %s, status %d, %d iterations, tf = %.9f' % (k, k, r.success, r.status, r.nit, r.fun))
    r = attempts[-1][1]
    print('---\nComputation Time: {}\n---'.format(dt))
    if not isinstance(r, TypeError):
        cpts = bezopt.reshapeVector(r.x)
        sep = bezopt.temporalSeparationConstraints(r.x)
        print('DEG_ELEV %d: tf* = %.9f; smallest separation / speed / angular-rate margin %.2e / %.2e / %.2e; end point %s'
              % (opt_mod.DEG_ELEV, r.x[-1], sep.min(), bezopt.maxSpeedConstraints(r.x).min(),
                 bezopt.maxAngularRateConstraints(r.x).min(), cpts[:, -1].tolist()))


if __name__ == '__main__':
    main()
