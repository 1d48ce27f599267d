#!/usr/bin/env python3
"""The flow of the reference's swarm driver (Examples/SwarmOfAerialVehicles.py:137-170: 3-D, degree 5,
Euclidean objective, temporal separation as the only constraint, SLSQP) on the MI355X path, on a
synthetic crossing swarm instead of the image-derived targets.  Solved twice: with SciPy's own finite
differences over the callback, and with the structured Jacobian provider (one `jac` key more).

    python examples/example2_swarm_3d.py [numVeh]

The start is the straight lines plus N(0, 0.2) noise (seed 2).  SLSQP on these sufficient-condition constraints is touchy about
its start: of ten (seed, std) pairs tried at 5 vehicles eight converge in 44-282 iterations to 57.6037, two (seed 1 std 0.2, seed 3
std 0.5) are still moving at 400 iterations, 2e-5 from feasibility -- which ones depends on the last bits of the callback's
values (tests/test_gpu_dropin.py holds feasibility and objective, not the iteration count).
"""
import os
import sys
import time

import numpy as np
import scipy.optimize as sop

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimalbeziertrajectorygeneration_amd.optimization import BezOptimization  # was: from optimization import ...


def crossing_swarm(numVeh, seed=3):
    """Vehicles start on a line at z = 0 and end on the mirrored line at z = 10: every straight path crosses the others'."""
    rng = np.random.default_rng(seed)
    xs = np.arange(numVeh) * 2.0
    init = np.stack([xs, np.zeros(numVeh), np.zeros(numVeh)], axis=1)
    final = np.stack([xs[::-1], np.full(numVeh, 1.0), np.full(numVeh, 10.0)], axis=1)
    return init + rng.normal(0, 0.05, init.shape), final + rng.normal(0, 0.05, final.shape)


def solve(numVeh=5, with_jac=True, maxiter=400, separationRows='all', seed=2, activeRows=2):
    init, final = crossing_swarm(numVeh)
    bezopt = BezOptimization(numVeh=numVeh, dimension=3, degree=5, minimizeGoal='Euclidean', maxSep=0.9,
                             initPoints=init, finalPoints=final, separationRows=separationRows, activeRows=activeRows)
    x0 = bezopt.generateGuess(std=0.2, seed=seed)
    con = {'type': 'ineq', 'fun': bezopt.temporalSeparationConstraints}
    if with_jac:
        con['jac'] = bezopt.temporalSeparationJacobian
    t0 = time.time()
    res = sop.minimize(bezopt.objectiveFunction, x0=x0, method='SLSQP', constraints=[con],
                       jac=bezopt.objectiveGradient if with_jac else None, options={'maxiter': maxiter, 'disp': False})
    return bezopt, res, time.time() - t0


def main():
    numVeh = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    k = 2 if numVeh <= 5 else 4             # rows per pair of the 'active' run (profiles/r05_experiments/active_rows_scan.txt)
    for with_jac, rows in ((False, 'all'), (True, 'all'), (True, 'min'), (True, 'active')):
        bezopt, res, dt = solve(numVeh, with_jac, separationRows=rows, activeRows=k)
        sep = bezopt.temporalSeparationConstraints(res.x)
        what = ('structured Jacobian provider' if with_jac else 'SciPy finite differences') + (', one row per pair' if rows == 'min' else ', the %d smallest rows per pair' % k if rows == 'active' else '')
        print('%-46s objective %.6f  nit %3d  nfev %5d  converged %s  min separation margin %+.2e  %4d rows  %.2f s'
              % (what, res.fun, res.nit, res.nfev, res.success, sep.min(), sep.size, dt))


if __name__ == '__main__':
    main()
