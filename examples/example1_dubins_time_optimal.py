#!/usr/bin/env python3
"""The reference's Example1 driver flow (Examples/Example1_DubinsCarTimeOptimal.py:94-148: two
Dubins cars, degree 10, time-optimal, SLSQP) on the MI355X path.  Only the import lines differ
from a reference driver; the second solve hands SLSQP the batched finite-difference Jacobians.

    python examples/example1_dubins_time_optimal.py
"""
import os
import sys
import time

import numpy as np
import scipy.optimize as sop

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import optimalbeziertrajectorygeneration_amd.bezier as bez                      # was: import bezier as bez
import optimalbeziertrajectorygeneration_amd.optimization as opt_mod
from optimalbeziertrajectorygeneration_amd.optimization import BezOptimization  # was: from optimization import ...


def main():
    numVeh, dim, deg = 2, 2, 10
    bezopt = BezOptimization(numVeh=numVeh, dimension=dim, degree=deg, minimizeGoal='TimeOpt', maxSep=1,
                             maxSpeed=5, maxAngRate=1, initPoints=[(0, 5), (3, 0)], finalPoints=[(8, 4), (7, 10)],
                             initSpeeds=[1] * numVeh, finalSpeeds=[1] * numVeh, initAngs=[0, np.pi / 2],
                             finalAngs=[0, np.pi / 2])
    for elev in (0, 30, 100):
        # like the reference example: only the separation constraint is elevated (its own function with
        # a degElev argument, Example1:19-52, 124-125); speed / angular rate keep DEG_ELEV = 0
        sepCons = lambda x: opt_mod._temporalSeparationConstraints(   # noqa: E731
            bezopt.reshapeVector(x), numVeh, dim, bezopt.model['maxSep'], elev)
        xGuess = bezopt.generateGuess(std=0)
        cons = [{'type': 'ineq', 'fun': sepCons},
                {'type': 'ineq', 'fun': bezopt.maxSpeedConstraints},
                {'type': 'ineq', 'fun': bezopt.maxAngularRateConstraints},
                {'type': 'ineq', 'fun': lambda x: x[-1]}]
        t0 = time.time()
        res = sop.minimize(bezopt.objectiveFunction, x0=xGuess, method='SLSQP', constraints=cons,
                           options={'maxiter': 250, 'disp': False})
        t1 = time.time()
        cons_j = [dict(c) for c in cons]
        if elev == 0:
            cons_j[0]['jac'] = bezopt.temporalSeparationJacobian
        cons_j[1]['jac'] = bezopt.maxSpeedJacobian
        cons_j[2]['jac'] = bezopt.maxAngularRateJacobian
        cons_j[3]['jac'] = lambda x: np.eye(1, x.size, x.size - 1)
        res_j = sop.minimize(bezopt.objectiveFunction, x0=xGuess, method='SLSQP', constraints=cons_j,
                             options={'maxiter': 250, 'disp': False})
        t2 = time.time()
        print('DEG_ELEV %3d: tf* = %.9f (nit %d, %.2f s with callbacks)   tf* = %.9f (nit %d, %.2f s with batched Jacobians)'
              % (elev, res.fun, res.nit, t1 - t0, res_j.fun, res_j.nit, t2 - t1))
        cpts = bezopt.reshapeVector(res.x)
        curves = [bez.Bezier(cpts[i * dim:(i + 1) * dim]) for i in range(numVeh)]
        print('   end points:', [c.cpts[:, -1].tolist() for c in curves])


if __name__ == '__main__':
    main()
