"""Cost of SLSQP's three constraint callbacks at one x (Example1's model): python tools/closure_probe.py"""
import sys, time, numpy as np
sys.path.insert(0, '.')
from optimalbeziertrajectorygeneration_amd.optimization import BezOptimization
bo = BezOptimization(numVeh=2, dimension=2, degree=10, minimizeGoal='TimeOpt', maxSep=1, maxSpeed=5, maxAngRate=1,
                     initPoints=[(0, 5), (3, 0)], finalPoints=[(8, 4), (7, 10)], initSpeeds=[1] * 2, finalSpeeds=[1] * 2,
                     initAngs=[0, np.pi / 2], finalAngs=[0, np.pi / 2])
x0 = bo.generateGuess(std=0)
fs = (bo.temporalSeparationConstraints, bo.maxSpeedConstraints, bo.maxAngularRateConstraints)
for f in fs: f(x0)
rng = np.random.default_rng(0)
xs = [x0 + rng.normal(0, 1e-3, x0.size) for _ in range(300)]
t = time.perf_counter()
for x in xs:
    for f in fs: f(x)
dt = (time.perf_counter() - t) / len(xs)
print('three constraint callbacks at one x: %.1f us' % (dt * 1e6))
