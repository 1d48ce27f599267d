#!/usr/bin/env python3
"""The flow of the reference's sequential planner (Examples/SequentialSwarm.py:157-192: 3-D, degree 3, vehicles planned
one after the other against the trajectories already fixed, SLSQP with the separation constraint only) on the MI355X
path.  Seeded targets stand in for the example's logo CSV: every vehicle climbs from the z = 0 face to a point of the
z = volume face a random few metres to the side of where it started, so neighbours get in each other's way without the
whole swarm crossing (the constraint -- every elevated control point of the squared distance above dsafe^2 -- is a
sufficient condition and is far from tight for head-on crossings at elev(10)).

    python examples/example3_sequential_swarm.py [numVeh]
    python examples/example3_sequential_swarm.py 1000 --logo      # the reference's own run: 1000 vehicles to the logo points

Runs the plan three ways: the reference's pairing through plain callbacks (SciPy's own finite differences), the same
with the one-call Jacobian, and the new vehicle against ALL fixed ones (what the example's docstrings describe).
--logo: Examples/SequentialSwarm.py:157-192 at its own size -- initial points and the 1000 logo targets as
tests/golden/sequential.npz holds them (written by the reference's Parameters / its CSV) -- planned with the new vehicle
against ALL earlier ones and the one-call Jacobian; the finished plan's margin over all 499 500 pairs from ONE
obtg_temporal_sep_min call.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimalbeziertrajectorygeneration_amd import sequential as SS  # was: the functions of Examples/SequentialSwarm.py


def climb_targets(inipts, volume, seed=5, sigma=6.0):
    rng = np.random.default_rng(seed)
    return np.clip(inipts[:, :2] + rng.normal(0.0, sigma, size=(inipts.shape[0], 2)), 0.0, volume)


def logo_run(nveh):
    """SequentialSwarm.py:157-192: nveh <= 1000 vehicles from the reference's initial points to its logo targets."""
    import time
    from optimalbeziertrajectorygeneration_amd import _capi
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "sequential.npz"))
    NDIM, DEG, VOLUME, DSAFE = 3, 3, 100, 1
    params = SS.Parameters(nveh, NDIM, DEG, VOLUME, DSAFE, finalpts=g["hawks_finalpts"][:nveh], seed=3)
    params.inipts = np.ascontiguousarray(g["hawks_inipts"][:nveh])
    for pairing in ("new_vs_all", "reference"):
        traj, results, dt = SS.plan(params, pairing=pairing, with_jac=True)
        t0 = time.time()
        ctx = _capi.Context(nveh, NDIM, DEG, 10)
        margins = ctx.temporal_sep_min(traj[None], DSAFE)[0]          # every pair of the finished plan, elev(10) minima - dsafe^2
        ctx.close()
        t_all = time.time() - t0
        nit = sum(r.nit for r in results)
        # pairs no plan can separate: targets (or starts) closer than dsafe -- the logo's 1000 points sit in a 100 x 100 square
        iu = np.triu_indices(nveh, 1)
        d_end = np.minimum(np.linalg.norm(params.finalpts[iu[0]] - params.finalpts[iu[1]], axis=1),
                           np.linalg.norm(params.inipts[iu[0]] - params.inipts[iu[1]], axis=1))
        doomed = d_end < DSAFE
        print("%-10s %4d vehicles to the logo points in %7.2f s: %d SLSQP iterations, %d vehicles not converged; all %d pairs "
              "checked in %.3f s: %d below 0, of them %d with end points closer than dsafe (no plan separates those); worst "
              "margin of the others %+.4e"
              % (pairing, nveh, dt, nit, sum(not r.success for r in results), margins.size, t_all, int((margins < 0).sum()),
                 int((doomed & (margins < 0)).sum()), margins[~doomed].min()))


def main():
    if "--logo" in sys.argv:
        return logo_run(int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1000)
    nveh = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    NDIM, DEG, VOLUME, DSAFE = 3, 3, 100, 1          # SequentialSwarm.py:158-162
    params = SS.Parameters(nveh, NDIM, DEG, VOLUME, DSAFE, seed=3)
    params = SS.Parameters(nveh, NDIM, DEG, VOLUME, DSAFE, finalpts=climb_targets(params.inipts, VOLUME), seed=3)
    for pairing, with_jac in (('reference', False), ('reference', True), ('new_vs_all', True)):
        traj, results, dt = SS.plan(params, pairing=pairing, with_jac=with_jac)
        # feasibility of the finished plan: every vehicle against every earlier one
        worst = np.inf
        for i in range(1, nveh):
            worst = min(worst, float(SS.new_vs_all(traj[i * NDIM:(i + 1) * NDIM], traj[:i * NDIM], NDIM, DSAFE).min()))
        print('%-10s %-22s %3d vehicles in %6.2f s  (%d SLSQP iterations, %d not converged)  worst pair margin %+.3e'
              % (pairing, 'one-call Jacobian' if with_jac else "SciPy's differences", nveh, dt,
                 sum(r.nit for r in results), sum(not r.success for r in results), worst))


if __name__ == '__main__':
    main()
