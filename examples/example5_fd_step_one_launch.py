#!/usr/bin/env python3
"""What one SLSQP iteration asks of the callbacks of a planar swarm problem -- every constraint family at x and at its
n_x forward-difference neighbours -- from ONE launch, results staying in HBM (obtg_constraint_sweep_fd_structured_dev):

    python examples/example5_fd_step_one_launch.py [numVeh]

The problem is the reference's `BezOptimization` set-up (optimization.py:21-63: given end points, fixed tf): x holds the
interior control points, SciPy's 2-point rule evaluates each closure at x and at x + h e_k (abs_step
1.4901161193847656e-08).  Row k + 1 of that batch is row 0 with ONE control point advanced, so the launch evaluates row 0
in full and per row only what the advanced point's vehicle touches, and writes the B = n_x + 1 rows of
  temporalSeparationConstraints   (optimization.py:83-107),
  maxSpeedConstraints             (:135-151),
  maxAngularRateConstraints       (:171-187)
and of the gjkNew hull sweep over every vehicle pair and vehicle-polygon pair (flag, closest points, distance).  The
dense Jacobians SLSQP wants are then (F[1:] - F[0]) / dx on the device.  Checked here against the per-family
`...Jacobian` providers of the drop-in class (what examples 1-4 hand to SciPy): separation and angular rate equal entry
for entry; the speed rows are formed by a different kernel there and agree to the last bits of F, i.e. to ~1e-7 in J.
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimalbeziertrajectorygeneration_amd import _capi, synth
from optimalbeziertrajectorygeneration_amd.optimization import BezOptimization

FD_STEP = 1.4901161193847656e-08


def main(N=16, n=10, M=3, verbose=True):
    rng = np.random.default_rng(7)
    ini, fin = rng.uniform(0, 100, size=(N, 2)), rng.uniform(0, 100, size=(N, 2))
    bezopt = BezOptimization(numVeh=N, dimension=2, degree=n, minimizeGoal='Euclidean', maxSep=0.9, maxSpeed=5.0,
                             maxAngRate=1.0, initPoints=ini, finalPoints=fin, tf=10.0)
    x = bezopt.generateGuess(std=2.0, seed=11)
    Y0 = np.ascontiguousarray(bezopt.reshapeVector(x))                 # [(N*2), n+1]: what every closure starts from
    nx, B = x.size, x.size + 1
    dx = (x + FD_STEP) - x                                             # SciPy's dx: the step actually taken

    ctx = _capi.Context(N, 2, n, 0)
    ctx.set_stream(_capi.torch_stream())
    pa, pb = synth.swarm_pairs(N, M)
    ctx.set_polygons(*synth.pack_polys(synth.polygon_obstacles(M, seed=5)))
    ctx.set_hull_pairs(pa, pb)
    f64, i32, dev = torch.float64, torch.int32, "cuda"
    P, L, Ps = ctx.num_pairs, 2 * n + 1, len(pa)
    d0 = torch.from_numpy(Y0).to(dev)
    d_tf = torch.full((B,), 10.0, dtype=f64, device=dev)
    sep = torch.empty((B, P * L), dtype=f64, device=dev)
    sp = torch.empty((B, ctx.len_speed), dtype=f64, device=dev)
    an = torch.empty((B, ctx.len_ang_rate), dtype=f64, device=dev)
    flag = torch.empty((B, Ps), dtype=i32, device=dev)
    p1, p2 = torch.empty((B, Ps, 3), dtype=f64, device=dev), torch.empty((B, Ps, 3), dtype=f64, device=dev)
    dist = torch.empty((B, Ps), dtype=f64, device=dev)
    st = torch.empty((B, Ps), dtype=i32, device=dev)

    def step():
        ctx.constraint_sweep_fd_structured_dev(d0.data_ptr(), 1, FD_STEP, d_tf.data_ptr(), B, 0.9, sep.data_ptr(), 5.0, True, 1.0,
                                               sp.data_ptr(), an.data_ptr(), flag.data_ptr(), p1.data_ptr(), p2.data_ptr(),
                                               dist.data_ptr(), None, st.data_ptr(), 128, 500)
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / 20
    ddx = torch.from_numpy(dx).to(dev)[:, None]
    J = {k: ((F[1:] - F[0:1]) / ddx).T.cpu().numpy() for k, F in (("sep", sep), ("speed", sp), ("ang", an))}

    # the drop-in class's providers: one structured call per family, on host arrays
    t0 = time.perf_counter()
    ref = {"sep": bezopt.temporalSeparationJacobian(x), "speed": bezopt.maxSpeedJacobian(x), "ang": bezopt.maxAngularRateJacobian(x)}
    ms_ref = 1e3 * (time.perf_counter() - t0)
    same = {k: bool(np.array_equal(J[k], ref[k])) for k in ("sep", "ang")}
    # the speed rows come out of the angular-rate kernel here (one pass forms |v|^2 for both) and out of the speed-only kernel
    # in the provider: the same values to an ulp or two, which the division by dx = 1.5e-8 turns into ~1e-7 in a Jacobian entry
    dj = float(np.max(np.abs(J["speed"] - ref["speed"])))
    same["speed (|dJ| <= 1e-5: another kernel's last bits / dx)"] = dj <= 1e-5
    if verbose:
        print("%d vehicles, degree %d, %d polygons: n_x = %d, %d + %d + %d constraint rows, %d hull pairs"
              % (N, n, M, nx, sep.shape[1], sp.shape[1], an.shape[1], Ps))
        print("one launch for the whole finite-difference step: %.3f ms (%.1f MB of results left in HBM)"
              % (ms, sum(t.numel() * t.element_size() for t in (sep, sp, an, flag, p1, p2, dist, st)) / 1e6))
        print("Jacobians from it equal the per-family providers' (%.1f ms for the three on host arrays): %s" % (ms_ref, same))
    ctx.use_own_stream()
    ctx.close()
    return same, ms


if __name__ == "__main__":
    ok, _ = main(N=int(sys.argv[1]) if len(sys.argv) > 1 else 16)
    sys.exit(0 if all(ok.values()) else 1)
