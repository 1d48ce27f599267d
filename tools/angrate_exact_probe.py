"""obtg_ctx_set_ang_rate_order(2) against exact rational arithmetic on the (n, R) grid of tools/angrate_order_probe.py:
per (n, R) the vehicles whose rows the double-double pass changed most (the ill-conditioned ones) and two others are
evaluated in fractions (tools/angrate_conditioning.exact_ang) and every element of their rows compared.
    python tools/angrate_exact_probe.py [--full]        (--full adds the slow (15, 200) corner)"""
import os, sys, json, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimalbeziertrajectorygeneration_amd import _capi as capi, synth
from tools.angrate_conditioning import exact_ang
from fractions import Fraction as F

grid = [(3, 10), (5, 30), (7, 60), (10, 100), (10, 30), (15, 60)] + ([(15, 200)] if "--full" in sys.argv else [])
out = {}
for n, R in grid:
    L4 = 4 * (n + R) + 1
    worst = {"exact_order": 0.0, "default_order": 0.0, "rows_checked": 0, "rows_changed_by_the_pass": 0}
    t0 = time.time()
    for seed in range(3):
        N, B = 64, 4
        Y = synth.swarm_control_points(N, 2, n, seed=40 + seed)
        Yb = synth.fd_batch(Y, B=B, h=0.5)
        tf = np.linspace(4.0, 25.0, B)
        res = {}
        for order in (0, 2):
            ctx = capi.Context(N, 2, n, R)
            ctx.set_ang_rate_order(order)
            res[order] = ctx.ang_rate(Yb, tf, 1.0).reshape(B, N, L4)
            ctx.close()
        diff = np.nanmax(np.abs(res[0] - res[2]) / np.maximum(1.0, np.abs(res[2])), axis=2)        # [B][N]
        worst["rows_changed_by_the_pass"] += int((diff > 0).sum())
        order_ = np.dstack(np.unravel_index(np.argsort(-diff, axis=None), diff.shape))[0]
        pick = [tuple(order_[0]), tuple(order_[1]), (B - 1, N - 1), (0, 0)]
        for b, v in pick:
            ex, _ = exact_ang(Yb[b, 2 * v], Yb[b, 2 * v + 1], F(float(tf[b])), R, 1)
            exf = np.array([float(e) if e is not None else np.nan for e in ex])
            fin = np.isfinite(exf)
            for order, key in ((2, "exact_order"), (0, "default_order")):
                g = res[order][b, v]
                el = np.abs(g[fin] - exf[fin]) / np.maximum(np.abs(exf[fin]), 1e-6 * np.abs(exf[fin]).max())
                worst[key] = max(worst[key], float(el.max()))
            worst["rows_checked"] += 1
    out["n=%d R=%d" % (n, R)] = worst
    print("n=%2d R=%3d  rows the pass changed %4d of %d;  element-wise distance from the exact value over %d checked rows: exact order %.2e, default order %.2e  (%.0f s)"
          % (n, R, worst["rows_changed_by_the_pass"], 3 * 4 * 64, worst["rows_checked"], worst["exact_order"], worst["default_order"], time.time() - t0), flush=True)
print(json.dumps(out))
