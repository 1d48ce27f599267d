#!/usr/bin/env python3
"""The computation of Examples/MinDistBez2Bez.py:42-100 on the MI355X path (its plots left out): minimum distance
between the script's literal Bezier curves (`Bezier.minDist`, bezier.py:840-852 -> `_minDist`, :1283-1408) and between
those curves and its convex polygons (`Bezier.minDist2Poly`, :854-857 -> `_minDist2Poly`, :1411-1496).

    python examples/example6_min_dist_curves.py

Three ways, same numbers:
  * the drop-in `Bezier` methods, one pair per call, as the script calls them;
  * ONE `obtg_min_dist` / `obtg_min_dist2poly` call for all pairs of the script (what a caller with many pairs does);
  * `robust=True`: the true minimum (obtg_min_dist_robust), which differs from the reference's answer exactly where the
    reference's search stops early or never returns (status, not an exception, in the batched form).
Expected values: the reference's own results on these inputs (tests/golden/mindist_script.npz, written by gen_golden.py
from the reference: 15 of the 20 ordered curve pairs and 11 of the 15 curve / polygon pairs return there).  A pair the
reference does not finish on (its recursion overflows the stack, or the search does not end) raises from the method as the
reference would raise or hang, and carries a status in the batched form.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import optimalbeziertrajectorygeneration_amd.bezier as bez  # was: import bezier as bez
from optimalbeziertrajectorygeneration_amd import _capi


def curves_and_polys():
    cpts1 = np.array([(0, 1, 2, 3, 4, 5), (1, 2, 0, 0, 2, 1), (0, 1, 2, 3, 4, 5)], dtype=float)
    cpts2 = np.array([(0, 1, 2, 3, 4, 5), (3, 2, 0, 0, 2, 3), (5, 4, 3, 2, 1, 0)], dtype=float)
    cpts3 = np.array([(0, 1, 2, 3, 4, 5), (0, 1, 2, 3, 4, 5), (0, 0, 0, 0, 0, 0)], dtype=float)
    cpts4 = np.array([(5, 4, 3, 2, 1, 0), (0, 1, 2, 3, 4, 5), (0, 0, 0, 0, 0, 0)], dtype=float)
    cpts4[1, :] -= 1
    cpts5 = cpts1 - 3
    poly1 = np.array([(1, 3, 3), (1, 3, 2), (1, 4, 1), (3, 3, 3), (1, 5, 1)], dtype=float)
    poly2 = np.array([(1, 1, 3), (1, 1, 2), (1, 2, 1), (4, 0, 2), (1, 3, 1)], dtype=float)
    poly3 = np.array([(1, 1, 0), (1, 3, 0), (2, 5, 0), (4, 4, 0)], dtype=float)
    return [cpts1, cpts2, cpts3, cpts4, cpts5], [poly1, poly2, poly3]


def main(verbose=True):
    cp, polys = curves_and_polys()
    c = [bez.Bezier(x) for x in cp]
    say = print if verbose else (lambda *a, **k: None)
    out = {}

    say("curve to curve, as the script calls it (c3 against the others):")
    for name, other in (("c1", 0), ("c2", 1), ("c4", 3), ("c5", 4)):
        try:
            d, t3, to = c[2].minDist(c[other])
            say("  c3.minDist(%s) = %.9f at t3 = %.6f, t = %.6f" % (name, d, t3, to))
            out["c3-" + name] = d
        except (RecursionError, RuntimeError) as e:
            d, t3, to = c[2].minDist(c[other], robust=True)
            say("  c3.minDist(%s): %s: %s\n      robust=True: %.9f at t3 = %.6f, t = %.6f" % (name, type(e).__name__, str(e)[:70], d, t3, to))
            out["c3-" + name] = None
            out["c3-" + name + "-robust"] = d

    say("all ordered pairs of the five curves in ONE call (status 0 = returned, 1 / 2 / 3 = depth cap / node cap / inner gjkNew):")
    ctx = bez._ctx()
    pa, pb = zip(*[(i, j) for i in range(5) for j in range(5) if i != j])
    r = ctx.min_dist(np.stack(cp), list(pa), list(pb), eps=1e-9, max_depth=128, max_nodes=4000000)
    rr = ctx.min_dist_robust(np.stack(cp), list(pa), list(pb), eps=1e-9)
    for k, (i, j) in enumerate(zip(pa, pb)):
        st = int(r["status"][k])
        ref = "%.9f" % r["res"][k][0] if st == 0 else "  (status %d)" % st
        say("  c%d - c%d: reference's search %s   true minimum %.9f" % (i + 1, j + 1, ref, rr["res"][k][0]))
    out["batch"] = r
    out["batch_robust"] = rr
    from optimalbeziertrajectorygeneration_amd import synth
    pts, off = synth.pack_polys(polys)
    pc, pp = zip(*[(i, k) for i in range(5) for k in range(len(polys))])
    out["batch_poly"] = ctx.min_dist2poly(np.stack(cp), pts, off, list(pc), list(pp), eps=1e-6, max_depth=128, max_nodes=4000000)

    say("curve to polygon:")
    for i in range(5):
        for jp, poly in enumerate(polys):
            try:
                d, t, pt = c[i].minDist2Poly(poly)
                say("  c%d.minDist2Poly(poly%d) = %.8f at t = %.6f, polygon point %s" % (i + 1, jp + 1, d, t, np.round(pt, 6)))
                out["c%d-poly%d" % (i + 1, jp + 1)] = d
            except (RecursionError, RuntimeError) as e:
                d, t, pt = c[i].minDist2Poly(poly, robust=True)
                say("  c%d.minDist2Poly(poly%d): %s; robust=True: %.8f at t = %.6f" % (i + 1, jp + 1, type(e).__name__, d, t))
                out["c%d-poly%d" % (i + 1, jp + 1)] = None
    return out


if __name__ == "__main__":
    main()
