"""Exact (rational-arithmetic) angular-rate control points of the near-stop fixture's vehicles, rounded to float64:
adds `exact_tf<tf>` arrays to tests/golden/nearstop.npz.  OUR computation (not the reference's): the polynomial
identities of optimization.py:578-611 evaluated in fractions, the yardstick for "which float64 evaluation is closer".
    python tools/nearstop_exact.py"""
import os, sys
from fractions import Fraction as F
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.angrate_conditioning import exact_ang

path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "nearstop.npz")
g = dict(np.load(path))
Y = g["Y"]
N, n, R = int(g["par"][0]), int(g["par"][2]), int(g["par"][3])
for tf in g["tfs"]:
    rows = []
    for v in range(N):
        ex, _ = exact_ang(Y[2 * v], Y[2 * v + 1], F(float(tf)), R, 1)
        rows.append([float(e) if e is not None else float("nan") for e in ex])
    g["exact_tf%g" % tf] = np.array(rows).reshape(-1)
    ref = g["angrate_tf%g" % tf]
    print("tf %g: reference vs exact, per vehicle max |ref - exact| / max|exact|:" % tf,
          ["%.2e" % (np.abs(ref.reshape(N, -1)[v] - np.array(rows[v])).max() / np.abs(rows[v]).max()) for v in range(N)])
np.savez_compressed(path, **g)
