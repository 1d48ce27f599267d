"""Reference-shaped NumPy restatement of the Bernstein constraint closures (BASELINE.md section 3,
item 1): the same per-pair / per-vehicle Python loop order as optimization.py:311-459, NumPy
vector ops per curve, but without the reference's `Bezier` objects.  TEST / BENCH INFRASTRUCTURE
ONLY (cpu_baseline leg); single thread like the reference.  Checked against the C oracle in
tests/test_oracle_golden.py."""
import numpy as np
from scipy.special import binom

_cache = {}


def _elev_matrix(N, R):                      # bezier.py:1127-1147
    key = ("e", N, R)
    if key not in _cache:
        T = np.zeros((N + 1, N + R + 1))
        for i in range(N + R + 1):
            den = binom(N + R, i)
            for j in range(N + 1):
                T[j, i] = binom(N, j) * binom(R, i - j) / den
        _cache[key] = T
    return _cache[key]


def _prod_matrix(m, n):                     # bezier.py:1183-1208: dense ((m+1)(n+1)) x (m+n+1)
    key = ("p", m, n)
    if key not in _cache:
        C = np.zeros(((m + 1) * (n + 1), m + n + 1))
        for k in range(m + n + 1):
            den = binom(m + n, k)
            for j in range(max(0, k - n), min(m, k) + 1):
                C[j * (n + 1) + (k - j), k] = binom(m, j) * binom(n, k - j) / den
        _cache[key] = C
    return _cache[key]


def _mul(a, b):                              # bezier.py:1211-1246: vec(outer(a, b)) @ coefMat
    return np.dot(np.outer(a, b).reshape(1, -1), _prod_matrix(a.size - 1, b.size - 1))[0]


def _normsq(x):                              # bezier.py:869-889 -> 1724-1756 ((d/2) quirk kept)
    d, nc = x.shape
    xaug = np.dot(x.T, x).reshape(nc * nc)
    return 0.5 * d * np.dot(xaug, _prod_matrix(nc - 1, nc - 1))


def _elev(c, R):                             # bezier.py:469-495
    return c @ _elev_matrix(c.shape[-1] - 1, R)


def _diff(x, T):                             # bezier.py:497-519 (derivative, then elev(1))
    n = x.shape[-1] - 1
    return _elev((x[..., 1:] - x[..., :-1]) * (n / T), 1)


def temporal_sep(Y, nveh, dim, R, max_sep):  # optimization.py:311-346
    out = []
    for i in range(nveh - 1):
        for j in range(i + 1, nveh):
            dv = Y[i * dim:(i + 1) * dim] - Y[j * dim:(j + 1) * dim]
            out.append(_elev(_normsq(dv), R))
    return np.concatenate(out) - max_sep ** 2


def speed(Y, nveh, dim, R, tf, bound, is_max):   # optimization.py:349-422
    out = [_elev(_normsq(_diff(Y[i * dim:(i + 1) * dim], tf)), R) for i in range(nveh)]
    v = np.concatenate(out)
    return bound ** 2 - v if is_max else v - bound ** 2


def ang_rate(Y, nveh, R, tf, max_rate):      # optimization.py:425-459, 578-611
    out = []
    for i in range(nveh):
        pe = _elev(Y[2 * i:2 * i + 2], R)
        d1 = _diff(pe, tf)
        d2 = _diff(d1, tf)
        num = _mul(d2[1], d1[0]) - _mul(d2[0], d1[1])
        den = _mul(d1[0], d1[0]) + _mul(d1[1], d1[1])
        with np.errstate(all="ignore"):
            out.append(_mul(num, num) / _mul(den, den))
    return max_rate ** 2 - np.concatenate(out)
