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


# ------------------------------------------------------------------------------------------------
# gjkNew, reference-shaped (gjk/gjk.py:230-681): the simplex is a dict keyed 'A'..'D' (+ 'Apts'..'Dpts',
# 'collision') exactly as SURVEY.md Appendix A describes it -- WHICH keys exist selects the case --
# and every step is a handful of NumPy 3-vector operations issued from Python.  The two helpers the
# reference compiles with Numba (`support`, `dot`) are a vectorised scan here (same products and sums
# in the same order, first maximum = the reference's strict `>` from index 0), so that the port is not
# slower than the reference with its JIT.  Checked against the C oracle (flags, support counts,
# distances) in tests/test_oracle_golden.py.
# ------------------------------------------------------------------------------------------------
def _dot3(a, b):                                   # gjk.py:194
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def _far(poly, d):                                 # gjk.py:87-114
    return poly[int(np.argmax(poly[:, 0] * d[0] + poly[:, 1] * d[1] + poly[:, 2] * d[2]))]


_n_support = [0]                                   # supportPts calls of the current gjk_new (instrumentation)


def _new_vertex(s, p1, p2, d):                     # gjk.py:493-501: Minkowski support, always stored as 'A'
    a, b = _far(p1, d), _far(p2, -d)
    s['A'], s['Apts'] = a - b, (a, b)
    _n_support[0] += 1


def _origin_to_segment(A, B):                      # gjk.py:397-437
    if (A == B).all():
        return 0, np.sqrt(_dot3(A, A))
    v = B - A
    t = min(1, max(0, -_dot3(v, A) / _dot3(v, v)))
    c = (1 - t) * A + t * B
    return t, np.sqrt(_dot3(c, c))


def _origin_to_plane(A, B, C):                     # gjk.py:440-477
    N = np.cross(B - A, C - A)
    nn = np.linalg.norm(N)
    u = N / nn
    t = (u[0] * A[0] + u[1] * A[1] + u[2] * A[2]) / (u[0] ** 2 + u[1] ** 2 + u[2] ** 2)
    c = t * u
    al = np.linalg.norm(np.cross(B - c, C - c)) / nn
    be = np.linalg.norm(np.cross(C - c, A - c)) / nn
    return (al, be, 1 - al - be), np.sqrt(_dot3(c, c))


def _move(s, dst, src):
    s[dst], s[dst + 'pts'] = s[src], s[src + 'pts']


def _triangle(s, p1, p2):                          # gjk.py:565-642
    A = s['A']
    A0, AB, AC = -A, s['B'] - A, s['C'] - A
    ABC = np.cross(AB, AC)
    edge_ab = False
    if np.cross(ABC, AC).dot(A0) > 0:
        if AC.dot(A0) > 0:
            d = np.cross(np.cross(AC, A0), AC)
            _move(s, 'B', 'A')
        elif AB.dot(A0) > 0:
            edge_ab = True
        else:
            d = A                                   # (sic) +A, gjk.py:595
            s.clear()
    elif np.cross(AB, ABC).dot(A0) > 0:
        if AB.dot(A0) > 0:
            edge_ab = True
        else:
            d = -A
            s.clear()
    else:
        side = ABC.dot(A0)
        if side == 0:
            s['collision'] = True
            d = np.array((0, 0, 0))
        elif side > 0:
            d = ABC
            _move(s, 'D', 'C'); _move(s, 'C', 'B'); _move(s, 'B', 'A')
        else:
            d = -ABC
            _move(s, 'D', 'B'); _move(s, 'B', 'A')
    if edge_ab:
        d = np.cross(np.cross(AB, A0), AB)
        _move(s, 'C', 'A')
    _new_vertex(s, p1, p2, d)
    return d


def _step(s, p1, p2, d):                           # doSimplex, gjk.py:505-526
    if 'A' not in s:                                # 0 points
        _new_vertex(s, p1, p2, d)
    elif 'B' not in s:                              # 1 point
        _move(s, 'B', 'A')
        d = -d
        _new_vertex(s, p1, p2, d)
    elif 'C' not in s:                              # 2 points
        t, _ = _origin_to_segment(s['A'], s['B'])
        d = -((1 - t) * s['A'] + t * s['B'])
        _move(s, 'C', 'A')
        _new_vertex(s, p1, p2, d)
    elif 'D' not in s:
        d = _triangle(s, p1, p2)
    else:                                           # 4 points, gjk.py:646-681
        A = s['A']
        A0, AB, AC, AD = -A, s['B'] - A, s['C'] - A, s['D'] - A
        if np.cross(AB, AC).dot(A0) > 0:
            s.pop('D')                              # 'Dpts' stays behind (gjk.py:660)
            d = _triangle(s, p1, p2)
        elif np.cross(AC, AD).dot(A0) > 0:
            _move(s, 'B', 'C')
            s['C'], s['Cpts'] = s.pop('D'), s.pop('Dpts')
            d = _triangle(s, p1, p2)
        elif np.cross(AD, AB).dot(A0) > 0:
            _move(s, 'C', 'B')
            s['B'], s['Bpts'] = s.pop('D'), s.pop('Dpts')
            d = _triangle(s, p1, p2)
        else:
            s['collision'] = True
            d = np.array((0, 0, 0))
    return d


def _same(a, b):
    try:
        return bool((a == b).all())
    except AttributeError:                          # tuples of points / the collision flag
        return bool(np.all(a == b)) if not isinstance(b, (bool, int)) else False


def _closest(s, p1, p2, d, md_cap):                # minimumDistance, gjk.py:273-360
    rounds = 0
    while True:
        old = dict(s)
        d = _step(s, p1, p2, d)
        hit = any(_same(s['A'], old[k]) for k in ('A', 'B', 'C', 'D') if k in old)
        if hit:
            s = old
            break
        rounds += 1
        if rounds >= md_cap:
            return None, None, np.nan
    def blend(t, X, Y):
        return ((1 - t) * s[X + 'pts'][0] + t * s[Y + 'pts'][0], (1 - t) * s[X + 'pts'][1] + t * s[Y + 'pts'][1])
    if 'C' in s:
        A = s['A']
        A0, AB, AC = -A, s['B'] - A, s['C'] - A
        ABC = np.cross(AB, AC)
        if np.cross(ABC, AC).dot(A0) >= 0:
            t, dist = _origin_to_segment(A, s['C'])
            q1, q2 = blend(t, 'A', 'C')
        elif np.cross(AB, ABC).dot(A0) >= 0:
            t, dist = _origin_to_segment(A, s['B'])
            q1, q2 = blend(t, 'A', 'B')
        else:
            w, dist = _origin_to_plane(A, s['B'], s['C'])
            q1 = sum(w[i] * (s[k] + s[k + 'pts'][1]) for i, k in enumerate('ABC'))
            q2 = sum(w[i] * (s[k + 'pts'][0] - s[k]) for i, k in enumerate('ABC'))
    elif 'B' in s:
        t, dist = _origin_to_segment(s['A'], s['B'])
        q1, q2 = blend(t, 'A', 'B')
    else:
        dist = np.linalg.norm(s['A'])
        q1, q2 = s['Apts']
    return q1, q2, dist


def gjk_new(poly1, poly2, max_iter=128, md_cap=4096):
    """-> (flag, (p1, p2, dist) | (), n_support); flag as gjk.py:234-237.  md_cap bounds the reference's
    uncapped `while True` (flag 1 with dist NaN when it fires)."""
    s = {}
    d = np.array((1, 0, 0), dtype=float)
    _n_support[0] = 0
    for _ in range(max_iter):
        d = _step(s, poly1, poly2, d)
        if 'collision' in s:
            return 0, (), _n_support[0]
        if s['A'].dot(d) < 0:
            return 1, _closest(s, poly1, poly2, d, md_cap), _n_support[0]
    return -1, (), _n_support[0]
