"""Seeded synthetic swarms.

The reference ships no data generator for its large cases; the benchmark and
parity configurations C2..C5 are synthetic and are defined here so that
`bench.py`, the tests and `tests/golden/gen_golden.py` all draw the same inputs.

Conventions: ``N`` vehicles, ``d`` spatial dimension, ``n`` Bernstein degree.
``Y`` is the reference's control-point matrix ``y`` of
``BezOptimization.reshapeVector`` (optimization.py:242-285): shape
``(N*d, n+1)``, vehicle ``i`` owning rows ``i*d .. (i+1)*d-1``.
"""
import numpy as np

FD_STEP = 1.4901161193847656e-08  # SciPy '2-point' abs_step = sqrt(eps)

# name -> (N, d, n, R, n_polygons)
CONFIGS = {
    "C2_file": dict(N=36, d=3, n=5, R=0, n_poly=0),
    "C2": dict(N=8, d=3, n=10, R=0, n_poly=0),
    "C3": dict(N=64, d=2, n=10, R=0, n_poly=8),
    "C4": dict(N=256, d=2, n=15, R=0, n_poly=0),
    # ComplexObstacles.py-style (Examples/ComplexObstacles.py:19-40): the obstacles are Bezier curves
    # ("tracks") handed over as shapeObstacles; spatialSeparationConstraints (optimization.py:109-133)
    # pairs every object with every other one, so the hull sweep covers C(64+32, 2) = 4560 pairs.
    "C5": dict(N=64, d=2, n=10, R=100, n_poly=0, n_curve_obs=32),
}


def swarm_points(N, d, seed=1234):
    """Initial / final points uniform in [0, 100]^d."""
    rng = np.random.default_rng(seed)
    init = rng.uniform(0.0, 100.0, size=(N, d))
    final = rng.uniform(0.0, 100.0, size=(N, d))
    return init, final


def swarm_control_points(N, d, n, seed=1234, noise=2.0):
    """Y[(N*d), n+1]: straight line init->final, N(0, noise^2) on interior points."""
    rng = np.random.default_rng(seed)
    init = rng.uniform(0.0, 100.0, size=(N, d))
    final = rng.uniform(0.0, 100.0, size=(N, d))
    s = np.linspace(0.0, 1.0, n + 1)
    Y = np.empty((N * d, n + 1))
    for i in range(N):
        for k in range(d):
            Y[i * d + k] = init[i, k] + (final[i, k] - init[i, k]) * s
    Y[:, 1:-1] += rng.normal(0.0, noise, size=(N * d, n - 1))
    return Y


def polygon_obstacles(n_poly, seed=1234):
    """List of (K,3) float64 vertex arrays, 4..8 vertices, z = 0."""
    rng = np.random.default_rng(seed + 7919)
    polys = []
    for _ in range(n_poly):
        c = rng.uniform(10.0, 90.0, size=2)
        K = int(rng.integers(4, 9))
        ang = np.sort(rng.uniform(0.0, 2.0 * np.pi, size=K))
        rad = rng.uniform(2.0, 6.0, size=K)
        P = np.zeros((K, 3))
        P[:, 0] = c[0] + rad * np.cos(ang)
        P[:, 1] = c[1] + rad * np.sin(ang)
        polys.append(P)
    return polys


def curve_obstacles(n_curves, d, n, seed=1234):
    """Curve ("track") obstacles of the C5 configuration: degree-n curves from the same generator
    as the vehicles, own seed stream.  -> Y_obs[(n_curves*d), n+1]."""
    return swarm_control_points(n_curves, d, n, seed=seed + 15485863)


def all_pairs(n_obj):
    """Every unordered pair i<j of n_obj objects, lexicographic: the pair loop of
    spatialSeparationConstraints (optimization.py:127-130) over vehicles AND obstacles."""
    a, b = np.triu_indices(n_obj, 1)
    return a.astype(np.int32), b.astype(np.int32)


def config_hull_sweep(name, seed=1234):
    """The gjkNew hull sweep of a named configuration: static objects registered behind the
    vehicles and the pair list.  -> (static_polys [list of (K,3)], pair_a, pair_b)."""
    cfg = CONFIGS[name]
    N, d, n = cfg["N"], cfg["d"], cfg["n"]
    M = cfg.get("n_curve_obs", 0)
    if M:
        statics = hulls_from_Y(curve_obstacles(M, d, n, seed=seed), d)
        pa, pb = all_pairs(N + M)
        return statics, pa, pb
    Mp = cfg.get("n_poly", 0)
    statics = polygon_obstacles(Mp, seed=seed)
    pa, pb = swarm_pairs(N, Mp)
    return statics, pa, pb


def fd_batch(Y, B=None, h=FD_STEP):
    """Finite-difference batch of control-point matrices.

    Row 0 is ``Y``; row k (k>=1) perturbs the k-th *free* control point
    (interior columns, row-major over (row, column) exactly like
    ``x.reshape(numRows, numCols)`` in optimization.py:283) by ``h``.
    ``B`` defaults to n_x + 1 with n_x = rows * (n+1-2).
    """
    rows, cols = Y.shape
    ncol_free = cols - 2
    n_x = rows * ncol_free
    if B is None:
        B = n_x + 1
    out = np.repeat(Y[None], B, axis=0)
    for b in range(1, B):
        k = (b - 1) % n_x
        r, c = divmod(k, ncol_free)
        out[b, r, 1 + c] += h
    return out


def hulls_from_Y(Y, d):
    """Control polygons as (K,3) point sets (2-D padded with z=0): the polys
    `_minDist` hands to gjkNew (bezier.py:1289-1308)."""
    rows, cols = Y.shape
    N = rows // d
    polys = []
    for i in range(N):
        P = np.zeros((cols, 3))
        P[:, :d] = Y[i * d:(i + 1) * d].T
        polys.append(P)
    return polys


def pack_polys(polys):
    """-> pts[(sum K),3] float64, off[n_poly+1] int32."""
    off = np.zeros(len(polys) + 1, dtype=np.int32)
    for i, p in enumerate(polys):
        off[i + 1] = off[i] + p.shape[0]
    pts = np.ascontiguousarray(np.vstack(polys), dtype=np.float64) if polys else np.zeros((0, 3))
    return pts, off


def swarm_pairs(N, M):
    """Hull pair list of the C3-style sweep: all vehicle<->vehicle pairs (i<j,
    lexicographic) followed by all vehicle<->obstacle pairs (obstacle ids N..N+M-1)."""
    a, b = [], []
    for i in range(N - 1):
        for j in range(i + 1, N):
            a.append(i)
            b.append(j)
    for i in range(N):
        for k in range(M):
            a.append(i)
            b.append(N + k)
    return np.asarray(a, dtype=np.int32), np.asarray(b, dtype=np.int32)
