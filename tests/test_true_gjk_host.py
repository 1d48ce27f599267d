"""csrc/gjk_true.h -- the textbook GJK hull distance behind the robust entry points (NOT the reference's gjkNew) --
compiled for the HOST and held against a quadratic program over the barycentric weights (SciPy SLSQP).  CPU only:
the same header is what the HIP kernels compile."""
import os

import numpy as np
import pytest
import scipy.optimize as sop

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def qp_distance(p1, p2):
    """min | sum a_i p1_i - sum b_j p2_j | over the two simplices of weights."""
    k1, k2 = len(p1), len(p2)

    def f(z):
        d = z[:k1] @ p1 - z[k1:] @ p2
        return d @ d

    def g(z):
        d = z[:k1] @ p1 - z[k1:] @ p2
        return np.concatenate((2 * p1 @ d, -2 * p2 @ d))
    best = np.inf
    for start in range(3):
        rng = np.random.default_rng(start)
        z0 = np.concatenate((rng.dirichlet(np.ones(k1)), rng.dirichlet(np.ones(k2))))
        cons = [{'type': 'eq', 'fun': lambda z: z[:k1].sum() - 1.0}, {'type': 'eq', 'fun': lambda z: z[k1:].sum() - 1.0}]
        r = sop.minimize(f, z0, jac=g, bounds=[(0, 1)] * (k1 + k2), constraints=cons, method='SLSQP',
                         options={'maxiter': 500, 'ftol': 1e-16})
        best = min(best, np.sqrt(max(r.fun, 0.0)))
    return best


@pytest.mark.parametrize("dim", [2, 3])
def test_true_gjk_equals_the_qp_distance(host_gjk, dim):
    rng = np.random.default_rng(11 + dim)
    n_sep = n_hit = 0
    for trial in range(120):
        k1, k2 = int(rng.integers(1, 12)), int(rng.integers(1, 12))
        p1 = np.zeros((k1, 3)); p2 = np.zeros((k2, 3))
        p1[:, :dim] = rng.normal(0, 1.0, (k1, dim))
        p2[:, :dim] = rng.normal(0, 1.0, (k2, dim)) + rng.normal(0, 2.0, dim)
        r = host_gjk(p1, p2)
        ref = qp_distance(p1, p2)
        assert r["status"] in (0, 2) and r["iters"] <= 40      # 2: stalled at rounding level without the certificate
        if ref > 1e-6:
            n_sep += 1
            assert r["flag"] == 1
            assert abs(r["dist"] - ref) <= 2e-6 * max(1.0, ref), (trial, r, ref)
            assert r["lower"] <= r["dist"] * (1 + 1e-15) and r["lower"] <= ref * (1 + 1e-9)               # a valid lower bound, always
            if r["status"] == 0:
                assert r["dist"] - r["lower"] <= 1e-9 * r["dist"]                                        # the certificate
            assert abs(np.linalg.norm(r["c1"] - r["c2"]) - r["dist"]) <= 1e-12 * max(1.0, ref)
        else:
            n_hit += 1
            assert r["dist"] <= 1e-6
    assert n_sep >= 30 and n_hit >= 10


def test_true_gjk_known_cases(host_gjk):
    sq = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], float)
    r = host_gjk(sq, sq + [3, 0, 0])
    assert r["flag"] == 1 and abs(r["dist"] - 2.0) < 1e-12
    r = host_gjk(sq, sq + [2, 2, 0])                         # corner to corner
    assert abs(r["dist"] - np.sqrt(2.0)) < 1e-12
    r = host_gjk(sq, sq + [0.5, 0.5, 0])                     # overlapping
    assert r["flag"] == 0 and r["dist"] == 0.0
    r = host_gjk([[0.5, 0.5, 2.0]], sq)                      # point above the square's interior: face region
    assert abs(r["dist"] - 2.0) < 1e-12 and np.allclose(r["c2"], [0.5, 0.5, 0.0])
    tet = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], float)
    r = host_gjk([[0.2, 0.2, 0.2]], tet)                     # inside a tetrahedron
    assert r["flag"] == 0
    r = host_gjk([[1.0, 1.0, 1.0]], tet)                     # nearest the slanted face x + y + z = 1
    assert abs(r["dist"] - 2.0 / np.sqrt(3.0)) < 1e-12
    # the demo pair on which gjkNew stops early (gjk/gjkTests.py:23-34): dyn4j's article value
    p1 = np.array([(4, 11, 0), (4, 5, 0), (9, 9, 0)], float)
    p2 = np.array([(8, 6, 0), (10, 2, 0), (13, 1, 0), (15, 6, 0)], float)
    assert abs(host_gjk(p1, p2)["dist"] - qp_distance(p1, p2)) < 1e-9
