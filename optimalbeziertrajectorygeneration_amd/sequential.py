"""The reference's sequential swarm planner (Examples/SequentialSwarm.py) on the MI355X path: vehicles are planned
one after the other, each by an SLSQP run whose only constraint is the separation from trajectories that are
already fixed.  Same function names, argument meaning and return values as the example:

    temporalSeparationConstraints(y, nveh, ndim, maxSep)   SequentialSwarm.py:43-70
    nonlcon(x, vidx, traj, nveh, params)                   :19-40
    cost(x, vidx, params)                                  :72-78
    reshape(x, traj, ndim, inipt, finalpt)                 :81-116
    initguess(vidx, params)                                :119-136
    Parameters(nveh, ndim, deg, volume, dsafe)             :139-163   (final points passed in; the example reads a CSV)
    plan(params, ...)                                      :176-192   (the vehicle-after-vehicle loop of `__main__`)

The constraint pairs ONE trajectory with K others and keeps the smallest elevated control point of each pair's squared
distance.  On the device that is `obtg_one_vs_many_min` (include/obtg.h): one lane per (candidate, other trajectory),
minimum kept in the lane, no pair table, and K simply grows by one per planned vehicle on ONE context per
(dimension, degree, elevation).

Pairing.  The example's `reshape` appends the vehicle being planned BEHIND the fixed trajectories while its constraint
takes rows 0..ndim-1 as "the" vehicle, so what the reference evaluates is trajectory 0 against everyone else (of which
only the last row moves with x).  `pairing='reference'` keeps exactly that; `pairing='new_vs_all'` is what the
docstrings describe -- the new vehicle against every fixed one -- and what a planner wants.
"""
import time

import numpy as np
import scipy.optimize as sop

from . import _capi

FD_STEP = 1.4901161193847656e-08   # SciPy '2-point' abs_step

_ctx_cache = {}


def _context(ndim, deg, deg_elev, device=None):
    """One context per (dimension, degree, elevation): the number of trajectories is an argument of every call.
    device None: this process's GPU (_capi.default_device)."""
    if device is None:
        device = _capi.default_device()
    key = (int(ndim), int(deg), int(deg_elev), int(device))
    c = _ctx_cache.get(key)
    if c is None:
        c = _capi.Context(1, key[0], key[1], key[2], device=key[3])
        _ctx_cache[key] = c
    return c


_generic_cache = {}


def _one_vs_many(one, many, ndim, maxSep, degElev):
    """out[B][K].  Degrees without a specialised kernel go through the any-degree separation kernel: a context of K + 1
    "vehicles" (candidate first) whose first K lexicographic pairs are exactly (candidate, other k)."""
    nc = many.shape[-1]
    if (_capi.fast_kernels(ndim, nc - 1) & 1) and degElev <= 512:  # (a specialised kernel exists: obtg_fast_kernels, bern_kernels.hip fast_shape)
        return _context(ndim, nc - 1, degElev).one_vs_many_min(one, many, maxSep)
    one = np.ascontiguousarray(one, dtype=np.float64).reshape(-1, ndim, nc)
    many = np.ascontiguousarray(many, dtype=np.float64).reshape(-1, ndim, nc)
    B, K = one.shape[0], many.shape[0]
    key = (K + 1, int(ndim), nc - 1, int(degElev))
    ctx = _generic_cache.get(key)
    if ctx is None:
        if len(_generic_cache) > 4:
            _generic_cache.pop(next(iter(_generic_cache))).close()
        ctx = _generic_cache[key] = _capi.Context(K + 1, int(ndim), nc - 1, int(degElev), device=_capi.default_device())
    Y = np.empty((B, (K + 1) * ndim, nc))
    Y[:, :ndim] = one
    Y[:, ndim:] = many.reshape(K * ndim, nc)
    return ctx.temporal_sep_min(Y, maxSep, pair_begin=0, pair_count=K)


def temporalSeparationConstraints(y, nveh, ndim, maxSep, degElev=10):
    """Examples/SequentialSwarm.py:43-70 (the elevation, hard-coded to 10 there, is a keyword here): trajectory 0
    (rows 0..ndim-1 of y) against trajectories 1..nveh-1 -> float64[nveh-1]."""
    if nveh <= 1:
        return np.atleast_1d(0.0)                     # SequentialSwarm.py:69-70
    y = np.ascontiguousarray(y, dtype=np.float64)
    return _one_vs_many(y[0:ndim], y[ndim:nveh * ndim], ndim, maxSep, degElev)[0]


def new_vs_all(ynew, traj, ndim, maxSep, degElev=10):
    """The candidate trajectories ynew[B][ndim][deg+1] (or one, [ndim][deg+1]) against every fixed trajectory of
    traj[(K*ndim), deg+1] -> float64[B][K]: the planner's constraint as its docstrings describe it, for a whole
    finite-difference batch of the new vehicle in one launch."""
    traj = np.ascontiguousarray(traj, dtype=np.float64)
    ynew = np.ascontiguousarray(ynew, dtype=np.float64)
    return _one_vs_many(ynew, traj, ndim, maxSep, degElev)


def reshape(x, traj, ndim, inipt, finalpt):
    """SequentialSwarm.py:81-116: x = the interior control points of the vehicle being planned, dimension after
    dimension; returns the fixed trajectories with the new one (end points added) appended as the LAST ndim rows."""
    inner = np.asarray(x, dtype=np.float64).reshape(ndim, -1)
    traj = np.asarray(traj, dtype=np.float64)
    k = traj.shape[0] if traj.size > 0 else 0
    y = np.empty((k + ndim, inner.shape[1] + 2))
    if k:
        y[:k] = traj
    y[k:, 0] = np.ravel(inipt)
    y[k:, 1:-1] = inner
    y[k:, -1] = np.ravel(finalpt)
    return y


def initguess(vidx, params):
    """SequentialSwarm.py:119-136: the interior control points of the straight line between the vehicle's end points."""
    line = np.linspace(params.inipts[vidx, :params.ndim], params.finalpts[vidx, :params.ndim], params.deg + 1, axis=1)
    return np.ascontiguousarray(line[:, 1:-1]).reshape(-1)


def cost(x, vidx, params):
    """SequentialSwarm.py:72-78: the example plans for feasibility only."""
    return 0


def nonlcon(x, vidx, traj, nveh, params, pairing='reference', degElev=10):
    """SequentialSwarm.py:19-40."""
    if pairing == 'reference':
        y = reshape(x, traj, params.ndim, params.inipts[vidx, :], params.finalpts[vidx, :])
        return np.concatenate([temporalSeparationConstraints(y, nveh, params.ndim, params.dsafe, degElev)])
    traj = np.asarray(traj)
    if traj.size == 0:
        return np.atleast_1d(0.0)
    ynew = reshape(x, np.atleast_2d([]), params.ndim, params.inipts[vidx, :], params.finalpts[vidx, :])
    return new_vs_all(ynew, traj, params.ndim, params.dsafe, degElev)[0]


def nonlcon_jac(x, vidx, traj, nveh, params, pairing='reference', degElev=10):
    """SciPy's 2-point Jacobian of `nonlcon` from ONE device call: the n_x + 1 candidates x, x + h e_k of the vehicle
    being planned against the trajectories its rows depend on (all K fixed ones for 'new_vs_all'; trajectory 0 alone
    for the reference's pairing, whose other rows do not move with x).  Entry for entry what approx_derivative builds
    from n_x + 1 calls of `nonlcon`."""
    x = np.asarray(x, dtype=np.float64)
    traj = np.asarray(traj)
    nx = x.size
    if traj.size == 0:
        return np.zeros((1, nx))
    X = np.repeat(x[None], nx + 1, axis=0)
    X[np.arange(1, nx + 1), np.arange(nx)] += FD_STEP
    nc = params.deg + 1
    Yc = np.empty((nx + 1, params.ndim, nc))
    Yc[:, :, 0] = params.inipts[vidx]
    Yc[:, :, -1] = params.finalpts[vidx]
    Yc[:, :, 1:-1] = X.reshape(nx + 1, params.ndim, nc - 2)
    if pairing == 'reference':
        K = traj.shape[0] // params.ndim
        J = np.zeros((K, nx))                   # rows: trajectory 0 vs trajectories 1..K-1 (constant) and vs the new one
        F = new_vs_all(Yc, traj[0:params.ndim], params.ndim, params.dsafe, degElev)[:, 0]
        J[K - 1] = (F[1:] - F[0]) / ((X[np.arange(1, nx + 1), np.arange(nx)]) - x)
        return J
    F = new_vs_all(Yc, traj, params.ndim, params.dsafe, degElev)
    return ((F[1:] - F[0]) / ((X[np.arange(1, nx + 1), np.arange(nx)]) - x)[:, None]).T


class Parameters(object):
    """SequentialSwarm.py:139-163.  Initial points random in the z = 0 face of the control volume (seeded here); final
    points are handed in -- the example reads Examples/HawksLogo_1000pts.csv -- or drawn in the z = volume face."""

    def __init__(self, nveh, ndim, deg, volume, dsafe, finalpts=None, seed=3):
        self.nveh, self.ndim, self.deg, self.volume, self.dsafe = nveh, ndim, deg, volume, dsafe
        rng = np.random.default_rng(seed)
        self.inipts = volume * np.concatenate([rng.random((nveh, ndim - 1)), np.zeros((nveh, 1))], axis=1)
        if finalpts is None:
            finalpts = volume * np.concatenate([rng.random((nveh, ndim - 1)), np.ones((nveh, 1))], axis=1)
        finalpts = np.asarray(finalpts, dtype=np.float64)
        if finalpts.shape[1] == ndim - 1:            # the CSV holds the in-plane coordinates only (:159-161)
            finalpts = np.concatenate((finalpts, volume * np.ones((nveh, 1))), axis=1)
        self.finalpts = np.ascontiguousarray(finalpts[:nveh])


def plan(params, nveh=None, pairing='reference', with_jac=True, degElev=10, maxiter=250, verbose=False, objective=None,
         on_failure=None):
    """The loop of SequentialSwarm.py:176-192: plan vehicle i by SLSQP against the trajectories fixed so far, append
    it, go on.  -> (traj[(nveh*ndim), deg+1], per-vehicle OptimizeResult list, seconds).
    objective: 'feasibility' is the example's constant cost (SequentialSwarm.py:72-74; the default for its own pairing);
    'deviation' = squared distance of the interior control points from the straight-line guess, which gives SLSQP a
    well-posed problem when the new vehicle is tied to EVERY fixed trajectory (the default for 'new_vs_all')."""
    nveh = params.nveh if nveh is None else nveh
    if objective is None:
        objective = 'feasibility' if pairing == 'reference' else 'deviation'
    # on_failure: what joins the fixed trajectories when SLSQP does not converge.  'keep' = whatever point it stopped at
    # (SequentialSwarm.py:186-190 appends res.x unconditionally: the default for its own pairing); 'guess' = the straight
    # line, unless the stopping point violates the constraint less (the default for 'new_vs_all', where a vehicle whose
    # target lies within dsafe of an earlier one has NO feasible trajectory and SLSQP's last iterate can be anywhere)
    if on_failure is None:
        on_failure = 'keep' if pairing == 'reference' else 'guess'
    traj = np.atleast_2d([])
    results = []
    t0 = time.time()
    for i in range(nveh):
        x0 = initguess(i, params)
        cons = {'type': 'ineq', 'fun': lambda x, i=i, traj=traj: nonlcon(x, i, traj, i + 1, params, pairing, degElev)}
        if with_jac:
            cons['jac'] = lambda x, i=i, traj=traj: nonlcon_jac(x, i, traj, i + 1, params, pairing, degElev)
        if objective == 'deviation':
            fun, grad = (lambda x, x0=x0: float(np.dot(x - x0, x - x0))), (lambda x, x0=x0: 2.0 * (x - x0))
        else:
            fun, grad = (lambda x, i=i: cost(x, i, params)), (lambda x: np.zeros_like(x))
        res = sop.minimize(fun, x0, constraints=[cons], method='SLSQP', jac=grad if with_jac else None,
                           options={'maxiter': maxiter, 'disp': False, 'iprint': 0})
        results.append(res)
        xi = res.x if np.all(np.isfinite(res.x)) else x0
        if on_failure == 'guess' and not res.success and xi is not x0:
            if float(np.min(cons['fun'](xi))) < float(np.min(cons['fun'](x0))):
                xi = x0
        traj = reshape(xi, traj, params.ndim, params.inipts[i, :], params.finalpts[i, :])
        if verbose:
            print('vehicle %d: nit %d, feasible margin %+.3e' % (i, res.nit, float(np.min(cons['fun'](xi)))))
    return traj, results, time.time() - t0
