"""`optimization.BezOptimization` look-alike: same constructor, attributes and callback
signatures as the reference (optimization.py:20-308), with every constraint / cost
evaluation dispatched to the MI355X through libobtg_hip.so.

A driver script changes only its import lines:

    import optimalbeziertrajectorygeneration_amd.bezier as bez
    from optimalbeziertrajectorygeneration_amd.optimization import BezOptimization

and hands the same closures to `scipy.optimize.minimize(method='SLSQP')`.  Beyond the
reference, each constraint has a `...Jacobian` provider that evaluates SciPy's whole
2-point finite-difference batch (n_x + 1 rows) in ONE launch; pass it as the 'jac' entry of
the constraint dict to remove the n_x + 1 serial callbacks per iteration.
"""
import numpy as np

from . import _capi
from . import bezier as bez

DEG_ELEV = 0   # module constant read at call time, like optimization.py:17

FD_STEP = 1.4901161193847656e-08   # SciPy '2-point' abs_step (sqrt of machine epsilon)


# ---------------------------------------------------------------------------------------------
# module-level evaluators with the reference's private signatures (optimization.py:311-459).
# Drivers copy / call these directly (Examples/Example1_DubinsCarTimeOptimal.py:19-52 re-implements
# the first one with a `degElev` argument), so they exist here too; contexts are cached per shape.
# ---------------------------------------------------------------------------------------------
import collections
import os

_CTX_CACHE_MAX = 8              # device contexts kept; the least recently used one is closed beyond that
_ctx_cache = collections.OrderedDict()


def _shape_ctx(nVeh, dim, deg, R, device=None):
    """A device context per (shape, DEG_ELEV): least-recently-used eviction, so that a driver sweeping degElev or the
    degree (Examples/Example1_DubinsCarTimeOptimal.py:19-52 takes degElev as an argument) does not pile up contexts.
    device None: this process's GPU (_capi.default_device: OBTG_DEVICE, else the launcher's LOCAL_RANK, else 0)."""
    if device is None:
        device = _capi.default_device()
    key = (int(nVeh), int(dim), int(deg), int(R), int(device))
    c = _ctx_cache.get(key)
    if c is None:
        c = _capi.Context(key[0], key[1], key[2], key[3], device=key[4])
        _ctx_cache[key] = c
        while len(_ctx_cache) > _CTX_CACHE_MAX:
            _, old = _ctx_cache.popitem(last=False)
            old.close()
    else:
        _ctx_cache.move_to_end(key)
    return c


# The private separation evaluator under SciPy's finite differences (round 6).  Examples/Example1_DubinsCarTimeOptimal.py:124-125
# hands SLSQP `lambda x: _temporalSeparationConstraints(bezopt.reshapeVector(x), ...)`: the function sees y, not x, and SciPy
# differences the lambda with n_x one-row calls per iteration.  As BezOptimization._serve does for the class closures: a call
# whose y differs from the last base in ONE element by exactly SciPy's step is taken for a row of that sweep -- the rows for
# every element of y are evaluated in one launch, kept, and this and the following calls are answered from them (a variable
# that moves several elements -- tf with prescribed speeds -- and any other point are evaluated directly and become the new
# base).  Values are those of the one-row call (the batch kernels' rows are its bits: tests/test_gpu_dropin.py).
_ysep_state = {}


def _temporalSeparationConstraints(y, nVeh, dim, maxSep, degElev=None):
    if nVeh <= 1:
        return None
    y = np.ascontiguousarray(y, dtype=np.float64)
    R = DEG_ELEV if degElev is None else degElev
    ctx = _shape_ctx(nVeh, dim, y.shape[1] - 1, R)
    if os.environ.get("OBTG_FD_BATCHING", "1") == "0":
        return ctx.temporal_sep(y, maxSep)[0]
    key = (int(nVeh), int(dim), y.shape, int(R), float(maxSep))
    st = _ysep_state.get(key)
    if st is not None and st['ctx'] is ctx:
        d = np.flatnonzero(y.ravel() != st['y0'].ravel())
        if d.size == 0:
            return st['base'].copy()
        if d.size == 1 and y.ravel()[d[0]] == st['y0'].ravel()[d[0]] + FD_STEP:
            if st['rows'] is None:
                E = y.size
                limit = float(os.environ.get("OBTG_FD_BATCH_MB", "512")) * 2.0 ** 20
                if 8.0 * E * st['base'].size > limit:
                    st['rows'] = False
                else:
                    Yb = np.repeat(st['y0'][None], E, axis=0)
                    Yb.reshape(E, E)[np.arange(E), np.arange(E)] += FD_STEP
                    st['rows'] = ctx.temporal_sep(Yb, maxSep)
            if st['rows'] is not False:
                return st['rows'][d[0]].copy()
            return ctx.temporal_sep(y, maxSep)[0]
    v = ctx.temporal_sep(y, maxSep)[0]
    if len(_ysep_state) > 8:
        _ysep_state.clear()
    _ysep_state[key] = {'ctx': ctx, 'y0': y.copy(), 'base': v.copy(), 'rows': None}
    return v


def _minSpeedConstraints(y, nVeh, dim, tf, minSpeed):
    y = np.ascontiguousarray(y, dtype=np.float64)
    return _shape_ctx(nVeh, dim, y.shape[1] - 1, DEG_ELEV).speed(y, tf, minSpeed, False)[0]


def _maxSpeedConstraints(y, nVeh, dim, tf, maxSpeed):
    y = np.ascontiguousarray(y, dtype=np.float64)
    return _shape_ctx(nVeh, dim, y.shape[1] - 1, DEG_ELEV).speed(y, tf, maxSpeed, True)[0]


def _maxAngularRateConstraints(y, nVeh, dim, tf, maxAngRate):
    if dim != 2:
        raise ValueError('The input curve must be two dimensional,\n'
                         'instead it is {} dimensional'.format(dim))
    y = np.ascontiguousarray(y, dtype=np.float64)
    return _shape_ctx(nVeh, dim, y.shape[1] - 1, DEG_ELEV).ang_rate(y, tf, maxAngRate)[0]


def _euclideanObjective(y, nVeh, dim):
    """optimization.py:463-490: summed distances between neighbouring control points, every vehicle (obtg_euclidean_obj)."""
    y = np.ascontiguousarray(y, dtype=np.float64)
    return float(_shape_ctx(nVeh, dim, y.shape[1] - 1, 0).euclidean_obj(y)[0])


def _minAccelObjective(y, nVeh, dim, tf):
    """optimization.py:503-520: sum of the elevated control points of |acceleration|^2 (obtg_accel_obj)."""
    y = np.ascontiguousarray(y, dtype=np.float64)
    return float(_shape_ctx(nVeh, dim, y.shape[1] - 1, DEG_ELEV).deriv_energy_obj(y, tf, 2)[0])


def _minJerkObjective(y, nVeh, dim, tf):
    """optimization.py:523-540: the same for the jerk (obtg_jerk_obj)."""
    y = np.ascontiguousarray(y, dtype=np.float64)
    return float(_shape_ctx(nVeh, dim, y.shape[1] - 1, DEG_ELEV).deriv_energy_obj(y, tf, 3)[0])


def _angularRateSqr(bezTraj):
    """optimization.py:578-611: the squared angular rate of ONE planar trajectory as a rational curve -- control points
    num / den (inf / nan kept), weights den = (|v|^2)^2.  The quotient is obtg_ang_rate's (bound 0: the kernel returns
    0 - num/den), the weights two device products of the speed curve; the trajectory is taken at the degree it has."""
    if bezTraj.dim != 2:
        raise ValueError('The input curve must be two dimensional,\n'
                         'instead it is {} dimensional'.format(bezTraj.dim))
    cpts = np.ascontiguousarray(bezTraj.cpts, dtype=np.float64)
    ctx = _capi.Context(1, 2, bezTraj.deg, 0)
    try:
        quotient = 0.0 - ctx.ang_rate(cpts, bezTraj.tf - bezTraj.t0, 0.0)[0]
    finally:
        ctx.close()
    speed2 = bezTraj.diff().normSquare()              # (d / 2) |v|^2 with d = 2
    return bez.RationalBezier(quotient[None, :], (speed2 * speed2).cpts)


# (key of BezOptimization.model, constructor keyword, container) -- optimization.py:49-63 defines the keys
_MODEL_FIELDS = (
    ('numVeh', 'numVeh', None), ('dim', 'dimension', None), ('deg', 'degree', None), ('minGoal', 'minimizeGoal', None),
    ('maxSep', 'maxSep', None), ('minSpeed', 'minSpeed', None), ('maxSpeed', 'maxSpeed', None),
    ('maxAngRate', 'maxAngRate', None),
    ('initPoints', 'initPoints', np.atleast_2d), ('finalPoints', 'finalPoints', np.atleast_2d),
    ('initSpeeds', 'initSpeeds', np.atleast_1d), ('finalSpeeds', 'finalSpeeds', np.atleast_1d),
    ('initAngs', 'initAngs', np.atleast_1d), ('finalAngs', 'finalAngs', np.atleast_1d),
    ('tf', 'tf', None),
)


def _raise_first_md(status):
    """The exception of the first pair, in list order, whose search did not end (bezier._raise_md): what the reference
    would hit first in its pair loop (optimization.py:127-131)."""
    bad = np.nonzero(np.asarray(status) != _capi.MD_OK)[0]
    if bad.size:
        bez._raise_md(int(status[bad[0]]))


def _spatial_jac_plan(Y, numVeh, dim, obstacle_curves):
    """The curve and pair lists of ONE `obtg_min_dist` call that yields the 2-point Jacobian of
    spatialSeparationConstraints (optimization.py:109-133): the base evaluation's C(n, 2) pairs of the n = numVeh +
    obstacles curves of row 0 of Y, then, for every finite-difference row k + 1, the pairs that contain a vehicle whose
    control points differ from row 0's, with that vehicle's perturbed curve appended to the curve list.
    Y: [n_x + 1][numVeh * dim][deg + 1]; obstacle_curves: padded [3][deg + 1] arrays.
    Returns (stack [n_curves][3][deg + 1], pa, pb, P, col, row, pos): entry e of the call's extra pairs is base pair
    row[e] re-evaluated for variable col[e] at position pos[e] of the pair list.  Pairs of a column come in base-pair
    order, columns in order (array form of the loops it replaced: tests/test_host_logic.py holds it to them)."""
    nx = Y.shape[0] - 1
    K = Y.shape[2]
    n = numVeh + len(obstacle_curves)
    Yv = Y.reshape(nx + 1, numVeh, dim, K)
    base = np.zeros((n, 3, K))
    base[:numVeh, :dim, :] = Yv[0]
    for o, c in enumerate(obstacle_curves):
        base[numVeh + o] = c
    changed = np.any(Yv[1:] != Yv[0], axis=(2, 3))              # [n_x][numVeh]
    ck, cv = np.nonzero(changed)                                # (column, vehicle), columns ascending, vehicles ascending
    extra = np.zeros((ck.size, 3, K))
    extra[:, :dim, :] = Yv[ck + 1, cv]
    # The pair lists depend on WHICH vehicles every column moves, not on the values: an SLSQP run asks for the same pattern at
    # every iterate, so the lists of the last pattern are kept (round 6: building them was 16 of the provider's 35 ms at C5 size).
    memo = _spatial_jac_plan.memo
    pattern = (n, numVeh, changed.shape, changed.tobytes())
    if memo.get('pattern') == pattern:
        pa, pb, P, col, row, pos = memo['lists']
        return np.concatenate((base, extra)), pa, pb, P, col, row, pos
    pa0, pb0 = np.triu_indices(n, 1)                            # i < j, lexicographic: the reference's pair loop
    P = pa0.size
    new_id = np.full((nx, n), -1, dtype=np.int64)               # curve index of vehicle v's perturbed copy in column k
    new_id[ck, cv] = n + np.arange(ck.size)
    # per column, the base pairs touched: those with an end among the column's changed vehicles.  A column that moves ONE
    # vehicle (all but a trailing tf) touches that vehicle's n - 1 pairs; the others go through a mask over the pair list.
    nchg = changed.sum(axis=1)
    pairs_of = np.empty((numVeh, n - 1), dtype=np.int64)        # base pairs containing vehicle v, ascending
    for v in range(numVeh):
        pairs_of[v] = np.nonzero((pa0 == v) | (pb0 == v))[0]
    k1 = np.nonzero(nchg == 1)[0]
    col = [np.repeat(k1, n - 1)]
    row = [pairs_of[np.argmax(changed[k1], axis=1)].ravel()] if k1.size else [np.zeros(0, dtype=np.int64)]
    for k in np.nonzero(nchg > 1)[0]:
        m = np.zeros(n, dtype=bool)
        m[:numVeh] = changed[k]
        q = np.nonzero(m[pa0] | m[pb0])[0]
        col.append(np.full(q.size, k))
        row.append(q)
    col, row = np.concatenate(col).astype(np.int64), np.concatenate(row)
    order = np.lexsort((row, col))                              # column-major, base-pair order within a column
    col, row = col[order], row[order]
    ia, ib = pa0[row], pb0[row]
    na, nb = new_id[col, ia], new_id[col, ib]
    pa = np.concatenate((pa0, np.where(na >= 0, na, ia))).astype(np.int32)
    pb = np.concatenate((pb0, np.where(nb >= 0, nb, ib))).astype(np.int32)
    pos = P + np.arange(col.size)
    memo['pattern'], memo['lists'] = pattern, (pa, pb, P, col, row, pos)
    return np.concatenate((base, extra)), pa, pb, P, col, row, pos


_spatial_jac_plan.memo = {}


class BezOptimization(object):
    def __init__(self,
                 numVeh=1,
                 dimension=1,
                 degree=5,
                 minimizeGoal='Euclidean',
                 maxSep=0.9,
                 minSpeed=0,
                 maxSpeed=1e6,
                 maxAngRate=1e6,
                 initPoints=None,
                 finalPoints=None,
                 initSpeeds=None,
                 finalSpeeds=None,
                 initAngs=None,
                 finalAngs=None,
                 tf=1.0,
                 pointObstacles=None,
                 shapeObstacles=None,
                 device=None,
                 separationRows='all',
                 angRateOrder='fast',
                 activeRows=2,
                 fdBatching=True):
        """Beyond the reference's keywords: `device` (HIP ordinal; None: this process's, see _capi.default_device) and `separationRows` --
        'all': temporalSeparationConstraints returns every elevated control point of every pair, as the
        reference does (optimization.py:337); 'min': one row per pair, the smallest of them -- the
        `dv.normSquare().min()` form the reference leaves commented at optimization.py:338 and uses in
        Examples/SequentialSwarm.py:65 -- which hands SLSQP 2n+R+1 times fewer rows (SURVEY.md 8(f) item 4:
        its dense least-squares step is what dominates an iteration once the callbacks are fast)."""
        # 'active' (round 5; SURVEY.md 8(f) item 4 as worded: "only active / near-active constraint rows"): per pair its
        # `activeRows` SMALLEST elevated control points, in control-point order (1..4; obtg_temporal_sep_active) -- a fixed number of rows, so
        # SLSQP's constraint count is constant, but more than the single piecewise-smooth minimum that makes it stall.
        if separationRows not in ('all', 'min', 'active'):
            raise ValueError("separationRows must be 'all', 'min' or 'active', not {!r}".format(separationRows))
        if separationRows == 'active' and not 1 <= int(activeRows) <= 4:
            raise ValueError("activeRows must be 1..4, not {!r}".format(activeRows))
        self.activeRows = int(activeRows)
        # DEG_ELEV > 0 only: 'fast' forms the angular rate's products at degree 4n and elevates them (0.2 ms at C5);
        # 'elevate_first' elevates the position first, the sequence of optimization.py:597 (1.8 ms); 'exact' = 'fast' plus a
        # double-double recompute of the rows of vehicles that nearly stop.  Until round 6 'elevate_first' was called
        # 'reference' (still accepted): it follows the reference's SEQUENCE of operations, but where a vehicle nearly stops it
        # is no closer to the reference's values than the other orders (nearstop.npz's worst vehicle: 4.6e-9 against 3.4e-9
        # for 'fast' and 3.0e-9 for 'exact' -- the reference itself is 3.0e-9 from the exact rational value there, and its
        # own sums run through OpenBLAS in an order no restatement short of that library's kernels reproduces): the name
        # promised what no float64 order delivers (tests/test_gpu_parity.py::test_near_stop_angular_rate_on_device)
        if angRateOrder == 'reference':
            angRateOrder = 'elevate_first'
        if angRateOrder not in ('fast', 'elevate_first', 'exact'):
            raise ValueError("angRateOrder must be 'fast', 'elevate_first' or 'exact', not {!r}".format(angRateOrder))
        self.angRateOrder = angRateOrder
        self.pointObstacles = pointObstacles
        self.shapeObstacles = shapeObstacles
        self._device = _capi.default_device() if device is None else int(device)
        self.separationRows = separationRows
        # SciPy's own finite differences, served from one batched evaluation (see _serve); OBTG_FD_BATCHING=0 turns it off everywhere
        self.fdBatching = bool(fdBatching) and os.environ.get("OBTG_FD_BATCHING", "1") != "0"
        self.fdBatchingStats = {'batches': 0, 'served': 0, 'direct': 0}
        self._fd_state = None

        given = locals()
        # `model` keeps the reference's keys (drivers read and edit them, Examples/*.py); what each holds is decided by
        # _MODEL_FIELDS: scalars as given, per-vehicle data as arrays with a vehicle axis
        self.model = {key: (given[kw] if shape is None else shape(given[kw])) for key, kw, shape in _MODEL_FIELDS}
        # free control points per coordinate row: the end points, and with prescribed speeds their neighbours, are
        # not variables (two columns each)
        self._numCols = degree + 1 - 2 * sum(given[kw] is not None for kw in ('initPoints', 'initSpeeds'))
        self._ctxs = {}

    # ------------------------------------------------------------------ device contexts
    def _ctx(self, with_point_obs):
        """Context for the current DEG_ELEV (created on first use; the reference's counterpart
        is the lazily filled class-level matrix caches, bezier.py:48-52)."""
        key = bool(with_point_obs)
        c = self._ctxs.get(key)
        if c is None:
            obs = self.pointObstacles if with_point_obs else None
            c = _capi.Context(self.model['numVeh'], self.model['dim'], self.model['deg'], int(DEG_ELEV),
                              point_obs=obs, device=self._device)
            c.set_ang_rate_order({'fast': 0, 'elevate_first': 1, 'exact': 2}[self.angRateOrder])
            self._ctxs[key] = c
        if c.deg_elev != int(DEG_ELEV):
            c.set_deg_elev(int(DEG_ELEV))
        return c

    def _ctx_one(self):
        """A one-vehicle context of the same degree: the per-vehicle families evaluated on a compact
        batch of single vehicles (structured Jacobians)."""
        c = self._ctxs.get('one')
        if c is None:
            c = _capi.Context(1, self.model['dim'], self.model['deg'], int(DEG_ELEV), device=self._device)
            c.set_ang_rate_order({'fast': 0, 'elevate_first': 1, 'exact': 2}[self.angRateOrder])
            self._ctxs['one'] = c
        if c.deg_elev != int(DEG_ELEV):
            c.set_deg_elev(int(DEG_ELEV))
        return c

    @property
    def angRateOrderInEffect(self):
        """'fast' / 'elevate_first' / 'exact': the order the angular rate of this problem REALLY runs in at the current DEG_ELEV
        (the constructor's `angRateOrder` is a request: DEG_ELEV = 0 has one order, and degrees or elevations without a
        products-then-elevation kernel run in the reference's order, without the double-double pass)."""
        return ('fast', 'elevate_first', 'exact')[self._ctx(False).ang_rate_order_in_effect()]

    def _active_k(self):
        """rows per pair of separationRows='active': never more than a pair has control points"""
        return min(self.activeRows, 2 * self.model['deg'] + int(DEG_ELEV) + 1)

    def _timeopt(self):
        return self.model['minGoal'].lower() == 'timeopt'

    def _tf_of(self, x):
        return x[-1] if self._timeopt() else self.model['tf']

    # ------------------------------------------------------------------ objective
    @property
    def objectiveFunction(self):
        minGoal = self.model['minGoal'].lower()
        objectivesDict = {'euclidean': self.euclideanObjective,
                          'timeopt': lambda x: x[-1],
                          'accel': self.accelObjective,
                          'jerk': self.jerkObjective,
                          }
        try:
            return objectivesDict[minGoal]
        except KeyError:
            err = ('The provided minimize goal, {}, is not a valid goal. '
                   'The available minimize goals are:\n{}'
                   ).format(minGoal, objectivesDict.keys())
            raise ValueError(err)

    def euclideanObjective(self, x):
        return float(self._serve('obj_euclidean', x, lambda x_: self._ctx(False).euclidean_obj(self.reshapeVector(x_))[:1])[0])

    def accelObjective(self, x):
        return float(self._serve('obj_accel', x, lambda x_: self._ctx(False).deriv_energy_obj(self.reshapeVector(x_), self.model['tf'], 2)[:1])[0])

    def jerkObjective(self, x):
        return float(self._serve('obj_jerk', x, lambda x_: self._ctx(False).deriv_energy_obj(self.reshapeVector(x_), self.model['tf'], 3)[:1])[0])

    def objectiveGradient(self, x):
        """Gradient of `objectiveFunction` the way SciPy would difference it (2-point, abs_step = sqrt(eps)), but from ONE
        batched evaluation of the n_x + 1 rows instead of n_x + 1 callbacks: pass it as `jac=` to `minimize`.  (With
        the constraint Jacobians supplied, the objective's finite differences are what is left of the per-iteration
        callback count: 17 983 of them in examples/example2_swarm_3d.py with 8 vehicles.)"""
        x = np.asarray(x, dtype=float)
        goal = self.model['minGoal'].lower()
        if goal == 'timeopt':
            g = np.zeros(x.size)
            g[-1] = 1.0
            return g
        X, dx = self._fd_rows(x)
        Y = self.reshapeVectors(X)
        c = self._ctx(False)
        if goal == 'euclidean':
            F = c.euclidean_obj(Y)
        elif goal in ('accel', 'jerk'):
            F = c.deriv_energy_obj(Y, self.model['tf'], 2 if goal == 'accel' else 3)
        else:
            self.objectiveFunction          # raises the reference's ValueError for an unknown goal
        return (F[1:] - F[0]) / dx

    # ------------------------------------------------------------------ constraints
    @property
    def temporalSeparationConstraints(self):
        # (pointObstacles is read when the closure is CALLED, here as in _fd_values: a driver that sets the obstacles after
        # taking the closure gets the served rows and the one-row calls from the same context)
        def wrapper(x):
            with_obs = self.pointObstacles is not None
            if self.model['numVeh'] + (len(self.pointObstacles) if with_obs else 0) <= 1:
                return None                      # optimization.py:345-346
            return self._serve('tsep', x, direct)

        def direct(x):
            with_obs = self.pointObstacles is not None
            y = self.reshapeVector(x)
            if self.separationRows == 'min':     # per-pair minimum, reduced on the device (obtg_temporal_sep_min)
                return self._ctx(with_obs).temporal_sep_min(y, self.model['maxSep'])[0]
            if self.separationRows == 'active':  # per pair its k smallest control points, selected on the device
                return self._ctx(with_obs).temporal_sep_active(y, self.model['maxSep'], self._active_k())[0]
            return self._ctx(with_obs).temporal_sep(y, self.model['maxSep'])[0]
        return wrapper

    @property
    def minSpeedConstraints(self):
        def direct(x):
            y = self.reshapeVector(x)
            return self._ctx(False).speed(y, self._tf_of(x), self.model['minSpeed'], False)[0]
        return lambda x: self._serve('vmin', x, direct)

    @property
    def maxSpeedConstraints(self):
        def direct(x):
            y = self.reshapeVector(x)
            return self._ctx(False).speed(y, self._tf_of(x), self.model['maxSpeed'], True)[0]
        return lambda x: self._serve('vmax', x, direct)

    @property
    def maxAngularRateConstraints(self):
        def wrapper(x):
            if self.model['dim'] != 2:
                msg = ('The input curve must be two dimensional,\n'
                       'instead it is {} dimensional'.format(self.model['dim']))
                raise ValueError(msg)            # optimization.py:590-593
            y = self.reshapeVector(x)
            tf = self._tf_of(x)
            if tf <= 0:
                # an SLSQP step to tf <= 0 (time-optimal drivers without a lower bound on tf): the reference's curve
                # arithmetic returns None for an empty span (bezier.py:340-343, 365-368) and optimization.py:603-604
                # multiplies it -- the driver dies with this TypeError.  Same exception, same text, no device call.
                raise TypeError("unsupported operand type(s) for *: 'NoneType' and 'NoneType'")
            return self._serve('ang', x, lambda x_: self._ctx(False).ang_rate(self.reshapeVector(x_), self._tf_of(x_), self.model['maxAngRate'])[0])
        return wrapper

    def spatialSeparationConstraints(self, x, robust=False):
        """All-pairs minDist over vehicles AND shape obstacles (optimization.py:109-133);
        returns shape (P, 3): (dist, t1, t2) - maxSep, as the reference does.
        robust=True: true minimum distances (obtg_min_dist_robust) instead of the reference's `_minDist`;
        pairs whose search budget runs out (curves coinciding over a stretch) report their best upper bound."""
        return self._serve('spatial_robust' if robust else 'spatial', x, lambda x_: self._spatial_direct(x_, robust))

    def _spatial_direct(self, x, robust):
        numVeh, dim, maxSep = self.model['numVeh'], self.model['dim'], self.model['maxSep']
        y = self.reshapeVector(x)
        curves = [bez.Bezier(y[i * dim:(i + 1) * dim, :]) for i in range(numVeh)] + list(self.shapeObstacles)
        n = len(curves)
        stack = np.stack([c._padded() for c in curves])
        pa, pb = np.triu_indices(n, 1)                              # i < j, lexicographic: the reference's pair loop
        if robust:
            return _capi.scratch_context().min_dist_robust(stack, pa, pb, eps=1e-9, max_nodes=400000)['res'] - maxSep
        r = _capi.scratch_context().min_dist(stack, pa, pb, eps=1e-9, max_depth=128, max_nodes=4000000)
        _raise_first_md(r['status'])
        return r['res'] - maxSep

    def _spatial_fd_values(self, x, robust):
        """F[k + 1] = spatialSeparationConstraints(x + h e_k), F[0] at x, from ONE device call (spatialSeparationJacobian's plan:
        the base pairs and, per variable, the pairs its vehicle touches); None when a search of the call does not end -- the
        closure's own calls then say which, as they always did."""
        numVeh, dim, maxSep = self.model['numVeh'], self.model['dim'], self.model['maxSep']
        X, _ = self._fd_rows(x)
        Y = self.reshapeVectors(X)
        obstacles = list(self.shapeObstacles) if self.shapeObstacles is not None else []
        stack, pa, pb, P, t_col, t_row, t_pos = _spatial_jac_plan(Y, numVeh, dim, [c._padded() for c in obstacles])
        if robust:
            res = _capi.scratch_context().min_dist_robust(stack, pa, pb, eps=1e-9, max_nodes=400000)['res']
        else:
            r = _capi.scratch_context().min_dist(stack, pa, pb, eps=1e-9, max_depth=128, max_nodes=4000000)
            if np.any(r['status'] != _capi.MD_OK):
                return None
            res = r['res']
        F = np.repeat((res[:P] - maxSep)[None], X.shape[0], axis=0)          # [n_x + 1][P][3]
        F[t_col + 1, t_row] = res[t_pos] - maxSep
        return F

    def spatialSeparationJacobian(self, x, robust=False, column=None, on_cap='raise'):
        """SciPy's 2-point Jacobian of `spatialSeparationConstraints` from ONE device call.  The reference hands the
        constraint to SLSQP as it is (Examples/ComplexObstacles.py:49-63), so SciPy evaluates it n_x + 1 times, each an
        all-pairs `_minDist` sweep.  A variable of vehicle v moves only the pairs that contain v: the call carries the
        base evaluation's C(N+M, 2) pairs plus, per variable, the N+M-1 pairs of its perturbed vehicle (a trailing tf
        that moves the speed columns of every vehicle takes all their pairs) as extra curves of the same
        `obtg_min_dist` launch.  Entry for entry what n_x + 1 calls of the closure give: shape (3 P, n_x), rows in the
        order of `spatialSeparationConstraints(x).ravel()`; column=0 keeps the distance rows only, shape (P, n_x) --
        the 1-D constraint a driver would hand to SLSQP.  Statuses are treated as by the closure: a pair on which the
        reference's search does not end (depth / node caps; the reference itself recurses until Python gives up) raises;
        on_cap='nan' marks the entries of such pairs NaN instead (max_depth 128, 4 000 000 nodes per pair as in the
        closure)."""
        numVeh, dim, maxSep = self.model['numVeh'], self.model['dim'], self.model['maxSep']
        X, dx = self._fd_rows(x)
        Y = self.reshapeVectors(X)                                  # [n_x + 1][numVeh*dim][deg+1]
        nx = X.shape[1]
        obstacles = list(self.shapeObstacles) if self.shapeObstacles is not None else []
        stack, pa, pb, P, t_col, t_row, t_pos = _spatial_jac_plan(Y, numVeh, dim, [c._padded() for c in obstacles])
        if robust:
            res = _capi.scratch_context().min_dist_robust(stack, pa, pb, eps=1e-9, max_nodes=400000)['res']
        else:
            r = _capi.scratch_context().min_dist(stack, pa, pb, eps=1e-9, max_depth=128, max_nodes=4000000)
            res = r['res']
            if on_cap == 'nan':
                res = np.where((r['status'] != 0)[:, None], np.nan, res)
            else:
                _raise_first_md(r['status'])
        F0 = res[:P] - maxSep
        if column is not None:           # the one column a driver hands to SLSQP: its (P, n_x) matrix alone (a third of the zeros to write)
            J1 = np.zeros((P, nx))
            J1[t_row, t_col] = ((res[t_pos, column] - maxSep) - F0[t_row, column]) / dx[t_col]
            return J1
        J = np.zeros((P, 3, nx))
        J[t_row, :, t_col] = ((res[t_pos] - maxSep) - F0[t_row]) / dx[t_col][:, None]
        return J.reshape(3 * P, nx)

    # ------------------------------------------------------------------ batched Jacobians (new)
    def _fd_rows(self, x):
        """x and its n_x forward-difference neighbours, SciPy-style: rows[k+1] = x + h e_k,
        dx[k] = (x_k + h) - x_k."""
        x = np.asarray(x, dtype=float)
        X = np.repeat(x[None], x.size + 1, axis=0)
        idx = np.arange(x.size)
        X[idx + 1, idx] += FD_STEP
        dx = X[idx + 1, idx] - x
        return X, dx

    def _serve(self, family, x, direct):
        """A constraint closure's value at x -- from ONE batched evaluation when x is a row of SciPy's finite differences.

        The reference hands SLSQP bare closures (Examples/*.py: `{'type': 'ineq', 'fun': bezopt.temporalSeparationConstraints}`),
        so SciPy differentiates each of them itself: n_x calls at x0 + h e_k per closure and iteration, each a device launch of
        one row (Example2's five vehicles: 3112 calls per solve).  The driver stays as it is; this notices the pattern instead.
        The first call that differs from the last base point x0 in ONE variable by exactly SciPy's step is taken for row k of a
        forward difference: the closure's whole batch F(x0), F(x0 + h e_1), ... is evaluated in one launch (the rows `_jac`
        forms, formed on the device from x0), kept, and this and the following calls are answered from it.  A call at any other
        point is a new base (a line-search step, the next iterate) and drops the batches.  Values are those of the one-row call,
        element for element (tests/test_gpu_dropin.py::test_scipy_finite_differences_served_from_one_batch: SLSQP takes the same
        iterates either way).  Batches above OBTG_FD_BATCH_MB (512) per closure are not formed; steps SciPy turned around at a
        bound (x0 - h) are evaluated directly."""
        if not self.fdBatching:
            return direct(x)
        x = np.asarray(x, dtype=float)
        st = self._fd_state
        d = None
        if st is not None and st['x0'].shape == x.shape:
            d = np.flatnonzero(x != st['x0'])
            if d.size > 1 or st['key'] != self._serve_key(True):      # several variables moved, or the problem was edited: a new base
                st = None
        else:
            st = None
        if st is None:
            self.fdBatchingStats['direct'] += 1
            v = direct(x)
            # (the key after the evaluation: reshapeVector has just refreshed the model's byte key, nothing is computed twice)
            self._fd_state = {'key': self._serve_key(False), 'x0': x.copy(), 'base': {family: v}, 'rows': {}}
            return None if v is None else v.copy()
        if d.size == 0:
            if family not in st['base']:
                self.fdBatchingStats['direct'] += 1
                st['base'][family] = direct(x)
            v = st['base'][family]
            return None if v is None else v.copy()
        k = int(d[0])
        if x[k] != st['x0'][k] + FD_STEP:
            self.fdBatchingStats['direct'] += 1              # (a step turned around at a bound, or not SciPy's at all)
            return direct(x)
        rows = st['rows'].get(family)
        if rows is None:
            base = st['base'].get(family)
            if base is None:
                base = st['base'][family] = direct(st['x0'])
                self.fdBatchingStats['direct'] += 1
            limit = float(os.environ.get("OBTG_FD_BATCH_MB", "512")) * 2.0 ** 20
            if base is None or 8.0 * base.size * (x.size + 1) > limit:
                rows = st['rows'][family] = False            # (too large, or a closure without rows: evaluate directly)
            elif family.startswith('spatial'):
                rows = self._spatial_fd_values(st['x0'], family == 'spatial_robust')
                rows = st['rows'][family] = False if rows is None else rows
                self.fdBatchingStats['batches'] += rows is not False
            else:
                rows = st['rows'][family] = self._fd_values(st['x0'], family)[0]
                self.fdBatchingStats['batches'] += 1
        if rows is False:
            self.fdBatchingStats['direct'] += 1
            return direct(x)
        self.fdBatchingStats['served'] += 1
        return rows[k + 1].copy()

    def _serve_key(self, refresh):
        """What a kept batch depends on besides x: DEG_ELEV, the row options, the model (its arrays by their bytes: the key
        reshapeVector keeps; refresh = recompute it now), the bounds' values, the obstacles.  The obstacles by their BYTES on
        every call, on purpose: drivers edit `pointObstacles` / a track's control points in place between solves, which a key of
        object identities would not see (4 point obstacles: 1 us; 32 curve obstacles: 30 us beside a `_minDist` sweep of
        milliseconds); _serve asks for the key only when at most one variable moved."""
        if refresh or getattr(self, '_rv_cache', None) is None:
            self._rv_parts()
        return (int(DEG_ELEV), self.separationRows, self.activeRows, self._rv_cache[0], self.model['maxSep'], self.model['maxSpeed'],
                self.model['minSpeed'], self.model['maxAngRate'], None if self._timeopt() else self.model['tf'],
                None if self.pointObstacles is None else np.asarray(self.pointObstacles, dtype=float).tobytes(),
                None if self.shapeObstacles is None else tuple(np.asarray(c.cpts, dtype=float).tobytes() for c in self.shapeObstacles))

    def _jac(self, x, family):
        F, dx = self._fd_values(x, family)
        return ((F[1:] - F[0:1]) / dx[:, None]).T

    def _fd_values(self, x, family):
        """(F, dx): the closure `family` on x and its n_x forward-difference neighbours, F[k + 1] = closure(x + h e_k), in one
        batched device call."""
        X, dx = self._fd_rows(x)
        Y = self.reshapeVectors(X)
        if family.startswith('obj_'):      # the objectives (one value per row; a column here)
            c = self._ctx(False)
            F = c.euclidean_obj(Y) if family == 'obj_euclidean' else c.deriv_energy_obj(Y, self.model['tf'], 2 if family == 'obj_accel' else 3)
            return np.asarray(F, dtype=float).reshape(-1, 1), dx
        if family == 'tsep':
            with_obs = self.pointObstacles is not None
            if self.separationRows == 'min':
                F = self._ctx(with_obs).temporal_sep_min(Y, self.model['maxSep'])
            elif self.separationRows == 'active':
                F = self._ctx(with_obs).temporal_sep_active(Y, self.model['maxSep'], self._active_k())
            else:
                F = self._ctx(with_obs).temporal_sep(Y, self.model['maxSep'])
        else:
            tf = X[:, -1] if self._timeopt() else np.full(X.shape[0], self.model['tf'])
            c = self._ctx(False)
            if family == 'vmax':
                F = c.speed(Y, tf, self.model['maxSpeed'], True)
            elif family == 'vmin':
                F = c.speed(Y, tf, self.model['minSpeed'], False)
            else:
                F = c.ang_rate(Y, tf, self.model['maxAngRate'])
        return F, dx

    def temporalSeparationJacobian(self, x, structured=True):
        """d(temporalSeparationConstraints)/dx by SciPy-style forward differences.

        structured=True: a variable of vehicle v only moves the N-1 pairs
        that contain v, so only those pairs are re-evaluated (obtg_temporal_sep_fd) and the dense
        matrix is assembled from them; every entry equals the brute-force batch's
        (structured=False) bit for bit.  Variables that move every vehicle (tf with prescribed
        speeds) and shapes outside the specialised kernels take the batch path."""
        if not structured or self.separationRows == 'active':
            # 'active': the forward differences of the order statistics themselves -- what SciPy builds from n_x + 1 calls
            # of the closure -- from one batched call (k values per pair and row leave the device, not 2n+R+1)
            return self._jac(x, 'tsep')
        x = np.asarray(x, dtype=float)
        X, dx = self._fd_rows(x)
        with_obs = self.pointObstacles is not None
        ctx = self._ctx(with_obs)
        n_obj = ctx.n_veh + ctx.n_obs
        if n_obj < 2:
            return self._jac(x, 'tsep')
        dim, numCols = self.model['dim'], self._numCols
        n_pts = self.model['numVeh'] * dim * numCols          # control-point variables; a trailing tf is not one
        offset = (self.model['deg'] + 1 - numCols) // 2
        Y0 = self.reshapeVectors(x[None])[0]
        k = np.arange(n_pts)
        prow, pcol = k // numCols, offset + k % numCols
        try:
            blk = ctx.temporal_sep_fd(Y0, prow, pcol, X[k + 1, k], self.model['maxSep'])
        except _capi.ObtgError:
            return self._jac(x, 'tsep')
        F0 = ctx.temporal_sep(Y0[None], self.model['maxSep'])[0]
        LR = blk.shape[2]
        reduced = self.separationRows == 'min'
        if reduced:
            # forward differences of the per-pair minimum itself: min over the perturbed pair's control points minus
            # min over the unperturbed ones -- exactly what SciPy would form from n_x + 1 calls of the 'min' closure
            blk = blk.min(axis=2, keepdims=True)
            F0 = F0.reshape(-1, LR).min(axis=1)
            LR = 1
        J = np.zeros((F0.size, x.size))
        veh = prow // dim
        partners = np.arange(n_obj - 1)[None, :] + (np.arange(n_obj - 1)[None, :] >= veh[:, None])   # [n_pts][n_obj-1]
        lo, hi = np.minimum(veh[:, None], partners), np.maximum(veh[:, None], partners)
        pidx = lo * n_obj - lo * (lo + 1) // 2 + (hi - lo - 1)                                        # lexicographic pair index
        rows = pidx[:, :, None] * LR + np.arange(LR)[None, None, :]
        J[rows.reshape(n_pts, -1), k[:, None]] = (blk - F0[rows]).reshape(n_pts, -1) / dx[:n_pts, None]
        if x.size > n_pts:      # tf: moves columns 1 / -2 of every vehicle when speeds are prescribed
            Yt = self.reshapeVectors(X[n_pts + 1:])
            Ft = ctx.temporal_sep_min(Yt, self.model['maxSep']) if reduced else ctx.temporal_sep(Yt, self.model['maxSep'])
            J[:, n_pts:] = ((Ft - F0[None]) / dx[n_pts:, None]).T
        return J

    def _jac_vehicle(self, x, family, structured=True):
        """Per-vehicle families (speed, angular rate): the Jacobian is block diagonal -- a control
        point of vehicle v only moves v's own rows.  structured=True evaluates, per variable, that one
        vehicle (a compact batch on a one-vehicle context: n_x vehicle evaluations instead of
        n_x N); entries equal the brute-force batch's bit for bit.  A trailing tf moves everything and
        takes the batch path."""
        if not structured:
            return self._jac(x, family)
        x = np.asarray(x, dtype=float)
        X, dx = self._fd_rows(x)
        dim, numCols, numVeh = self.model['dim'], self._numCols, self.model['numVeh']
        n_pts = numVeh * dim * numCols
        offset = (self.model['deg'] + 1 - numCols) // 2
        Y0 = self.reshapeVectors(x[None])[0]
        tf0 = float(self._tf_of(x))

        def evaluate(ctx, Y, tf):
            if family == 'vmax':
                return ctx.speed(Y, tf, self.model['maxSpeed'], True)
            if family == 'vmin':
                return ctx.speed(Y, tf, self.model['minSpeed'], False)
            return ctx.ang_rate(Y, tf, self.model['maxAngRate'])

        F0 = evaluate(self._ctx(False), Y0[None], np.array([tf0]))[0]
        per = F0.size // numVeh
        k = np.arange(n_pts)
        prow, pcol = k // numCols, offset + k % numCols
        veh = prow // dim
        Yc = Y0.reshape(numVeh, dim, -1)[veh].copy()              # [n_pts][dim][deg+1]: the touched vehicle of each variable
        Yc[k, prow % dim, pcol] = X[k + 1, k]
        Fc = evaluate(self._ctx_one(), Yc, np.full(n_pts, tf0))   # [n_pts][per]
        J = np.zeros((F0.size, x.size))
        rows = veh[:, None] * per + np.arange(per)[None, :]
        J[rows, k[:, None]] = (Fc - F0[rows]) / dx[:n_pts, None]
        if x.size > n_pts:
            Xt = X[n_pts + 1:]
            Ft = evaluate(self._ctx(False), self.reshapeVectors(Xt), Xt[:, -1])
            J[:, n_pts:] = ((Ft - F0[None]) / dx[n_pts:, None]).T
        return J

    def maxSpeedJacobian(self, x, structured=True):
        return self._jac_vehicle(x, 'vmax', structured)

    def minSpeedJacobian(self, x, structured=True):
        return self._jac_vehicle(x, 'vmin', structured)

    def maxAngularRateJacobian(self, x, structured=True):
        return self._jac_vehicle(x, 'ang', structured)

    # ------------------------------------------------------------------ x <-> y
    def generateGuess(self, std=0, seed=None):
        """Straight-line initial guess with N(0, std^2) noise (optimization.py:189-240), all vehicles at once.
        One `randn` draw of numVeh*dim*len values consumes NumPy's legacy stream in the order the reference's
        per-row draws do, so a given `seed` yields the reference's guess."""
        m = self.model
        numVeh, dim, deg = m['numVeh'], m['dim'], m['deg']
        start = np.asarray(m['initPoints'], dtype=float)[:numVeh]
        stop = np.asarray(m['finalPoints'], dtype=float)[:numVeh]
        npts = deg + 1
        if m['initSpeeds'][0] is not None:
            if dim != 2:
                raise ValueError('The dimension must be 2 for initial and final speeds and angles.')
            # prescribed speeds pin the second and the second-to-last control point: the line runs between those
            heading0 = np.stack((np.cos(m['initAngs']), np.sin(m['initAngs'])), axis=1)
            heading1 = np.stack((np.cos(m['finalAngs']), np.sin(m['finalAngs'])), axis=1)
            start = start + (np.asarray(m['initSpeeds'], dtype=float) * m['tf'] / deg)[:, None] * heading0
            stop = stop - (np.asarray(m['finalSpeeds'], dtype=float) * m['tf'] / deg)[:, None] * heading1
            npts = deg - 1
        np.random.seed(seed)
        # row by row: np.linspace with array end points switches formula when ANY row has start == stop, which
        # would move the other rows' last bits away from the reference's scalar calls
        lines = np.array([np.linspace(a, b, npts) for a, b in zip(start.ravel(), stop.ravel())])
        lines += np.random.randn(numVeh * dim, npts) * std
        guess = lines[:, 1:-1].reshape(-1)
        return np.append(guess, m['tf']) if self._timeopt() else guess

    def reshapeVector(self, x):
        """x -> y[(numVeh*dim) x (deg+1)] (optimization.py:242-285).

        SLSQP calls this once per callback, and at Example1's size a dozen small NumPy operations cost more than the
        device call they feed (23 us).  Everything that does not depend on x -- end points, the headings' sines and
        cosines -- is worked out once per problem (`_rv_parts`); a call copies that template, sets the two speed
        columns from tf and drops x into the free columns: 8 us.  `reshapeVectors` is the same thing with a leading
        batch axis (one row per finite-difference neighbour of x)."""
        return self.reshapeVectors(x)

    def reshapeVectors(self, X):
        """X[B][n_x] -> Y[B][numVeh*dim][deg+1] (any number of leading axes, none included): the template broadcast
        over the rows."""
        template, first_free, headings = self._rv_parts()
        X = np.asarray(X, dtype=float)
        if self._timeopt():                                   # time-optimal problems carry tf as the last variable
            tf, X = (X[..., -1:] if X.ndim > 1 else X[-1]), X[..., :-1]
        else:
            tf = float(self.model['tf'])
        lead = X.shape[:-1]
        if lead:
            Y = np.empty(lead + template.shape)
            Y[...] = template
        else:
            Y = template.copy()
        if headings is not None:
            # the second / second-to-last control points realise the prescribed speeds: p0 + (v0 tf / deg) (cos, sin)
            # and pf - (vf tf / deg) (cos, sin); even rows of y are x coordinates, odd rows y coordinates
            deg = self.model['deg']
            for col, (speed, px, py, c, s) in ((1, headings[0]), (-2, headings[1])):
                reach = speed * tf / deg                      # [...][numVeh]; the final end's cos / sin carry the minus
                Y[..., 0::2, col] = px + reach * c
                Y[..., 1::2, col] = py + reach * s
        Y[..., first_free:template.shape[1] - first_free] = X.reshape(lead + (template.shape[0], self._numCols))
        return Y

    def _rv_parts(self):
        """(template of y with its constant columns set, index of the first free column, per end of the curve
        (speed, x, y, cos, sin) or None) -- rebuilt when the problem's `model` entries change."""
        m = self.model
        fields = [m[k] for k in ('initPoints', 'finalPoints', 'initSpeeds', 'finalSpeeds', 'initAngs', 'finalAngs')]
        key = (m['numVeh'], m['dim'], m['deg']) + tuple(a.tobytes() if a.dtype != object else id(a) for a in fields)
        cached = getattr(self, '_rv_cache', None)
        if cached is not None and cached[0] == key:
            return cached[1]
        p0, pf, v0, vf, a0, af = fields
        dim = m['dim']
        # atleast_2d(None) is an object array: like the reference, a problem without end points fails on its first
        # reshape (optimization.py:267-271 assigns None into y) -- float() raises the same TypeError here
        p0, pf = p0.astype(float), pf.astype(float)
        template = np.empty((dim * m['numVeh'], m['deg'] + 1))
        rows = p0.shape[0] * dim
        template[:rows, 0] = p0.reshape(-1)[:rows]
        template[:rows, -1] = pf.reshape(-1)[:rows]
        first_free, headings = 1, None
        if v0[0] is not None:
            first_free = 2
            headings = ((v0.astype(float), p0[:, 0].copy(), p0[:, 1].copy(), np.cos(a0), np.sin(a0)),
                        (vf.astype(float), pf[:, 0].copy(), pf[:, 1].copy(), -np.cos(af), -np.sin(af)))
        self._rv_cache = (key, (template, first_free, headings))
        return self._rv_cache[1]
