"""`bezier.Bezier` look-alike whose arithmetic runs on the MI355X through libobtg_hip.so.

Mirrors the part of the reference's object API that the SLSQP constraint callbacks and the
drivers use (reference bezier.py:34-270 container, 318-519 add/sub/mul/div/elev/diff,
840-889 minDist/minDist2Poly/normSquare).  Every method below that computes control points
calls the C ABI (include/obtg.h); there is no CPU implementation of those in this package.

Differences from the reference, on purpose:
  * the constructor accepts any array-like (the reference needs an ndarray and crashes on
    the lists its own `_minDist` passes, bezier.py:58, 1304);
  * `minDist` / `minDist2Poly` work (at the reference's HEAD they raise because `gjkNew` is
    never imported, bezier.py:21-22) and report the reference's non-terminating inputs as
    exceptions instead of hanging;
  * plotting (`plot`, matplotlib) and temporal alignment of curves with different [t0, tf] are outside the
    accelerated path and are not provided; `curve` / `__call__` -- what the drivers' own
    plotting code reads after a solve -- are (obtg_bern_eval).
"""
import numpy as np

from . import _capi


def _ctx():
    return _capi.scratch_context()


class _Grid(object):
    """Time span of a curve: its two ends and, on demand only, the sampling grid over them (the
    accelerated path never samples; drivers that plot do)."""
    __slots__ = ('ends', 'samples')
    POINTS = 1001                                          # the reference's grid size (bezier.py:118)

    def __init__(self, t0, tf, samples=None):
        if samples is None:
            self.ends = [float(t0), float(tf)]
        else:                                              # a grid that is handed in decides the span
            self.ends = [samples[0], samples[-1]]
        self.samples = samples

    def move_end(self, which, value):
        self.ends[which] = float(value)
        self.samples = None                                # a stale grid must not outlive its span

    def grid(self):
        if not isinstance(self.samples, np.ndarray):
            self.samples = (np.linspace(self.ends[0], self.ends[1], self.POINTS) if self.samples is None
                            else np.array(self.samples))
        return self.samples


def _span_end(which, name):
    return property(lambda self: self._span.ends[which], lambda self, v: self._span.move_end(which, v),
                    doc='%s of the span; assigning drops a cached tau grid' % name)


def _shape_field(axis, off, name):
    return property(lambda self: None if self._cpts is None else self._cpts.shape[axis] - off, doc=name)


class BezierParams(object):
    """Container half of the reference's curve type (bezier.py:34-145): control points d x (n+1) plus the
    time span.  The reference keeps `_dim` / `_deg` / `_t0` / `_tf` / `_tau` slots side by side and
    updates them in every setter; here the shape IS dim and deg (read off the array) and the span is
    one `_Grid`, so there is nothing to keep in step."""
    _cpts = None

    def __init__(self, cpts=None, tau=None, t0=0.0, tf=1.0):
        if cpts is not None:
            self.cpts = cpts
        self._span = _Grid(t0, tf, tau)

    @property
    def cpts(self):
        return self._cpts

    @cpts.setter
    def cpts(self, value):
        # float64 matrices are taken as they are (the kernels' results arrive that way); anything else --
        # lists, 1-D arrays, integer arrays -- is promoted the way the reference's setter does (bezier.py:92)
        ready = isinstance(value, np.ndarray) and value.ndim == 2 and value.dtype == np.float64
        self._cpts = value if ready else np.array(value, ndmin=2, dtype=float)

    dim = dimension = _shape_field(0, 0, 'rows of cpts')
    deg = degree = _shape_field(1, 1, 'columns of cpts, less one')
    t0 = _span_end(0, 't0')
    tf = _span_end(1, 'tf')

    @property
    def tau(self):
        return self._span.grid()

    @tau.setter
    def tau(self, val):
        self._span = _Grid(0.0, 0.0, np.array(val))


class Bezier(BezierParams):
    """Bezier(cpts=None, t0=0.0, tf=1.0, tau=None) -- reference bezier.py:148-166."""

    def __init__(self, cpts=None, t0=0.0, tf=1.0, tau=None):
        super(Bezier, self).__init__(cpts=cpts, tau=tau, t0=t0, tf=tf)

    def __add__(self, curve):
        return self.add(curve)

    def __sub__(self, curve):
        return self.sub(curve)

    def __mul__(self, curve):
        return self.mul(curve)

    def __truediv__(self, curve):
        return self.div(curve)

    def __repr__(self):
        return 'Bezier({}, {}, {}, {})'.format(self.cpts, self.tau, self.t0, self.tf)       # (bezier.py:180-182: tau is printed too)

    @property
    def x(self):
        return Bezier(self.cpts[0], t0=self.t0, tf=self.tf)

    @property
    def y(self):
        return Bezier(self.cpts[1], t0=self.t0, tf=self.tf) if self.dim > 1 else None

    @property
    def z(self):
        return Bezier(self.cpts[2], t0=self.t0, tf=self.tf) if self.dim > 2 else None

    def __call__(self, t):
        """The curve at the value(s) t, dim x len(t) (bezier.py:184-199); not cached."""
        return _ctx().bern_eval(self.cpts, np.atleast_1d(t), self.t0, self.tf)

    @property
    def curve(self):
        """The curve at every value of `tau` (1001 samples over [t0, tf] unless a grid was given), dim x len(tau)
        (bezier.py:233-258): what the drivers plot after a solve.  Sampled on the device on every access -- the
        reference caches it until cpts / tau change; a sample set is 24 KB and one launch."""
        return _ctx().bern_eval(self.cpts, self.tau, self.t0, self.tf)

    def copy(self):
        return Bezier(self.cpts, self.t0, self.tf)

    # ---- arithmetic (bezier.py:318-374): equal time spans only
    def _same_span(self, other):
        if not (self.t0 == other.t0 and self.tf == other.tf):
            raise NotImplementedError('curves with different [t0, tf] need the reference\'s temporal '
                                      'alignment (bezier.py:903-941), which is outside the accelerated path')

    def add(self, other):
        self._same_span(other)
        if self.t0 >= self.tf:
            return None
        return Bezier(self.cpts + other.cpts, t0=self.t0, tf=self.tf)

    def sub(self, other):
        self._same_span(other)
        if self.t0 >= self.tf:
            return None
        return Bezier(self.cpts - other.cpts, t0=self.t0, tf=self.tf)

    def mul(self, multiplicand):
        """Product of two curves (bezier.py:376-432), dimension by dimension."""
        if not isinstance(multiplicand, Bezier):
            raise TypeError('The multiplicand must be a {} object, not a {}'.format(Bezier, type(multiplicand)))
        if multiplicand.dim != self.dim:
            raise ValueError('The dimension of both Bezier curves must be the same.\n'
                             'The first dimension is {} and the second is {}'.format(self.dim, multiplicand.dim))
        new = self.copy()
        new.cpts = _ctx().bern_mul(self.cpts, multiplicand.cpts)
        return new

    def div(self, denominator):
        """Rational curve numerator/denominator (bezier.py:434-467): element-wise on control points."""
        if not isinstance(denominator, Bezier):
            raise TypeError('The denominator must be a Bezier object, not a {}. '
                            'Or the module has been reloaded.'.format(type(denominator)))
        num, den = self.cpts, denominator.cpts
        with np.errstate(divide='ignore', invalid='ignore'):
            cpts = np.where(num == 0, 0.0, np.where(den == 0, np.inf, num / den))
        return RationalBezier(cpts.astype(np.float64), den.astype(np.float64), tau=self.tau, tf=self.tf)

    def elev(self, R=1):
        """Degree elevation by R (bezier.py:469-495)."""
        new = self.copy()
        new.cpts = _ctx().bern_elev(self.cpts, int(R))
        return new

    def diff(self):
        """Derivative, elevated back to the original degree (bezier.py:497-519)."""
        new = self.copy()
        new.cpts = _ctx().bern_diff(self.cpts, self.tf - self.t0)
        return new

    def split(self, tDiv):
        """Two curves, before and after tDiv (bezier.py:533-572): de Casteljau at (tDiv - t0)/(tf - t0); the
        pieces keep the original span's ends, [t0, tDiv] and [tDiv, tf]."""
        if np.isnan(tDiv):
            print('[!] Warning, tDiv is {}, changing to 0.'.format(tDiv))
            tDiv = 0
        left, right = _ctx().bern_split(self.cpts, (tDiv - self.t0) / (self.tf - self.t0))
        return Bezier(left, t0=self.t0, tf=tDiv), Bezier(right, t0=tDiv, tf=self.tf)

    def normSquare(self):
        """(d/2) * |curve|^2 as a 1 x (2n+1) curve -- the reference's factor is kept (bezier.py:869-889)."""
        new = self.copy()
        new.cpts = _ctx().bern_normsq(self.cpts)
        return new

    # ---- distances (bezier.py:840-857)
    def _padded(self):
        c = np.zeros((3, self.deg + 1))
        c[:self.dim] = self.cpts
        return c

    def minDist(self, otherCurve, eps=1e-9, max_depth=128, max_nodes=4000000, robust=False):
        """(dist, t1, t2).  Default: the reference's `_minDist` step for step (bezier.py:1283-1408), including
        its non-minimal answers.  robust=True: obtg_min_dist_robust, the true minimum within relative eps."""
        if self.dim < 2 or self.dim > 3 or otherCurve.dim < 2 or otherCurve.dim > 3:
            raise ValueError('Both curves must be either 2D or 3D, not {}D and {}D.'.format(self.dim, otherCurve.dim))
        if self.deg != otherCurve.deg:
            raise ValueError('minDist needs curves of equal degree here (got {} and {})'.format(self.deg, otherCurve.deg))
        if robust:
            r = _ctx().min_dist_robust(np.stack([self._padded(), otherCurve._padded()]), [0], [1], eps=eps,
                                       max_nodes=max_nodes)
            if r['status'][0] != _capi.MD_OK:
                raise RuntimeError('minDist(robust): search budget exhausted (curves coincide over a stretch?); '
                                   'best distance so far %g' % r['res'][0][0])
            a, t1, t2 = r['res'][0]
            return (float(a), float(t1), float(t2))
        r = _ctx().min_dist(np.stack([self._padded(), otherCurve._padded()]), [0], [1], eps=eps,
                            max_depth=max_depth, max_nodes=max_nodes)
        _raise_md(r['status'][0])
        a, t1, t2 = r['res'][0]
        return (float(a), float(t1), float(t2))

    def minDist2Poly(self, poly, eps=1e-6, max_depth=128, max_nodes=4000000, robust=False):
        """(dist, t, closest point on the polygon).  Default: the reference's `_minDist2Poly` step for step
        (bezier.py:1411-1496).  robust=True: obtg_min_dist2poly_robust, the true minimum within relative eps."""
        poly = np.asarray(poly, dtype=float)
        if robust:
            r = _ctx().min_dist2poly_robust(self._padded()[None], poly, [0, poly.shape[0]], [0], [0], eps=min(eps, 1e-9),
                                            max_nodes=max_nodes)
            if r['status'][0] != _capi.MD_OK:
                raise RuntimeError('minDist2Poly(robust): search budget exhausted; best distance so far %g' % r['res'][0][0])
            res = r['res'][0]
            return (float(res[0]), float(res[1]), res[2:].copy())
        r = _ctx().min_dist2poly(self._padded()[None], poly, [0, poly.shape[0]], [0], [0], eps=eps,
                                 max_depth=max_depth, max_nodes=max_nodes)
        _raise_md(r['status'][0])
        res = r['res'][0]
        return (float(res[0]), float(res[1]), res[2:].copy())


def _raise_md(status):
    if status == _capi.MD_OK:
        return
    if status == _capi.MD_DEPTH_CAP:
        raise RecursionError('minDist: subdivision deeper than max_depth (the reference overflows its stack here)')
    if status == _capi.MD_NODE_CAP:
        raise RuntimeError('minDist: node budget exhausted (the reference does not return on this input)')
    raise RuntimeError('minDist: an inner gjkNew did not converge (the reference loops forever here)')


class RationalBezier(BezierParams):
    """Container for control points + weights (bezier.py:894-900)."""

    def __init__(self, cpts=None, weights=None, tau=None, tf=1.0):
        super(RationalBezier, self).__init__(cpts=cpts, tau=tau, tf=tf)
        self._weights = np.array(weights, ndmin=2)
