"""One-vs-many separation constraint of the reference's sequential swarm planner
(Examples/SequentialSwarm.py:43-70): the trajectory in rows 0..ndim-1 of `y` against every other
trajectory, reduced to the minimum elevated control point per pair:

    distVeh[i-1] = (y_0 - y_i).normSquare().elev(10).cpts.min()  ->  distVeh - maxSep**2

On the MI355X this is the fused per-pair minimum sweep (`obtg_temporal_sep_min`) restricted to the
first nveh-1 pairs of the lexicographic pair list -- the pairs (0, i).  Nothing of the
(nveh-1) x (2n+R+1) intermediate leaves the chip.
"""
import numpy as np

from . import _capi

_ctx_cache = {}


def _context(nveh, ndim, deg, deg_elev):
    key = (nveh, ndim, deg, deg_elev)
    c = _ctx_cache.get(key)
    if c is None:
        if len(_ctx_cache) > 8:          # planning vehicle after vehicle changes nveh every call
            _ctx_cache.pop(next(iter(_ctx_cache))).close()
        c = _capi.Context(nveh, ndim, deg, deg_elev)
        _ctx_cache[key] = c
    return c


def temporalSeparationConstraints(y, nveh, ndim, maxSep, degElev=10):
    """Same signature as Examples/SequentialSwarm.py:43 (the elevation, hard-coded to 10 there,
    is a keyword here)."""
    if nveh <= 1:
        return np.atleast_1d(0.0)                     # SequentialSwarm.py:69-70
    y = np.ascontiguousarray(y, dtype=np.float64)
    ctx = _context(nveh, ndim, y.shape[1] - 1, int(degElev))
    return ctx.temporal_sep_min(y[None], maxSep, pair_begin=0, pair_count=nveh - 1)[0]
