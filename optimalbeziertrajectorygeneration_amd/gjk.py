"""`gjk.gjkNew` look-alike (reference gjk/gjk.py:230-270) on the MI355X.

    flag, info = gjkNew(poly1, poly2, maxIter=128, verbose=False)

flag -1: maximum iterations met, 0: collision, 1: separated with
info = (point on poly1, point on poly2, distance); info == () otherwise.  The reference's
decision sequence is reproduced exactly (support indices are bit-identical), including its
habit of stopping at a non-minimal distance.  Inputs on which the reference's
`minimumDistance` loops forever (gjk.py:277) raise RuntimeError here: as soon as the loop's state
repeats (3-D input) or after `md_cap` rounds.
"""
import numpy as np

from . import _capi


def gjkNew(poly1, poly2, maxIter=128, verbose=False, md_cap=4096):
    p1 = np.asarray(poly1, dtype=float).reshape(-1, 3)
    p2 = np.asarray(poly2, dtype=float).reshape(-1, 3)
    pts = np.vstack((p1, p2))
    off = [0, p1.shape[0], p1.shape[0] + p2.shape[0]]
    r = _capi.scratch_context().gjk_pairs(pts, off, [0], [1], max_iter=int(maxIter), md_cap=md_cap)
    flag, status = int(r['flag'][0]), int(r['status'][0])
    if status == _capi.ST_CYCLE:
        raise RuntimeError('gjkNew: minimumDistance cycles (the reference loops forever on this input)')
    if status == _capi.ST_MD_CAP:
        raise RuntimeError('gjkNew: minimumDistance did not converge in %d rounds '
                           '(the reference loops forever on this input)' % md_cap)
    if flag == 0:
        return 0, ()
    if flag == 1:
        return 1, (r['c1'][0].copy(), r['c2'][0].copy(), float(r['dist'][0]))
    print('Maximum iterations met')
    return -1, ()


def gjkPairs(polys, pair_a, pair_b, maxIter=128, md_cap=4096, trace_cap=0):
    """Batched form: `polys` is a list of (K,3) arrays; returns the arrays of obtg_gjk_pairs."""
    off = np.zeros(len(polys) + 1, dtype=np.int32)
    for i, p in enumerate(polys):
        off[i + 1] = off[i] + np.asarray(p).reshape(-1, 3).shape[0]
    pts = np.vstack([np.asarray(p, dtype=float).reshape(-1, 3) for p in polys])
    return _capi.scratch_context().gjk_pairs(pts, off, pair_a, pair_b, max_iter=int(maxIter), md_cap=md_cap,
                                             trace_cap=trace_cap)


def gjkTrue(poly1, poly2, eps=1e-10, maxIter=64):
    """The true distance between the convex hulls of two point sets -- a textbook GJK with a certified exit
    (obtg_gjk_true_pairs), offered beside gjkNew because gjkNew stops at a non-minimal distance on about 30 % of
    separated pairs.  Same return convention: (1, (p1, p2, dist)) or (0, ())."""
    p1 = np.asarray(poly1, dtype=float).reshape(-1, 3)
    p2 = np.asarray(poly2, dtype=float).reshape(-1, 3)
    r = _capi.scratch_context().gjk_true_pairs(np.vstack((p1, p2)), [0, p1.shape[0], p1.shape[0] + p2.shape[0]], [0], [1],
                                               eps=eps, max_iter=int(maxIter))
    if int(r['flag'][0]) == 0:
        return 0, ()
    return 1, (r['c1'][0].copy(), r['c2'][0].copy(), float(r['dist'][0]))
