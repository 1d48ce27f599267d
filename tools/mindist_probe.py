"""spatialSeparationConstraints-sized `_minDist` sweep: all pairs of 64 vehicles + 32 curve obstacles (C5)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from optimalbeziertrajectorygeneration_amd import _capi, synth
N, M, n = 64, 32, 10
Y = synth.swarm_control_points(N + M, 2, n, seed=1234)           # obstacles: curves from the same generator
curves = np.zeros((N + M, 3, n + 1)); curves[:, :2, :] = Y.reshape(N + M, 2, n + 1)
pa, pb = np.triu_indices(N + M, 1)
ctx = _capi.scratch_context()
for cap in (2000, 200000):
    t = time.perf_counter()
    r = ctx.min_dist(curves, pa, pb, eps=1e-9, max_depth=128, max_nodes=cap)
    dt = time.perf_counter() - t
    st = np.bincount(r['status'], minlength=4)
    print('max_nodes %d: %d pairs in %.3f s; status counts ok/depth/nodes/gjk = %s; gjk calls total %d, median %d, max %d'
          % (cap, len(pa), dt, st.tolist(), r['gjk_calls'].sum(), np.median(r['gjk_calls']), r['gjk_calls'].max()))
t = time.perf_counter()
rr = ctx.min_dist_robust(curves, pa, pb, eps=1e-9, max_nodes=400000)
dt = time.perf_counter() - t
ok = r['status'] == 0
both = ok & (rr['status'] == 0)
worse = (r['res'][both, 0] > rr['res'][both, 0] * (1 + 1e-6)).sum()
print('robust: %d pairs in %.3f s; status counts ok/nodes/depth = %s; nodes total %d, median %d, max %d; largest frontier %d; '
      'reference-style answer non-minimal on %d of the %d pairs both finish (max ratio %.2f)'
      % (len(pa), dt, np.bincount(rr['status'], minlength=3).tolist(), rr['nodes'].sum(), np.median(rr['nodes']), rr['nodes'].max(),
         rr['frontier'].max(), worse, both.sum(), np.max(np.where(rr['res'][both, 0] > 1e-6, r['res'][both, 0] / np.maximum(rr['res'][both, 0], 1e-6), 1.0))))
