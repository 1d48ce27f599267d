"""obtg_min_dist on pair lists it has not seen (no node-count history): the C5-sized list rotated by a different amount per call.
OBTG_MD_HISTORY=0 gives list order for comparison."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from optimalbeziertrajectorygeneration_amd import _capi, synth
N, M, n = 64, 32, 10
Yc = np.vstack((synth.swarm_control_points(N, 2, n, seed=1234), synth.curve_obstacles(M, 2, n, seed=1234)))
curves = np.zeros((N + M, 3, n + 1)); curves[:, :2, :] = Yc.reshape(N + M, 2, n + 1)
pa, pb = synth.all_pairs(N + M)
ctx = _capi.scratch_context()
ctx.min_dist(curves, pa, pb, eps=1e-9, max_depth=128, max_nodes=2000)          # allocations, module load
ts = []
for i in range(1, 9):
    a, b = np.roll(pa, 17 * i), np.roll(pb, 17 * i)
    t0 = time.perf_counter(); r = ctx.min_dist(curves, a, b, eps=1e-9, max_depth=128, max_nodes=2000); ts.append(1e3 * (time.perf_counter() - t0))
print("first evaluation of a pair list (8 different lists): median %.2f ms, min %.2f, max %.2f; checksum %.6f" % (np.median(ts), min(ts), max(ts), np.nansum(r["res"][:, 0])))
t0 = time.perf_counter()
for _ in range(5): r = ctx.min_dist(curves, a, b, eps=1e-9, max_depth=128, max_nodes=2000)
print("the same list again (history order): %.2f ms" % (1e3 * (time.perf_counter() - t0) / 5))
