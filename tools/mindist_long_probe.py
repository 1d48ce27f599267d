"""The capped pairs of the C5-sized list alone (545 searches of 2000 nodes): how long the slowest takes when nobody shares its SIMD
(OBTG_MD_WAVES_PER_SIMD=1: 545 <= 1024 workers) and when two workers per SIMD are placed as they come (=2)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from optimalbeziertrajectorygeneration_amd import _capi, synth
N, M, n = 64, 32, 10
Yc = np.vstack((synth.swarm_control_points(N, 2, n, seed=1234), synth.curve_obstacles(M, 2, n, seed=1234)))
curves = np.zeros((N + M, 3, n + 1)); curves[:, :2, :] = Yc.reshape(N + M, 2, n + 1)
pa, pb = synth.all_pairs(N + M)
ctx = _capi.scratch_context()
r = ctx.min_dist(curves, pa, pb, eps=1e-9, max_depth=128, max_nodes=2000)
big = r["nodes"] >= 2000
a, b = pa[big], pb[big]
for _ in range(3): ctx.min_dist(curves, a, b, eps=1e-9, max_depth=128, max_nodes=2000)
t0 = time.perf_counter()
for _ in range(5): ctx.min_dist(curves, a, b, eps=1e-9, max_depth=128, max_nodes=2000)
print("%d capped pairs alone: %.2f ms per call" % (big.sum(), 1e3 * (time.perf_counter() - t0) / 5))
