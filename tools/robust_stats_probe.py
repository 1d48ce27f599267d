import numpy as np, sys, time
sys.path.insert(0, '.')
from optimalbeziertrajectorygeneration_amd import _capi, synth
N, M, n = 64, 32, 10
Yc = np.vstack((synth.swarm_control_points(N, 2, n, seed=1234), synth.curve_obstacles(M, 2, n, seed=1234)))
curves = np.zeros((N + M, 3, n + 1)); curves[:, :2, :] = Yc.reshape(N + M, 2, n + 1)
pa, pb = synth.all_pairs(N + M)
ctx = _capi.scratch_context()
r = ctx.min_dist_robust(curves, pa, pb, eps=1e-9, max_nodes=400000)
nodes, lev, fr = r["nodes"], r["levels"], r["frontier"]
o = np.argsort(-nodes)
print("total nodes", nodes.sum(), "pairs", len(nodes))
print("top 12 by nodes: ", [(int(nodes[i]), int(lev[i]), int(fr[i])) for i in o[:12]])
print("percentiles of nodes 50/90/99/99.9:", np.percentile(nodes, [50, 90, 99, 99.9]))
steps = np.ceil(np.maximum(fr, 1) / 64)
print("sum over pairs of nodes/64 (wave-steps lower bound):", int(np.ceil(nodes / 64).sum()), " max pair wave-steps >= ", int(np.ceil(nodes[o[0]] / 64)))
for reps in range(2):
    t = time.perf_counter(); ctx.min_dist_robust(curves, pa, pb, eps=1e-9, max_nodes=400000); print("ms", 1e3 * (time.perf_counter() - t))
# the longest pair alone
i = o[0]
t = time.perf_counter(); ctx.min_dist_robust(curves, pa[i:i+1], pb[i:i+1], eps=1e-9, max_nodes=400000); print("longest pair alone ms", 1e3 * (time.perf_counter() - t))
t = time.perf_counter(); ctx.min_dist_robust(curves, pa[i:i+1], pb[i:i+1], eps=1e-9, max_nodes=400000); print("longest pair alone ms", 1e3 * (time.perf_counter() - t))
