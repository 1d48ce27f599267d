"""Where a node's clocks go in k_min_dist_wave: run with OBTG_LIB pointing at a build with -DOBTG_MD_TIMING=1
(tools/build_variant.sh mdtm -DOBTG_MD_TIMING=1), whose info[] carries phase totals in units of 1024 clocks instead of
(calls, depth, status)."""
import sys
import numpy as np
sys.path.insert(0, '.')
from optimalbeziertrajectorygeneration_amd import _capi, synth
N, M, n = 64, 32, 10
Yc = np.vstack((synth.swarm_control_points(N, 2, n, seed=1234), synth.curve_obstacles(M, 2, n, seed=1234)))
curves = np.zeros((N + M, 3, n + 1)); curves[:, :2, :] = Yc.reshape(N + M, 2, n + 1)
pa, pb = synth.all_pairs(N + M)
ctx = _capi.scratch_context()
r = ctx.min_dist(curves, pa, pb, eps=1e-9, max_depth=128, max_nodes=2000)
nodes = r["nodes"].astype(np.int64)
gjk = r["gjk_calls"].astype(np.int64) * 1024
ev = r["depth"].astype(np.int64) * 1024
st = r["status"].astype(np.int64)
desc, fetch = (st & 0xffff) * 1024, (st >> 16) * 1024
tot = nodes.sum()
print("pairs %d nodes %d; clocks per node: gjkNew %.0f, rest of the node's evaluation %.0f, descend (fetch + split + store) %.0f of which re-fetch of the parent frame %.0f"
      % (len(pa), tot, gjk.sum() / tot, ev.sum() / tot, desc.sum() / tot, fetch.sum() / tot))
big = nodes >= 2000
print("pairs at the 2000-node cap: %d; their clocks per node: gjkNew %.0f, evaluation %.0f, descend %.0f (re-fetch %.0f); sum %.0f = %.2f us at 2.4 GHz"
      % (big.sum(), gjk[big].sum() / nodes[big].sum(), ev[big].sum() / nodes[big].sum(), desc[big].sum() / nodes[big].sum(), fetch[big].sum() / nodes[big].sum(),
         (gjk[big] + ev[big] + desc[big]).sum() / nodes[big].sum(), (gjk[big] + ev[big] + desc[big]).sum() / nodes[big].sum() / 2400.0))
