"""Where an expansion's clocks go in k_min_dist_quad: run with OBTG_LIB pointing at a build with -DOBTG_MD_TIMING=1
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
split, walk = (st & 0xffff) * 1024, (st >> 16) * 1024
import os
if os.environ.get("OBTG_PROBE_SET") == "2":      # a build with -DOBTG_MD_TIMING=2: split parameters, bound + record, blob store, blob re-fetch
    for name, sel in (("all pairs", nodes > 0), ("pairs at the 2000-node cap", nodes >= 2000)):
        tn = nodes[sel].sum(); trips = (tn - sel.sum()) / 4.0 + sel.sum()
        print("%s: clocks per evaluation: split parameters (hull_param x 2) %.0f, end-point bound + record %.0f, blob store %.0f, blob re-fetch %.0f"
              % (name, gjk[sel].sum() / trips, ev[sel].sum() / trips, split[sel].sum() / trips, walk[sel].sum() / trips))
    sys.exit(0)
for name, sel in (("all pairs", nodes > 0), ("pairs at the 2000-node cap", nodes >= 2000)):
    tn = nodes[sel].sum(); trips = (tn - sel.sum()) / 4.0 + sel.sum()
    print("%s: %d pairs, %d nodes, ~%.0f evaluations of four; clocks per evaluation: lockstep gjkNew %.0f, whole evaluation %.0f, "
          "walk between evaluations %.0f of which split %.0f; per node %.0f"
          % (name, sel.sum(), tn, trips, gjk[sel].sum() / trips, ev[sel].sum() / trips, walk[sel].sum() / trips, split[sel].sum() / trips,
             (ev[sel] + walk[sel]).sum() / tn))
tot = (ev + walk)[nodes >= 2000] / 2.4e6
print("pairs at the cap, evaluation + walk clocks as ms at 2.4 GHz: min %.2f  p10 %.2f  median %.2f  p90 %.2f  max %.2f" % (
    tot.min(), np.percentile(tot, 10), np.median(tot), np.percentile(tot, 90), tot.max()))
g = gjk[nodes >= 2000] / 2.4e6
print("  of which the lockstep gjkNew: min %.2f median %.2f max %.2f" % (g.min(), np.median(g), g.max()))
