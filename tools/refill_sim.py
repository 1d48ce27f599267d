"""Scratch model of the gjkNew sweep's lane-refill loop (phase 1 of gjk_planar_body): how many wave-iterations and
VALU wave-instructions a workgroup's chunk costs under different refill policies, on the oracle's true trip counts.
Cost model (SQ_INSTS_VALU per wave-iteration): always 196 (scan 136, tests 30, bookkeeping 20, old copy 10),
+50 when any lane runs the three-point update, +R_REFILL + 45 when any lane refills (refill path + two-point update)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from optimalbeziertrajectorygeneration_amd import synth
from oracle import oracle as O

C_COMMON, C_3PT, C_REFILL, C_2PT = 196, 50, 66, 45


def lens_c3(seed=1234):
    cfg = synth.CONFIGS["C3"]
    Y = synth.swarm_control_points(cfg["N"], cfg["d"], cfg["n"], seed=seed)
    statics, pa, pb = synth.config_hull_sweep("C3", seed=seed)
    hp, ho = synth.pack_polys(synth.hulls_from_Y(Y, 2) + statics)
    o = O.gjk_pairs(hp, ho, pa, pb, md_cap=1000)
    return o["n_support"].astype(int)


def simulate(lens, policy, T=1, waves=4, pred=None, sort=True):
    """lens: true scan counts of the chunk's pairs. pred: predicted (history) counts used for ordering."""
    pred = lens if pred is None else pred
    order = np.argsort(-pred, kind="stable") if sort else np.arange(len(lens))
    rem_all = np.maximum(lens[order] - 2, 0)      # loop iterations after the two tabulated steps
    nxt = 0
    n = len(rem_all)
    W = [dict(rem=np.zeros(64, int), first=np.zeros(64, bool), busy=np.zeros(64, bool), done=False) for _ in range(waves)]
    iters = instr = lane_busy = 0
    refill_iters = 0
    while not all(w["done"] for w in W):
        for w in W:
            if w["done"]:
                continue
            idle = ~w["busy"]
            refilled = False
            do_refill = False
            if nxt < n and idle.any():
                if policy == "any":
                    do_refill = True
                elif policy == "threshold":
                    do_refill = idle.sum() >= T
                elif policy == "all":
                    do_refill = idle.all()
            if do_refill:
                for l in np.nonzero(idle)[0]:
                    if nxt >= n:
                        break
                    r = rem_all[nxt]; nxt += 1
                    if r > 0:
                        w["rem"][l] = r; w["busy"][l] = True; w["first"][l] = True
                    refilled = True
            if not w["busy"].any():
                if nxt >= n:
                    w["done"] = True
                    continue
                # threshold policy with nothing busy: force refill
                for l in range(64):
                    if nxt >= n:
                        break
                    r = rem_all[nxt]; nxt += 1
                    if r > 0:
                        w["rem"][l] = r; w["busy"][l] = True; w["first"][l] = True
                    refilled = True
                if not w["busy"].any():
                    continue
            iters += 1
            c = C_COMMON
            if (w["busy"] & ~w["first"]).any():
                c += C_3PT
            if refilled:
                c += C_REFILL
                refill_iters += 1
            if (w["busy"] & w["first"]).any():
                c += C_2PT
            instr += c
            lane_busy += w["busy"].sum()
            w["first"][:] = False
            w["rem"][w["busy"]] -= 1
            w["busy"] &= w["rem"] > 0
    return dict(iters=iters, instr=instr, per_iter=instr / iters, occ=lane_busy / (64.0 * iters), refill_frac=refill_iters / iters)


if __name__ == "__main__":
    L = lens_c3()
    print("pairs", len(L), "mean scans", L.mean(), np.bincount(L))
    chunks = [L[:1280], L[1280:]]
    rng = np.random.default_rng(0)
    for name, kw in [("any (today)", dict(policy="any")),
                     ("threshold 8", dict(policy="threshold", T=8)), ("threshold 16", dict(policy="threshold", T=16)),
                     ("threshold 32", dict(policy="threshold", T=32)), ("threshold 48", dict(policy="threshold", T=48)),
                     ("all idle", dict(policy="all")), ("any, list order", dict(policy="any", sort=False)),
                     ("all idle, list order", dict(policy="all", sort=False)),
                     ("threshold 32, list order", dict(policy="threshold", T=32, sort=False))]:
        tot = dict(iters=0, instr=0)
        occ = []
        rf = []
        for ch in chunks:
            r = simulate(ch, **kw)
            tot["iters"] += r["iters"]; tot["instr"] += r["instr"]; occ.append(r["occ"]); rf.append(r["refill_frac"])
        print("%-26s iters %4d  instr %7d  per-iter %5.1f  occupancy %.3f  refill-iters %.2f" %
              (name, tot["iters"], tot["instr"], tot["instr"] / tot["iters"], np.mean(occ), np.mean(rf)))
    # imperfect history: predicted = true + noise on 10 % of the pairs
    for frac in (0.1, 0.3):
        for name, kw in [("any", dict(policy="any")), ("threshold 32", dict(policy="threshold", T=32)), ("all idle", dict(policy="all"))]:
            tot = dict(iters=0, instr=0)
            for ch in chunks:
                pred = ch.copy()
                m = rng.random(len(ch)) < frac
                pred[m] = np.clip(pred[m] + rng.integers(-2, 3, m.sum()), 2, 30)
                r = simulate(ch, pred=pred, **kw)
                tot["iters"] += r["iters"]; tot["instr"] += r["instr"]
            print("mispredicted %.0f%%: %-14s iters %4d instr %7d" % (100 * frac, name, tot["iters"], tot["instr"]))
