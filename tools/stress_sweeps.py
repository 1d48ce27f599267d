"""One-off stress run of the planar sweeps over random shapes (small rows, rows beyond 48 KB of LDS = tiled, partial
pair lists), each against the ORACLE and the one-launch pair sweep against the two separate launches.
    python tools/stress_sweeps.py [trials] [seed]        (needs an MI355X; the oracle is the checker)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from optimalbeziertrajectorygeneration_amd import _capi as capi, synth
from oracle import oracle


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
    dev = torch.device("cuda")
    t0 = time.time()
    for trial in range(trials):
        n = int(rng.choice([3, 5, 7, 10, 15]))
        big = trial % 3 == 0
        N = int(rng.integers(150, 700)) if big else int(rng.integers(2, 120))
        if big and 16 * N * ((n + 1) | 1) < 50 * 1024:
            N = 50 * 1024 // (16 * ((n + 1) | 1)) + int(rng.integers(5, 200))
        M = int(rng.integers(0, 6))
        B = int(rng.choice([1, 2, 3]))
        Y = synth.swarm_control_points(N, 2, n, seed=1000 + trial)
        Yb = synth.fd_batch(Y, B=B, h=0.5) if B > 1 else Y[None].copy()
        pa, pb = synth.swarm_pairs(N, M)
        partial = trial % 4 == 1
        if partial and len(pa) > 3:
            sel = rng.permutation(len(pa))[:max(1, len(pa) * 2 // 3)]
            pa, pb = pa[sel], pb[sel]
        polys = [p_[:min(len(p_), n + 1)] for p_ in synth.polygon_obstacles(M, seed=trial)] if M else []
        ctx = capi.Context(N, 2, n, 0)
        if M:
            ctx.set_polygons(*synth.pack_polys(polys))
        ctx.set_hull_pairs(pa, pb)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        dY = torch.from_numpy(Yb).to(dev)
        P, L, Ps = ctx.num_pairs, 2 * n + 1, len(pa)

        def bufs():
            return [torch.full((B, max(P, 1) * L), np.nan, dtype=torch.float64, device=dev),
                    torch.full((B, Ps), -7, dtype=torch.int32, device=dev),
                    torch.full((B, Ps, 3), -1.0, dtype=torch.float64, device=dev),
                    torch.full((B, Ps, 3), -1.0, dtype=torch.float64, device=dev),
                    torch.full((B, Ps), -1.0, dtype=torch.float64, device=dev),
                    torch.zeros((B, Ps), dtype=torch.int32, device=dev),
                    torch.full((B, Ps), -7, dtype=torch.int32, device=dev)]
        a, b = bufs(), bufs()
        for rnd in range(2):
            if P:
                ctx.temporal_sep_dev(dY.data_ptr(), B, 0.9, a[0].data_ptr())
            ctx.gjk_swarm_dev(dY.data_ptr(), B, a[1].data_ptr(), a[2].data_ptr(), a[3].data_ptr(), a[4].data_ptr(),
                              a[5].data_ptr(), a[6].data_ptr(), 128, 300)
            ctx.pair_sweep_dev(dY.data_ptr(), B, 0.9, b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr(), b[3].data_ptr(),
                               b[4].data_ptr(), b[5].data_ptr(), b[6].data_ptr(), 128, 300)
            torch.cuda.synchronize()
            for i, (x, y) in enumerate(zip(a, b)):
                assert np.array_equal(x.cpu().numpy(), y.cpu().numpy(), equal_nan=True), ("fused != separate", trial, rnd, i, N, n, M, B, Ps)
        # every family in one call (planar rows: the dynamics groups inside the sweep's grid) against the separate launches
        dtf = torch.from_numpy(np.linspace(3.0, 9.0, B)).to(dev)
        sp_a = torch.full((B, ctx.len_speed), np.nan, dtype=torch.float64, device=dev)
        an_a = torch.full((B, ctx.len_ang_rate), np.nan, dtype=torch.float64, device=dev)
        sp_b, an_b = torch.full_like(sp_a, np.nan), torch.full_like(an_a, np.nan)
        ctx.dynamics_dev(dY.data_ptr(), dtf.data_ptr(), B, 4.0, True, 1.5, sp_a.data_ptr(), an_a.data_ptr())
        c = bufs()
        ctx.constraint_sweep_dev(dY.data_ptr(), dtf.data_ptr(), B, 0.9, c[0].data_ptr(), 4.0, True, 1.5, sp_b.data_ptr(), an_b.data_ptr(),
                                 c[1].data_ptr(), c[2].data_ptr(), c[3].data_ptr(), c[4].data_ptr(), c[5].data_ptr(), c[6].data_ptr(), 128, 300)
        torch.cuda.synchronize()
        for i, (x, y) in enumerate(zip(a, c)):
            if i == 0 and not P:
                continue
            assert np.array_equal(x.cpu().numpy(), y.cpu().numpy(), equal_nan=True), ("constraint sweep != separate", trial, i, N, n, M, B)
        assert torch.equal(sp_a.view(torch.uint8), sp_b.view(torch.uint8)) and torch.equal(an_a.view(torch.uint8), an_b.view(torch.uint8)), ("dynamics", trial, N, n)
        fl, ns, st, di = (a[1].cpu().numpy(), a[5].cpu().numpy(), a[6].cpu().numpy(), a[4].cpu().numpy())
        c1, c2 = a[2].cpu().numpy(), a[3].cpu().numpy()
        for r in range(B):
            o = oracle.gjk_pairs(*synth.pack_polys(synth.hulls_from_Y(Yb[r], 2) + polys), pa, pb, md_cap=300, nthreads=8)
            assert (fl[r] == o["flag"]).all() and (ns[r] == o["n_support"]).all() and (st[r] == o["status"]).all(), ("gjk", trial, N, n, M)
            sep = (o["flag"] == 1) & (o["status"] == 0)
            for got, ref in ((di[r], o["dist"]), (c1[r], o["c1"]), (c2[r], o["c2"])):
                if sep.any():
                    assert np.array_equal(got[sep], ref[sep]), ("dist / closest points differ from the oracle", trial)
        if P:
            ref_sep, _, _ = oracle.eval_batch(Yb, 10.0, N, 2, 0, 0.9, 5.0, 1.0, want=("sep",), nthreads=8)
            got = a[0].cpu().numpy()
            err = np.max(np.abs(got - ref_sep) / np.maximum(1.0, np.abs(ref_sep)))
            assert err < 1e-9, ("sep", trial, err)
        ctx.use_own_stream()
        ctx.close()
        print("trial %d ok: N=%d n=%d M=%d B=%d hull pairs=%d%s%s  (%.0f s)" % (trial, N, n, M, B, Ps, " tiled-size" if big else "",
              " partial" if partial else "", time.time() - t0), flush=True)
    print("stress ok: %d trials" % trials)


def families(trials, seed):
    """Bernstein families (temporal separation, speed, angular rate; any DEG_ELEV, with and without point obstacles)
    and the 3-D sweep, random shapes, against the oracle."""
    rng = np.random.default_rng(seed)
    t0 = time.time()
    for trial in range(trials):
        d = int(rng.choice([2, 2, 3]))
        n = int(rng.choice([3, 4, 5, 7, 9, 10, 12, 15, 20]))
        R = int(rng.choice([0, 0, 1, 3, 10, 30, 100]))
        N = int(rng.integers(1, 90))
        n_obs = int(rng.choice([0, 0, 3]))
        B = int(rng.choice([1, 2, 7]))
        Y = synth.swarm_control_points(N, d, n, seed=500 + trial)
        Yb = synth.fd_batch(Y, B=B, h=0.25) if B > 1 else Y[None].copy()
        obs = rng.uniform(0, 100, size=(n_obs, d)) if n_obs else None
        tf = rng.uniform(5, 20, size=B)
        ctx = capi.Context(N, d, n, R, point_obs=obs)
        Yo = Yb if not n_obs else np.concatenate([Yb, np.broadcast_to(np.repeat(obs.reshape(-1)[:, None], n + 1, 1)[None], (B, n_obs * d, n + 1))], axis=1)
        nveh_o = N + n_obs
        r_sep, _, _ = oracle.eval_batch(Yo, tf, nveh_o, d, R, 0.9, 5.0, 1.0, want=("sep",), nthreads=8) if nveh_o > 1 else (None, None, None)
        _, r_sp, r_an = oracle.eval_batch(Yb, tf, N, d, R, 0.9, 5.0, 1.0, want=("speed", "ang") if d == 2 else ("speed",), nthreads=8)

        def close(got, ref, what, tol=1e-9):
            got, ref = np.asarray(got), np.asarray(ref)
            assert got.shape == ref.shape, (what, got.shape, ref.shape)
            fin = np.isfinite(ref)
            assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(np.isinf(got), np.isinf(ref)), (what, trial)
            if fin.any():
                err = np.max(np.abs(got[fin] - ref[fin]) / np.maximum(1.0, np.abs(ref[fin])))
                assert err < tol, (what, trial, err, N, d, n, R)
        if nveh_o > 1:
            close(ctx.temporal_sep(Yb, 0.9), r_sep, "sep")
        close(ctx.speed(Yb, tf, 5.0, True), r_sp, "speed")
        if d == 2:
            # with DEG_ELEV > 0 near-singular elements (vehicle almost at rest) differ by up to ~1e-8 between any two
            # orders of operations, the oracle's included (tools/angrate_conditioning.py)
            close(ctx.ang_rate(Yb, tf, 1.0), r_an, "ang", 1e-9 if R == 0 else 5e-8)
        ctx.close()
        print("families trial %d ok: N=%d d=%d n=%d R=%d obs=%d B=%d (%.0f s)" % (trial, N, d, n, R, n_obs, B, time.time() - t0), flush=True)
    print("families ok: %d trials" % trials)


def gjk3d(trials, seed):
    """The 3-D sweep (and, for shapes outside it, the general kernel) on random 3-D swarms with 3-D polygons against
    the oracle: flags, statuses (incl. the cycle detector's), support counts exact; distances and closest points identical (round 5: csrc/libm_pow2.h)."""
    rng = np.random.default_rng(seed)
    t0 = time.time()
    for trial in range(trials):
        n = int(rng.choice([3, 5, 7, 10, 15, 4, 9]))
        N = int(rng.integers(2, 60))
        M = int(rng.integers(0, 5))
        B = int(rng.choice([1, 2, 5, 40]))
        Y = synth.swarm_control_points(N, 3, n, seed=900 + trial)
        Yb = synth.fd_batch(Y, B=B, h=0.3) if B > 1 else Y[None].copy()
        polys = []
        for _ in range(M):
            K = int(rng.integers(3, n + 2))
            polys.append(rng.uniform(0, 100, size=(1, 3)) + rng.normal(0, 6.0, size=(K, 3)))
        pa, pb = synth.swarm_pairs(N, M)
        ctx = capi.Context(N, 3, n, 0)
        if M:
            ctx.set_polygons(*synth.pack_polys(polys))
        ctx.set_hull_pairs(pa, pb)
        for rnd in range(2):
            r = ctx.gjk_swarm(Yb, md_cap=300)
            for b in range(0, B, max(1, B // 3)):
                o = oracle.gjk_pairs(*synth.pack_polys(synth.hulls_from_Y(Yb[b], 3) + polys), pa, pb, md_cap=300, nthreads=8)
                assert (r["flag"][b] == o["flag"]).all() and (r["n_support"][b] == o["n_support"]).all() and \
                       (r["status"][b] == o["status"]).all(), ("gjk3d", trial, rnd, N, n, M, B)
                sep = (o["flag"] == 1) & (o["status"] == 0)
                for key in ("dist", "c1", "c2"):
                    if sep.any():
                        assert np.array_equal(r[key][b][sep], o[key][sep]), (key, "differs from the oracle", trial)
        ctx.close()
        print("gjk3d trial %d ok: N=%d n=%d M=%d B=%d (%.0f s)" % (trial, N, n, M, B, time.time() - t0), flush=True)
    print("gjk3d ok: %d trials" % trials)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "gjk3d":
        gjk3d(int(sys.argv[2]) if len(sys.argv) > 2 else 40, int(sys.argv[3]) if len(sys.argv) > 3 else 11)
    elif len(sys.argv) > 1 and sys.argv[1] == "families":
        families(int(sys.argv[2]) if len(sys.argv) > 2 else 40, int(sys.argv[3]) if len(sys.argv) > 3 else 7)
    else:
        main()
