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
        fl, ns, st, di = (a[1].cpu().numpy(), a[5].cpu().numpy(), a[6].cpu().numpy(), a[4].cpu().numpy())
        c1, c2 = a[2].cpu().numpy(), a[3].cpu().numpy()
        for r in range(B):
            o = oracle.gjk_pairs(*synth.pack_polys(synth.hulls_from_Y(Yb[r], 2) + polys), pa, pb, md_cap=300, nthreads=8)
            assert (fl[r] == o["flag"]).all() and (ns[r] == o["n_support"]).all() and (st[r] == o["status"]).all(), ("gjk", trial, N, n, M)
            sep = (o["flag"] == 1) & (o["status"] == 0)
            for got, ref in ((di[r], o["dist"]), (c1[r], o["c1"]), (c2[r], o["c2"])):
                if sep.any():
                    assert np.max(np.abs(got[sep] - ref[sep]) / np.maximum(1.0, np.abs(ref[sep]))) < 1e-12, ("dist", trial)
        if P:
            ref_sep, _, _ = oracle.eval_batch(Yb, 10.0, N, 2, 0, 0.9, 5.0, 1.0, want=("sep",), nthreads=8)
            got = a[0].cpu().numpy()
            err = np.max(np.abs(got - ref_sep) / np.maximum(1.0, np.abs(ref_sep)))
            assert err < 1e-9, ("sep", trial, err)
        ctx.set_stream(0)
        ctx.close()
        print("trial %d ok: N=%d n=%d M=%d B=%d hull pairs=%d%s%s  (%.0f s)" % (trial, N, n, M, B, Ps, " tiled-size" if big else "",
              " partial" if partial else "", time.time() - t0), flush=True)
    print("stress ok: %d trials" % trials)


if __name__ == "__main__":
    main()
