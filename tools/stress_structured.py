"""One-off stress run of the structured finite-difference step (obtg_constraint_sweep_fd_structured_dev) over random
shapes -- vehicle counts off the 64-item groups, several groups per row, partial batches, DEG_ELEV on and off, rows with
their own tf -- each against the brute-force sweep of the same view, bit for bit.
    python tools/stress_structured.py [trials] [seed]        (needs an MI355X)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from optimalbeziertrajectorygeneration_amd import _capi as capi, synth


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
    only = int(sys.argv[3]) if len(sys.argv) > 3 else -1          # run this trial alone (same random draws)
    t0 = time.time()
    done = skipped = 0
    for trial in range(trials):
        n = int(rng.choice([3, 5, 7, 10, 15]))
        R = int(rng.choice([0, 0, 0, 2, 7, 30])) if n != 15 else 0
        N = int(rng.integers(2, 200)) if trial % 5 else int(rng.integers(200, 330))
        if R:
            N = min(N, 70)
        M = int(rng.integers(0, 5))
        fixed = int(rng.choice([1, 2])) if n >= 5 else 1
        n_free = N * 2 * (n + 1 - 2 * fixed)
        B = int(rng.choice([1, 2, 3, 9, 64, 65, 130, n_free + 1]))
        B = max(1, min(B, n_free + 1, 400))
        tf = np.full(B, 5.5)
        kind = trial % 3
        if kind == 1:
            tf[rng.integers(0, B, size=max(1, B // 7))] += 0.25
        elif kind == 2:
            tf = np.linspace(3.0, 8.0, B)
        if only >= 0 and trial != only:
            continue
        Y = synth.swarm_control_points(N, 2, n, seed=5000 + trial)
        n_obs = int(rng.integers(0, 4)) if trial % 4 == 3 else 0            # pointObstacles (optimization.py:86-98)
        pobs = 10.0 + 80.0 * rng.random((n_obs, 2)) if n_obs else None
        ctx = capi.Context(N, 2, n, R, point_obs=pobs)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        pa, pb = synth.swarm_pairs(N, M)
        # (hull objects of the planar sweeps hold at most n + 1 points: larger polygons take the general gjkNew kernels,
        # brute force and structured step alike -- tools/stress_sweeps.py clips the same way)
        polys = [q[:min(len(q), n + 1)] for q in synth.polygon_obstacles(M, seed=trial)] if M else []
        ctx.set_polygons(*(synth.pack_polys(polys) if M else (None, [0])))
        ctx.set_hull_pairs(pa, pb)
        d0, dtf = torch.from_numpy(Y).cuda(), torch.from_numpy(tf).cuda()
        P, L, Ps = ctx.num_pairs, 2 * n + R + 1, len(pa)
        f64, i32 = torch.float64, torch.int32

        def bufs():
            def nan(*sh):
                return torch.full(sh, float("nan"), dtype=f64, device="cuda")
            return dict(sep=nan(B, P * L), flag=torch.full((B, Ps), -7, dtype=i32, device="cuda"), p1=nan(B, Ps, 3), p2=nan(B, Ps, 3),
                        dist=nan(B, Ps), ns=torch.full((B, Ps), -7, dtype=i32, device="cuda"),
                        st=torch.full((B, Ps), -7, dtype=i32, device="cuda"), sp=nan(B, ctx.len_speed), an=nan(B, ctx.len_ang_rate))
        a, b = bufs(), bufs()
        h = 1e-3
        what = "N=%d n=%d R=%d M=%d obs=%d fixed=%d B=%d tf kind %d" % (N, n, R, M, n_obs, fixed, B, kind)
        try:
            ctx.constraint_sweep_fd_structured_dev(d0.data_ptr(), fixed, h, dtf.data_ptr(), B, 0.9, b["sep"].data_ptr(), 4.0, True, 1.5,
                                                   b["sp"].data_ptr(), b["an"].data_ptr(), b["flag"].data_ptr(), b["p1"].data_ptr(),
                                                   b["p2"].data_ptr(), b["dist"].data_ptr(), b["ns"].data_ptr(), b["st"].data_ptr(), 128, 500)
        except capi.ObtgError as e:
            skipped += 1
            print("trial %d skipped (%s): %s" % (trial, str(e)[:60], what), flush=True)
            ctx.use_own_stream(); ctx.close()
            continue
        ctx.fd_view_begin(d0.data_ptr(), fixed, h, B)
        ctx.constraint_sweep_dev(None, dtf.data_ptr(), B, 0.9, a["sep"].data_ptr(), 4.0, True, 1.5, a["sp"].data_ptr(),
                                 a["an"].data_ptr(), a["flag"].data_ptr(), a["p1"].data_ptr(), a["p2"].data_ptr(),
                                 a["dist"].data_ptr(), a["ns"].data_ptr(), a["st"].data_ptr(), 128, 500)
        ctx.fd_view_end()
        torch.cuda.synchronize()
        for key in a:
            if not torch.equal(a[key].view(torch.uint8), b[key].view(torch.uint8)):
                bad = (a[key].view(torch.uint8) != b[key].view(torch.uint8)).nonzero()
                raise SystemExit("trial %d MISMATCH in %s (%d bytes, first at %s): %s" % (trial, key, len(bad), bad[0].tolist(), what))
        # ... and a random row RANGE of it (obtg_constraint_sweep_fd_structured_rows_dev): the whole-batch call's rows
        if B >= 2:
            rng2 = np.random.default_rng(7000 + trial)
            r0 = int(rng2.integers(0, B))
            cnt = int(rng2.integers(1, B - r0 + 1))
            dtf_r = dtf[r0:r0 + cnt].contiguous()

            def nanr(*sh):
                return torch.full(sh, float("nan"), dtype=f64, device="cuda")
            c_ = dict(sep=nanr(cnt, P * L), flag=torch.full((cnt, Ps), -7, dtype=i32, device="cuda"), p1=nanr(cnt, Ps, 3), p2=nanr(cnt, Ps, 3),
                      dist=nanr(cnt, Ps), ns=torch.full((cnt, Ps), -7, dtype=i32, device="cuda"),
                      st=torch.full((cnt, Ps), -7, dtype=i32, device="cuda"), sp=nanr(cnt, ctx.len_speed), an=nanr(cnt, ctx.len_ang_rate))
            ctx.constraint_sweep_fd_structured_dev(d0.data_ptr(), fixed, h, dtf_r.data_ptr(), cnt, 0.9, c_["sep"].data_ptr(), 4.0, True, 1.5,
                                                   c_["sp"].data_ptr(), c_["an"].data_ptr(), c_["flag"].data_ptr(), c_["p1"].data_ptr(),
                                                   c_["p2"].data_ptr(), c_["dist"].data_ptr(), c_["ns"].data_ptr(), c_["st"].data_ptr(), 128, 500,
                                                   row_begin=r0)
            torch.cuda.synchronize()
            # (rows with their own tf are compared too: a range whose local row 0 has its own tf streams into the rows that share
            # THAT tf and evaluates the others in full -- the same numbers either way)
            for key in c_:
                if not torch.equal(c_[key].view(torch.uint8), b[key][r0:r0 + cnt].contiguous().view(torch.uint8)):
                    raise SystemExit("trial %d MISMATCH in %s of the row range [%d, %d): %s" % (trial, key, r0, r0 + cnt, what))
        done += 1
        print("trial %d ok: %s  (%.0f s)" % (trial, what, time.time() - t0), flush=True)
        ctx.use_own_stream(); ctx.close()
    print("stress ok: %d trials compared, %d outside the structured step's shapes" % (done, skipped))


if __name__ == "__main__":
    main()
