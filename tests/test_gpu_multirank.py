"""Multi-rank paths on the one-GPU box: the HIP evaluators behind the pair-partitioned sweeps (world_size 1), and
`bench.py --gpus 2` starting its own two ranks (gloo, both on device 0: a rehearsal of the launcher and of the
collective path, not a scaling measurement).  `pytest -m gpu`."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pair_partitioned_sweeps_with_hip_evaluators(oracle):
    """distributed.PairPartitionedSweep over gpu_temporal_sep_evaluator and GpuHullPairSweep with the HIP library as
    the evaluator: blocks of a 3-way partition glue to the unpartitioned result, and that equals the oracle."""
    import torch
    from optimalbeziertrajectorygeneration_amd import _capi, synth
    from optimalbeziertrajectorygeneration_amd.distributed import (GpuHullPairSweep, PairPartitionedSweep,
                                                                   gpu_temporal_sep_evaluator, partition)
    N, d, n, M, B = 20, 2, 10, 4, 6
    Y = synth.swarm_control_points(N, d, n, seed=5)
    Yb = synth.fd_batch(Y, B=B)
    polys = synth.polygon_obstacles(M, seed=5)
    pa, pb = synth.swarm_pairs(N, M)
    ctx = _capi.Context(N, d, n, 0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_polygons(*synth.pack_polys(polys))
    dY = torch.from_numpy(Yb).cuda()
    P, L = ctx.num_pairs, 2 * n + 1
    o_sep = np.stack([oracle.temporal_sep(Yb[b], N, d, 0, 0.9) for b in range(B)])
    for min_only in (False, True):
        ev = gpu_temporal_sep_evaluator(ctx, dY, B, 0.9, min_only=min_only)
        full = PairPartitionedSweep(P, 1 if min_only else L).run(ev, B, dY.device)      # world_size 1: the whole list
        torch.cuda.synchronize()
        w = 1 if min_only else L
        glued = torch.cat([ev(b0, c) for (b0, c) in partition(P, 3)], dim=1)
        torch.cuda.synchronize()
        assert torch.equal(glued, full) and full.shape == (B, P * w)
        ref = o_sep.reshape(B, P, L).min(axis=2) if min_only else o_sep
        assert np.abs(full.cpu().numpy() - ref).max() <= 1e-9 * np.abs(ref).max()
    hull = GpuHullPairSweep(ctx, pa, pb, md_cap=500)
    dist_, flag = hull.run(dY, B)
    torch.cuda.synchronize()
    dist_, flag = dist_.cpu().numpy().copy(), flag.cpu().numpy().copy()
    for b in range(B):
        o = oracle.gjk_pairs(*synth.pack_polys(synth.hulls_from_Y(Yb[b], d) + polys), pa, pb, md_cap=500)
        assert (flag[b] == o["flag"]).all()
        sep = o["flag"] == 1
        assert np.array_equal(dist_[b][sep], o["dist"][sep])          # (identical since round 5: csrc/libm_pow2.h)
        assert np.isnan(dist_[b][~sep]).all()
    # a rank's block alone (what rank 1 of 3 would register and sweep)
    b0, c = partition(len(pa), 3)[1]
    ctx.set_hull_pairs(pa[b0:b0 + c], pb[b0:b0 + c])
    r = ctx.gjk_swarm(Yb, md_cap=500)
    assert np.array_equal(r["flag"], flag[:, b0:b0 + c]) and np.array_equal(r["dist"], dist_[:, b0:b0 + c], equal_nan=True)
    ctx.use_own_stream()
    ctx.close()


def _bench(args, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=timeout, universal_newlines=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` (no torchrun): two child ranks, one JSON line from rank 0 with n_gpus == 2 and the
    whole-job rate (2 x B rows per step)."""
    common = ["--steps", "6", "--warmup", "2", "--no-cpu", "--workload", "C2", "--backend", "gloo", "--one-device"]
    one = _bench(["--gpus", "1", "--steps", "6", "--warmup", "2", "--no-cpu", "--workload", "C2"])
    two = _bench(["--gpus", "2"] + common)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "weak"
    assert two["config"]["ranks_seen"] == 2 and two["config"]["backend"] == "gloo" and len(two["config"]["devices"]) == 2
    B = two["config"]["evals_per_step_per_gpu"]
    assert abs(two["value"] - 2 * B * two["steps"] / (two["ms_per_step"] * 1e-3 * two["steps"])) < 1e-3 * two["value"]


def test_bench_pairs_mode_two_ranks_equals_one_rank():
    """--mode pairs: temporal-separation minima AND gjkNew (dist, flag) of one batch, pair lists split over two
    ranks + one all-gather == the single-rank sweep (identical checksums)."""
    args = ["--mode", "pairs", "--workload", "C3", "--batch", "9", "--steps", "2", "--warmup", "1", "--no-cpu"]
    one = _bench(["--gpus", "1"] + args)
    two = _bench(["--gpus", "2", "--backend", "gloo", "--one-device"] + args)
    assert two["n_gpus"] == 2 and two["scaling"] == "strong"
    c1, c2 = one["config"]["checksum"], two["config"]["checksum"]
    assert c1["gjk_flag_sum"] == c2["gjk_flag_sum"]
    assert c1["sep_min_sum"] == c2["sep_min_sum"] and abs(c1["gjk_dist_nansum"] - c2["gjk_dist_nansum"]) <= 1e-12 * abs(c1["gjk_dist_nansum"])
    # round 5: what travels is what north_star names -- the separation minima, 8 bytes per pair and evaluation; gjkNew's
    # (dist, flag) stay with the rank that computed them unless --gather-gjk asks (then one packed all-gather carries all three)
    ab = two["config"]["allgather_bytes"]
    assert ab["per_evaluation"] == 8 * 2 * 1008 and ab["per_step"] == 9 * ab["per_evaluation"] and "gjkNew" not in ab["what"]
    full = _bench(["--gpus", "2", "--backend", "gloo", "--one-device", "--gather-gjk"] + args)
    assert full["config"]["allgather_bytes"]["per_evaluation"] == 8 * 2 * 1008 + 12 * 2 * 1264
    cf = full["config"]["checksum"]
    assert cf["gjk_flag_sum"] == c1["gjk_flag_sum"] and cf["gjk_dist_nansum"] == c1["gjk_dist_nansum"] and cf["sep_min_sum"] == c1["sep_min_sum"]
    # B = 1, the mode's default: ONE evaluation (a line-search / callback evaluation) with its pair lists over the ranks
    one1 = _bench(["--gpus", "1"] + [a for a in args if a not in ("--batch", "9")])
    two1 = _bench(["--gpus", "2", "--backend", "gloo", "--one-device"] + [a for a in args if a not in ("--batch", "9")])
    assert "B=1 rows" in two1["config"]["workload"] and two1["config"]["allgather_bytes"]["per_step"] == 8 * 2 * 1008
    assert one1["config"]["checksum"]["sep_min_sum"] == two1["config"]["checksum"]["sep_min_sum"]


def test_bench_rows_mode_two_ranks_tile_the_iteration():
    """--mode rows: ONE SLSQP iteration's n_x + 1 rows split over the ranks, each rank a row-range view
    (obtg_fd_view_begin_rows), nothing exchanged on the data path -> `scaling` strong, value = all rows / max time; the
    checksums over every rank's rows equal the one-rank run's."""
    args = ["--mode", "rows", "--workload", "C3", "--batch", "301", "--steps", "3", "--warmup", "1", "--no-cpu", "--no-variants"]
    one = _bench(["--gpus", "1"] + args)
    two = _bench(["--gpus", "2", "--backend", "gloo", "--one-device"] + args)
    three = _bench(["--gpus", "3", "--backend", "gloo", "--one-device", "--gather-minima"] + args)
    for line, g in ((one, 1), (two, 2), (three, 3)):
        c = line["config"]
        assert line["n_gpus"] == g and line["scaling"] == "strong" and c["mode"] == "rows" and c["ranks_seen"] == g
        assert c["rows_per_step_all_ranks"] == 301 and sum(c["checksum"]["rows_per_rank"]) == 301
        assert abs(line["value"] - 301 / (line["ms_per_step"] * 1e-3)) < 2e-3 * line["value"]
        assert c["checksum"]["gjk_flag_sum"] == one["config"]["checksum"]["gjk_flag_sum"]
        for k in ("sep_min_sum", "speed_sum"):
            assert abs(c["checksum"][k] - one["config"]["checksum"][k]) <= 1e-11 * abs(one["config"]["checksum"][k]), k
    assert two["config"]["checksum"]["rows_per_rank"] == [151, 150]
    # the reduced gather (distributed.SparseMinimaGather): per row the 71 pairs of its vehicle + row 0's 2016 minima once;
    # every rank rebuilds dense rows from it and finds its own reduction bit for bit
    c3 = three["config"]
    assert c3["gather_minima"] == "sparse" and c3["gather_check"] is True
    assert c3["allgather_bytes"] == 8 * 3 * (101 * 63 + 2016)              # against 8 x 301 x 2016 = 4.9 MB dense
    dense = _bench(["--gpus", "2", "--backend", "gloo", "--one-device", "--gather-minima", "dense"] + args)
    assert dense["config"]["gather_minima"] == "dense" and dense["config"]["allgather_bytes"] == 8 * 2 * 151 * 2016
    # the same ranges through the structured step (obtg_constraint_sweep_fd_structured_rows_dev): bit-identical rows, so
    # the sums over all ranks' rows are the brute-force sums exactly
    assert one["config"]["rows_structured"] is None                        # (--no-variants)
    args_v = [a for a in args if a != "--no-variants"]
    for g in (1, 2, 3):
        line = _bench(["--gpus", str(g)] + (["--backend", "gloo", "--one-device"] if g > 1 else []) + args_v)
        rs = line["config"]["rows_structured"]
        assert rs["checksum_equals_brute_force"] is True and rs["ms_per_step"] > 0, rs
        assert rs["checksum"]["gjk_flag_sum"] == one["config"]["checksum"]["gjk_flag_sum"]


def test_rccl_path_with_one_rank():
    """The backend the multi-GPU runs use (`nccl` == RCCL), exercised on the one-GPU box: process-group init with a device
    id, barrier, the MAX all-reduce of the timing and -- in pairs mode -- the packed all-gather on device tensors, with a
    single rank (`--force-dist`).  Same checksums as the run without torch.distributed."""
    args = ["--mode", "pairs", "--workload", "C3", "--batch", "9", "--steps", "2", "--warmup", "1", "--no-cpu"]
    plain = _bench(["--gpus", "1"] + args)
    rccl = _bench(["--gpus", "1", "--force-dist", "--backend", "nccl"] + args)
    assert plain["config"]["checksum"] == rccl["config"]["checksum"]
    line = _bench(["--gpus", "1", "--force-dist", "--backend", "nccl", "--steps", "4", "--warmup", "2", "--no-cpu", "--workload", "C2"])
    assert line["n_gpus"] == 1 and line["value"] > 0
    # --mode rows over RCCL (one rank): the row-range view, the reduced gather's all_gather_into_tensor + broadcast on device
    # tensors, the MAX all-reduce of the timing -- the collectives of that mode have now run on the backend the 8-GPU run uses
    rargs = ["--mode", "rows", "--workload", "C3", "--batch", "301", "--steps", "3", "--warmup", "1", "--no-cpu", "--no-variants"]
    r_plain = _bench(["--gpus", "1"] + rargs)
    r_rccl = _bench(["--gpus", "1", "--force-dist", "--backend", "nccl", "--gather-minima"] + rargs)
    assert r_rccl["config"]["backend"] == "nccl" and r_rccl["config"]["gather_check"] is True
    assert r_rccl["config"]["checksum"] == r_plain["config"]["checksum"]


def k_bytes(d):
    return d["config"]["alg_bytes_per_eval"] * d["config"]["evals_per_step_per_gpu"]


def test_bench_all_modes_two_ranks():
    """`bench.py --gpus N --all-modes` -- beside the weak line (every rank its own swarm), the strong lines of
    ONE SLSQP iteration's rows over the ranks with the sparse minima gather, for C3 and for C4, in the same processes and the same
    ONE JSON line (`modes`): one driver command, both curves.  Rehearsed with two gloo ranks sharing the one GPU: each mode's
    checksums over all ranks' rows equal a one-rank run's."""
    two = _bench(["--gpus", "2", "--backend", "gloo", "--one-device", "--all-modes", "--steps", "3", "--warmup", "1", "--no-cpu", "--no-variants"],
                 timeout=900)
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and set(two["modes"]) == {"rows_C3", "rows_C4"}
    for wl, rows in (("C3", 1153), ("C4", 7169)):
        m = two["modes"]["rows_" + wl]
        c = m["config"]
        assert m["scaling"] == "strong" and m["n_gpus"] == 2 and c["mode"] == "rows" and c["ranks_seen"] == 2
        assert c["rows_per_step_all_ranks"] == rows and sum(c["checksum"]["rows_per_rank"]) == rows
        assert c["gather_minima"] == "sparse" and c["gather_check"] is True and c["allgather_bytes"] > 0
        assert abs(m["value"] - rows / (m["ms_per_step"] * 1e-3)) < 2e-3 * m["value"]
    one = _bench(["--gpus", "1", "--mode", "rows", "--workload", "C3", "--steps", "3", "--warmup", "1", "--no-cpu", "--no-variants"])
    c1, c2 = one["config"]["checksum"], two["modes"]["rows_C3"]["config"]["checksum"]
    assert c1["gjk_flag_sum"] == c2["gjk_flag_sum"]
    for k in ("sep_min_sum", "speed_sum"):
        assert abs(c1[k] - c2[k]) <= 1e-11 * abs(c1[k]), k


def test_default_bench_line_keeps_the_contract():
    """`python bench.py` as the driver runs it (shortened): ONE JSON line with the contract's keys, the C3 workload as one
    launch per step, `roofline` and `cpu_baseline` objects complete, and the bench's own check of its device buffers
    against the oracle green."""
    d = _bench(["--steps", "12", "--warmup", "3", "--cpu-seconds", "1.5"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 12 and d["warmup"] == 3 and d["higher_is_better"] is True
    assert d["unit"] == "constraint-evals/s" and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["config"]["workload"].startswith("C3") and d["config"]["launches_per_step"] == 1
    assert abs(d["value"] - d["config"]["evals_per_step_per_gpu"] / (d["ms_per_step"] * 1e-3)) < 2e-3 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "valu_issue") and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["kernel"] == "pair_sweep"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and 0.2 < r["frac"] < 1.0 and "traffic" in r
    # round 6: counters only with their provenance, and only when taken on the kernels that run (bench.counters_for)
    if r["traffic"] is not None:
        ts = r["traffic_source"]
        assert ts["matches_running_library"] is True and ts["source_hash"] and ts["source"].startswith("profiles/")
        assert 0.5 < r["traffic"] / k_bytes(d) < 1.5
    if r["bound"] == "valu_issue":
        assert r["issue"]["busy_frac"] > r["frac"] and r["issue"]["valu_wave_insts"] > 1e6
    # round 6: every other BASELINE configuration in the same line, each a child process with its own timed region
    cf = d["configs"]
    assert set(cf) >= {"C1_text", "C2", "C2_file", "C4", "C5", "C5_mindist"}
    for name in ("C1_text", "C2", "C2_file", "C4", "C5"):
        e = cf[name]
        assert e.get("exit_code") == 0 and e["ms_per_step"] > 0 and e["parity"]["ok"] is True, (name, e)
        assert e["kernel"]["avg_ms"] > 0 and e["kernel"]["alg_bytes_per_launch"] > 0 and e["cpu_baseline"]["value"] > 0, (name, e)
    md = cf["C5_mindist"]
    assert md["exit_code"] == 0 and md["parity"]["ok"] is True
    assert md["jacobian_list"]["pairs"] == 114000 and md["jacobian_list"]["parity_check"]["ok"] is True
    assert md["jacobian_list"]["cpu_baseline"]["value"] > 0 and md["jacobian_list"]["cpu_baseline"]["all_cores"]["cores"] >= 1
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["unit"] == d["unit"] and c["sample"]
    p = d["parity_check"]
    assert p["ok"] is True and p["flags_equal"] is True and p["status_nonok"] == 0 and p["max_rel"] < 1e-9
    assert p["max_rel_elementwise"] < 1e-6 and set(p["per_family_elementwise"]) == set(p["per_family"])
    # roofline.frac follows SURVEY 8(d): the one launch's bytes are the per-eval figure x rows
    k = [k for k in d["kernels"] if k["kernel"] == "pair_sweep"][0]
    assert k["alg_bytes_per_launch"] == d["config"]["alg_bytes_per_eval"] * d["config"]["evals_per_step_per_gpu"]
    assert k["launches"] >= 12                      # short runs: events on every launch of the dominant kernel
    assert d["config"]["ranks_seen"] == 1 and len(d["config"]["devices"]) == 1
    v = d["variants"]
    assert set(v) >= {"history_off", "moving_x", "moving_x_history_off"} and all(x["ms_per_step"] > 0 for x in v.values())
    assert all(0 <= x["spread"] < 0.5 for x in v.values())        # median of three runs, (max - min) / median beside it
    sp = v["fd_structured"]["parity"]                              # the structured step's own buffers against the oracle
    assert sp["ok"] is True and sp["flags_equal"] is True and sp["max_rel"] < 1e-9
    pr = d["strong_scaling_proxy"]["rows"]
    assert [e["G"] for e in pr] == [1, 2, 4, 8] and pr[0]["rows"] == 1153 and pr[3]["rows"] == 145
    assert all(0.05 < e["brute_force_efficiency"] <= 1.05 and e["structured_ms"] > 0 for e in pr)


def test_row_sharded_fd_step_helper():
    """distributed.RowShardedFdStep: the rows of three hand-laid ranks, structured and brute force, glue to the one-rank
    step bit for bit (no process group needed: world / rank passed in; what the ranks of `bench.py --mode rows` do)."""
    import torch
    from optimalbeziertrajectorygeneration_amd import _capi, synth
    from optimalbeziertrajectorygeneration_amd.distributed import RowShardedFdStep
    N, n, M = 20, 10, 3
    Y = synth.swarm_control_points(N, 2, n, seed=8)
    B = N * 2 * (n - 1) + 1
    ctx = _capi.Context(N, 2, n, 0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_polygons(*synth.pack_polys(synth.polygon_obstacles(M, seed=8)))
    pa, pb = synth.swarm_pairs(N, M)
    ctx.set_hull_pairs(pa, pb)
    d0 = torch.from_numpy(Y).cuda()
    tf = torch.full((B,), 7.0, dtype=torch.float64, device="cuda")
    whole = RowShardedFdStep(B)
    assert (whole.world, whole.rank, whole.begin, whole.count) == (1, 0, 0, B)
    ref = whole.run(ctx, d0.data_ptr(), 1, synth.FD_STEP, tf.data_ptr(), 0.9, 5.0, True, 1.0, whole.buffers(ctx, len(pa), "cuda"))
    torch.cuda.synchronize()
    assert whole.strategy == "structured"
    for structured in (True, False):
        parts = []
        for r in range(3):
            st = RowShardedFdStep(B, world=3, rank=r)
            o = st.run(ctx, d0.data_ptr(), 1, synth.FD_STEP, tf[st.begin:st.begin + st.count].contiguous().data_ptr(), 0.9, 5.0, True, 1.0,
                       st.buffers(ctx, len(pa), "cuda"), structured=structured)
            torch.cuda.synchronize()
            parts.append(o)
        for k in ref:
            glued = torch.cat([p[k] for p in parts], dim=0)
            assert torch.equal(glued.view(torch.uint8), ref[k].view(torch.uint8)), (structured, k)
    ctx.use_own_stream()
    ctx.close()
    # 3-D rows (the SwarmOfAerialVehicles shapes): no structured step -- "auto" runs the one-launch brute-force sweep on the
    # row-range view, structured=True says so
    N3, n3 = 9, 5
    Y3 = synth.swarm_control_points(N3, 3, n3, seed=8)
    B3 = N3 * 3 * (n3 - 1) + 1
    c3 = _capi.Context(N3, 3, n3, 0)
    c3.set_stream(torch.cuda.current_stream().cuda_stream)
    c3.set_polygons(None, [0])
    pa3, pb3 = synth.swarm_pairs(N3, 0)
    c3.set_hull_pairs(pa3, pb3)
    d3 = torch.from_numpy(Y3).cuda()
    tf3 = torch.full((B3,), 7.0, dtype=torch.float64, device="cuda")
    w3 = RowShardedFdStep(B3)
    ref3 = w3.run(c3, d3.data_ptr(), 1, synth.FD_STEP, tf3.data_ptr(), 0.9, 5.0, True, 1.0, w3.buffers(c3, len(pa3), "cuda"))
    torch.cuda.synchronize()
    assert w3.strategy == "brute force" and ref3["ang"] is None
    parts = []
    for r in range(2):
        st = RowShardedFdStep(B3, world=2, rank=r)
        parts.append(st.run(c3, d3.data_ptr(), 1, synth.FD_STEP, tf3[st.begin:st.begin + st.count].contiguous().data_ptr(), 0.9, 5.0, True,
                            1.0, st.buffers(c3, len(pa3), "cuda")))
        torch.cuda.synchronize()
    for k in ("sep", "speed", "flag", "dist", "status"):
        assert torch.equal(torch.cat([p[k] for p in parts], dim=0).view(torch.uint8), ref3[k].view(torch.uint8)), k
    import pytest
    with pytest.raises(_capi.ObtgError):
        w3.run(c3, d3.data_ptr(), 1, synth.FD_STEP, tf3.data_ptr(), 0.9, 5.0, True, 1.0, w3.buffers(c3, len(pa3), "cuda"), structured=True)
    c3.use_own_stream()
    c3.close()


def test_collective_behind_the_c_abi():
    """include/obtg.h obtg_comm_*: the RCCL all-gather a caller without PyTorch uses (comm.cpp binds librccl at run time).
    On the one-GPU box: a communicator of ONE rank -- ncclGetUniqueId, ncclCommInitRank and ncclAllGather on the context's
    stream really run -- whose obtg_temporal_sep_min_gather_dev equals the plain per-pair minima bit for bit, inside a
    finite-difference view as well; and, because RCCL refuses two ranks on one device, the partition and the unpacking of
    rank blocks for 3 and 8 ranks with the blocks evaluated one after the other (obtg_pair_block,
    obtg_temporal_sep_min_dev(pair_begin, pair_count), obtg_unpack_pair_blocks_dev): what the G ranks' all-gather delivers."""
    import torch
    from optimalbeziertrajectorygeneration_amd import _capi, synth
    from optimalbeziertrajectorygeneration_amd.distributed import partition
    N, d, n, R, B = 23, 2, 10, 3, 5                      # 253 pairs: ragged over 3 and over 8 ranks
    Y = synth.swarm_control_points(N, d, n, seed=11)
    Yb = synth.fd_batch(Y, B=B, h=1e-3)
    ctx = _capi.Context(N, d, n, R)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    P = ctx.num_pairs
    want = ctx.temporal_sep_min(Yb, 0.9)
    dY = torch.from_numpy(Yb).cuda()
    uid = _capi.Comm.unique_id()
    assert len(uid) == 128
    comm = _capi.Comm(1, 0, uid, device=0)
    out = torch.full((B, P), float("nan"), dtype=torch.float64, device="cuda")
    ctx.temporal_sep_min_gather_dev(comm, dY.data_ptr(), B, 0.9, out.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), want)
    # inside a view (dY = NULL): the rows are formed from x's control points while staging
    d0 = torch.from_numpy(Y).cuda()
    out.fill_(float("nan"))
    ctx.fd_view_begin(d0.data_ptr(), 1, 1e-3, B)
    ctx.temporal_sep_min_gather_dev(comm, None, B, 0.9, out.data_ptr())
    ctx.fd_view_end()
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), want)
    # what a rank of a ROW-sharded step sends (obtg_temporal_sep_fd_min_rows_dev): per row the minima of the pairs its vehicle
    # touches, straight from x's control points -- the entries SparseMinimaGather.compact picks out of the dense block
    from optimalbeziertrajectorygeneration_amd.distributed import SparseMinimaGather
    Bx = N * d * (n - 1) + 1
    dense = ctx.temporal_sep_min(synth.fd_batch(Y, B=Bx, h=1e-3), 0.9)
    for (r0, cnt, world, rank) in ((1, Bx - 1, 1, 0), (150, 57, 3, 1)):
        comp = torch.full((cnt, N - 1), float("nan"), dtype=torch.float64, device="cuda")
        ctx.temporal_sep_fd_min_rows_dev(d0.data_ptr(), 1, 1e-3, r0, cnt, 0.9, comp.data_ptr())
        torch.cuda.synchronize()
        g = SparseMinimaGather(Bx, N, N, d, n - 1, world=1, rank=0)
        ref = g.compact(torch.from_numpy(dense))[r0:r0 + cnt]
        assert np.array_equal(comp.cpu().numpy(), ref.numpy()), (r0, cnt)
    with pytest.raises(_capi.ObtgError):
        ctx.temporal_sep_fd_min_rows_dev(d0.data_ptr(), 1, 1e-3, 0, 3, 0.9, comp.data_ptr())      # row 0 advances nothing
    # the byte-typed primitive
    send = torch.arange(1000, dtype=torch.int32, device="cuda")
    recv = torch.zeros_like(send)
    comm.all_gather_dev(ctx, send.data_ptr(), recv.data_ptr(), send.numel() * 4)
    torch.cuda.synchronize()
    assert torch.equal(send, recv)
    comm.close()
    # G ranks' blocks, evaluated here one after the other, in the layout their all-gather delivers
    for G in (3, 8):
        blocks = [ctx.pair_block(G, r) for r in range(G)]
        assert blocks == partition(P, G)
        cmax = max(c for _, c in blocks)
        recv = torch.full((G, B * cmax), float("nan"), dtype=torch.float64, device="cuda")
        for r, (b0, cnt) in enumerate(blocks):
            ctx.temporal_sep_min_dev(dY.data_ptr(), B, 0.9, recv[r].data_ptr(), b0, cnt)
        rows = torch.full((B, P), float("nan"), dtype=torch.float64, device="cuda")
        ctx.unpack_pair_blocks_dev(recv.data_ptr(), B, G, rows.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(rows.cpu().numpy(), want), G
    ctx.use_own_stream()
    ctx.close()
