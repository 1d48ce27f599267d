"""The N>1 orchestration on CPU: world_size-2 gloo processes run the pair-partitioned sweep and
the batch-sharded sweep with the CPU oracle injected as the evaluator (tests may use the oracle;
the product's evaluator is the HIP path), and must reproduce the single-process result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from optimalbeziertrajectorygeneration_amd import synth
        from optimalbeziertrajectorygeneration_amd.distributed import PairPartitionedSweep, shard_rows
        from oracle import oracle as O
        N, d, n, R, B = 11, 2, 6, 2, 5           # 55 pairs: an odd count, so the blocks are ragged
        L = 2 * n + R + 1
        Y = synth.swarm_control_points(N, d, n, seed=3)
        Yb = synth.fd_batch(Y, B=B)
        full = np.stack([O.temporal_sep(Yb[b], N, d, R, 0.9) for b in range(B)])

        def evaluate(begin, count):      # oracle restricted to a block of the lexicographic pair list
            return torch.from_numpy(full.reshape(B, -1, L)[:, begin:begin + count].reshape(B, -1).copy())

        sweep = PairPartitionedSweep(N * (N - 1) // 2, L)
        got = sweep.run(evaluate, B, torch.device("cpu"))
        ok1 = bool(np.array_equal(got.numpy(), full))
        # per-pair minima variant (width 1)
        mins = full.reshape(B, -1, L).min(axis=2)
        sweep1 = PairPartitionedSweep(N * (N - 1) // 2, 1)
        got1 = sweep1.run(lambda b, c: torch.from_numpy(mins[:, b:b + c].copy()), B, torch.device("cpu"))
        ok2 = bool(np.array_equal(got1.numpy(), mins))
        # batch sharding: no collective on the data path; gather only to check
        rb, rc = shard_rows(B, world, rank)
        mine = torch.from_numpy(full[rb:rb + rc].copy())
        parts = [None] * world
        dist.all_gather_object(parts, (rb, mine.numpy()))
        glued = np.concatenate([p for _, p in sorted(parts, key=lambda t: t[0])])
        ok3 = bool(np.array_equal(glued, full))
        # the hull family partitioned the same way: gjkNew's (dist float64, flag int32) per pair and the separation
        # minima travel in ONE byte-packed all-gather (what bench.py --mode pairs does with the HIP evaluators)
        from optimalbeziertrajectorygeneration_amd.distributed import all_gather_pair_blocks, partition
        polys = synth.polygon_obstacles(3, seed=3)
        pa, pb = synth.swarm_pairs(N, 3)                       # 55 + 33 = 88 hull pairs
        g = [O.gjk_pairs(*synth.pack_polys(synth.hulls_from_Y(Yb[b], d) + polys), pa, pb, md_cap=500) for b in range(B)]
        g_dist = np.stack([x["dist"] for x in g])
        g_flag = np.stack([x["flag"] for x in g])
        hb = partition(len(pa), world)
        h0, hc = hb[rank]
        s0, sc = sweep1.my_block
        parts = [(torch.from_numpy(mins[:, s0:s0 + sc].copy()), sweep1.blocks, 1),
                 (torch.from_numpy(g_dist[:, h0:h0 + hc].copy()), hb, 1),
                 (torch.from_numpy(g_flag[:, h0:h0 + hc].copy()), hb, 1)]
        a_min, a_dist, a_flag = all_gather_pair_blocks(parts)
        ok4 = bool(np.array_equal(a_min.numpy(), mins) and np.array_equal(a_dist.numpy(), g_dist, equal_nan=True)
                   and np.array_equal(a_flag.numpy(), g_flag) and a_flag.dtype == torch.int32)
        q.put((rank, ok1, ok2, ok3 and ok4, sweep.blocks))
    finally:
        dist.destroy_process_group()


def test_partition_helpers():
    from optimalbeziertrajectorygeneration_amd.distributed import partition, shard_rows
    assert partition(10, 3) == [(0, 4), (4, 3), (7, 3)]
    assert partition(2, 4) == [(0, 1), (1, 1), (2, 0), (2, 0)]
    assert partition(32640, 8)[0] == (0, 4080) and sum(c for _, c in partition(32640, 8)) == 32640
    assert shard_rows(1153, 8, 7) == (1009, 144)


def test_row_sharded_fd_step_layout():
    """RowShardedFdStep's host side (no GPU): the ranks' blocks tile the n_x + 1 rows, ranks beyond the rows get empty
    blocks and `run` leaves them alone, world / rank default to "no process group"."""
    from optimalbeziertrajectorygeneration_amd.distributed import RowShardedFdStep
    steps = [RowShardedFdStep(1153, world=8, rank=r) for r in range(8)]
    assert [s.count for s in steps] == [145] + [144] * 7 and steps[0].begin == 0
    assert all(a.begin + a.count == b.begin for a, b in zip(steps, steps[1:])) and steps[-1].begin + steps[-1].count == 1153
    lone = RowShardedFdStep(7)
    assert (lone.world, lone.rank, lone.begin, lone.count) == (1, 0, 0, 7)
    empty = RowShardedFdStep(3, world=5, rank=4)
    assert empty.count == 0
    sentinel = {"ang": None}
    assert empty.run(None, 0, 1, 1e-3, 0, 0.9, 5.0, True, 1.0, sentinel) is sentinel and empty.strategy is None


def test_pair_partition_and_batch_sharding_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok1, ok2, ok3, blocks in res:
        assert ok1 and ok2 and ok3, "rank %d" % rank
        assert blocks == [(0, 28), (28, 27)]


def _worker8(rank, world, port, q):
    """world_size 8 on C4's list sizes: 32 640 temporal-separation pairs + a 32 640-pair hull list, synthetic per-pair
    values (this test is about the partition and the packed all-gather, not about the evaluators), and a short list of
    5 pairs that leaves three ranks with EMPTY blocks."""
    import sys
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from optimalbeziertrajectorygeneration_amd.distributed import (PairPartitionedSweep, all_gather_pair_blocks, partition,
                                                                       shard_rows)
        B, P = 3, 32640
        rng = np.random.default_rng(5)
        mins = rng.normal(size=(B, P))
        g_dist = rng.uniform(0.0, 9.0, size=(B, P)); g_dist[:, ::7] = np.nan
        g_flag = rng.integers(-1, 2, size=(B, P)).astype(np.int32)
        tiny = rng.normal(size=(B, 5))                              # 5 pairs over 8 ranks: ranks 5..7 own nothing
        sweep = PairPartitionedSweep(P, 1)
        hb, tb = partition(P, world), partition(5, world)
        s0, sc = sweep.my_block
        h0, hc = hb[rank]
        t0, tc = tb[rank]
        parts = [(torch.from_numpy(mins[:, s0:s0 + sc].copy()), sweep.blocks, 1),
                 (torch.from_numpy(g_dist[:, h0:h0 + hc].copy()), hb, 1),
                 (torch.from_numpy(g_flag[:, h0:h0 + hc].copy()), hb, 1),
                 (torch.from_numpy(tiny[:, t0:t0 + tc].copy()), tb, 1)]
        a_min, a_dist, a_flag, a_tiny = all_gather_pair_blocks(parts)
        ok = bool(np.array_equal(a_min.numpy(), mins) and np.array_equal(a_dist.numpy(), g_dist, equal_nan=True)
                  and np.array_equal(a_flag.numpy(), g_flag) and np.array_equal(a_tiny.numpy(), tiny))
        # width > 1 through PairPartitionedSweep.run, ragged: 31 doubles per pair as at C4 (degree 15)
        full = rng.normal(size=(2, 1000 * 31))
        sw = PairPartitionedSweep(1000, 31)
        got = sw.run(lambda b, c: torch.from_numpy(full.reshape(2, 1000, 31)[:, b:b + c].reshape(2, -1).copy()), 2,
                     torch.device("cpu"))
        ok = ok and bool(np.array_equal(got.numpy(), full))
        q.put((rank, ok, sweep.blocks[rank], tb[rank], shard_rows(7169, world, rank)))
    finally:
        dist.destroy_process_group()


def test_pair_partition_world8_with_empty_blocks():
    """The first real 8-GPU run must not be the first 8-way run: partition + packed all-gather at world size 8."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _, _ in res)
    assert [b for _, _, b, _, _ in res] == [(4080 * r, 4080) for r in range(8)]
    assert [t for _, _, _, t, _ in res] == [(0, 1), (1, 1), (2, 1), (3, 1), (4, 1), (5, 0), (5, 0), (5, 0)]
    rows = [s for _, _, _, _, s in res]
    assert sum(c for _, c in rows) == 7169 and rows[0] == (0, 897) and rows[7][0] + rows[7][1] == 7169


def _worker_sparse(rank, world, port, q):
    """The reduced gather of a row-sharded finite-difference step: per owned row only the minima of the pairs its vehicle
    touches + row 0's minima by broadcast (distributed.SparseMinimaGather), oracle as the evaluator; with point obstacles
    (objects that no row advances) and a row count that leaves the blocks ragged."""
    import sys
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from optimalbeziertrajectorygeneration_amd import synth
        from optimalbeziertrajectorygeneration_amd.distributed import SparseMinimaGather
        from oracle import oracle as O
        N, d, n, R = 6, 2, 5, 1
        obs = [[20.0, 30.0], [55.0, 41.0]]
        n_obj, L = N + len(obs), 2 * n + R + 1
        Y = synth.swarm_control_points(N, d, n, seed=9)
        B = N * d * (n - 1) + 1                                  # 49 rows: 25 + 24 over two ranks
        Yb = synth.fd_batch(Y, B=B, h=1e-3)

        def minima(row):                                         # the class path's rows: obstacles as constant curves
            yo = np.vstack([Yb[row]] + [np.full((1, n + 1), v) for o in obs for v in o])
            return O.temporal_sep(yo, n_obj, d, R, 0.9).reshape(-1, L).min(axis=1)
        dense = np.stack([minima(b) for b in range(B)])
        g = SparseMinimaGather(B, N, n_obj, d, n - 1)
        sparse, row0 = g.exchange(torch.from_numpy(dense[g.begin:g.begin + g.count].copy()))
        ok = tuple(sparse.shape) == (B, n_obj - 1) and bool(np.array_equal(row0.numpy(), dense[0]))
        rebuilt = g.dense_rows(sparse, row0, list(range(B))).numpy()
        ok = ok and bool(np.array_equal(rebuilt, dense))         # every row of the dense block, on every rank
        # and it really is reduced: what travelled is B x (n_obj - 1) (+ padding) + P doubles, the dense block B x P
        P = n_obj * (n_obj - 1) // 2
        ok = ok and g.bytes_per_step == 8 * world * (g.max_count * (n_obj - 1) + P) and g.bytes_per_step < 8 * B * P / 2
        q.put((rank, ok, (g.begin, g.count)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sparse_minima_gather(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sparse, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert [b for _, _, b in res] == ([(0, 25), (25, 24)] if world == 2 else [(0, 17), (17, 16), (33, 16)])


def _run_bench(args, env_extra=None, timeout=300):
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=timeout, universal_newlines=True)


def test_bench_launcher_never_prints_a_multi_gpu_line_from_one_process():
    """`python bench.py --gpus N` starts N rank processes itself; a rank count that does not match --gpus is refused,
    and without a GPU every rank fails loudly (no CPU fallback) -> non-zero exit and NO JSON line."""
    r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "refusing" in r.stderr and r.stdout.strip() == ""
    if torch.cuda.is_available():
        pytest.skip("the spawn path is exercised with real ranks in tests/test_gpu_multirank.py")
    r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu"])
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert r.stderr.count("no HIP device visible") == 2          # both ranks were started and both refused


def test_bench_launcher_returns_the_failing_ranks_code():
    """A rank that exits non-zero while the others still wait (as they would in a barrier): the launcher stops them and
    returns THAT rank's code with its message -- whichever position the rank has in the launcher's iteration order."""
    import time
    for failing in (0, 1, 2):
        t0 = time.time()
        r = _run_bench(["--gpus", "3", "--steps", "1", "--warmup", "1", "--no-cpu"],
                       {"OBTG_BENCH_REHEARSE_EXIT": "%d:7" % failing}, timeout=60)
        assert r.returncode == 7, (failing, r.returncode, r.stderr)
        assert "a rank exited with code 7" in r.stderr and "Traceback" not in r.stderr
        assert time.time() - t0 < 30          # the waiting ranks were stopped, not waited for
