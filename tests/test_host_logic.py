"""Host-side logic of the look-alike problem layer (no GPU needed): x <-> y layout,
initial guesses, finite-difference row construction -- against values captured from the
reference (tests/golden/problem.npz)."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def P(golden_dir):
    return np.load(golden_dir + "/problem.npz")


def _example1():
    from optimalbeziertrajectorygeneration_amd.optimization import BezOptimization
    return BezOptimization(numVeh=2, dimension=2, degree=10, minimizeGoal='TimeOpt', maxSep=1, maxSpeed=5,
                           maxAngRate=1, initPoints=[(0, 5), (3, 0)], finalPoints=[(8, 4), (7, 10)],
                           initSpeeds=[1] * 2, finalSpeeds=[1] * 2, initAngs=[0, np.pi / 2],
                           finalAngs=[0, np.pi / 2], pointObstacles=[[3, 2], [6, 7]])


def test_example1_guess_and_reshape(P):
    bo = _example1()
    xg = bo.generateGuess(std=0)
    assert np.array_equal(xg, P["ex1_xguess"])
    assert np.array_equal(bo.reshapeVector(xg), P["ex1_yguess"])
    assert np.array_equal(bo.reshapeVector(P["ex1_x"]), P["ex1_y"])
    assert np.allclose(bo.reshapeVector(xg)[0, :3], [0, 0.1, 1.075])
    assert np.array_equal(bo.generateGuess(std=0.5, seed=3), P["ex1_xguess_std"])
    assert bo.objectiveFunction(P["ex1_x"]) == float(P["ex1_obj"]) == 3.7
    assert bo.model['numVeh'] == 2 and bo.pointObstacles == [[3, 2], [6, 7]]


def test_swarm_and_fixed_tf_layouts(P):
    from optimalbeziertrajectorygeneration_amd.optimization import BezOptimization
    bs = BezOptimization(numVeh=36, dimension=3, degree=5, minimizeGoal='Euclidean', maxSep=0.9,
                         initPoints=P["sw_init"], finalPoints=P["sw_final"])
    assert np.array_equal(bs.reshapeVector(P["sw_xguess"]), P["sw_yguess"])
    assert np.array_equal(bs.reshapeVector(P["sw_x"]), P["sw_y"])
    assert np.array_equal(bs.generateGuess(std=0), P["sw_genguess"])
    bf = BezOptimization(numVeh=3, dimension=2, degree=7, minimizeGoal='Euclidean', maxSep=0.5, maxSpeed=4,
                         minSpeed=0.2, maxAngRate=2, initPoints=[(0, 0), (1, 5), (9, 2)],
                         finalPoints=[(10, 1), (8, 8), (0, 7)], initSpeeds=[1, 2, 0.5], finalSpeeds=[1, 1, 2],
                         initAngs=[0.1, -0.4, 2.0], finalAngs=[0.3, 0.0, 2.5], tf=7.0)
    assert np.array_equal(bf.generateGuess(std=0.4, seed=11), P["fx_x"])
    assert np.array_equal(bf.reshapeVector(P["fx_x"]), P["fx_y"])
    bm = BezOptimization(numVeh=2, dimension=2, degree=5, initPoints=np.array([[1, 2], [3, 4]]),
                         finalPoints=np.array([[5, 6], [7, 8]]), initSpeeds=np.array([3, 3]),
                         finalSpeeds=np.array([10, 10]), initAngs=np.array([np.pi / 2, np.pi / 2]),
                         finalAngs=np.array([0, 0]), pointObstacles=[[1, 2], [3, 4]])
    assert np.array_equal(bm.reshapeVector(P["main_x"]), P["main_y"])
    assert np.array_equal(bm.generateGuess(), P["main_guess"])


def test_batched_reshape_equals_rowwise(P):
    bo = _example1()
    X, dx = bo._fd_rows(P["ex1_x"])
    assert X.shape == (30, 29) and np.all(dx > 0)
    Y = bo.reshapeVectors(X)
    for k in (0, 1, 17, 29):
        assert np.array_equal(Y[k], bo.reshapeVector(X[k]))
    # the tf row moves the speed columns of every vehicle (optimization.py:276-281)
    assert (Y[29, :, 1] != Y[0, :, 1]).any() and np.array_equal(Y[29, :, 2:-2], Y[0, :, 2:-2])


def test_invalid_goal_raises():
    from optimalbeziertrajectorygeneration_amd.optimization import BezOptimization
    bo = BezOptimization(numVeh=1, dimension=2, degree=5, minimizeGoal='nope', initPoints=[(0, 0)],
                         finalPoints=[(1, 1)])
    with pytest.raises(ValueError):
        bo.objectiveFunction


def test_bezier_container_semantics():
    from optimalbeziertrajectorygeneration_amd.bezier import Bezier, RationalBezier
    c = Bezier([[0, 1, 2], [3, 4, 5]], tf=2.5)          # array-likes are accepted
    assert c.dim == c.dimension == 2 and c.deg == c.degree == 2 and c.t0 == 0.0 and c.tf == 2.5
    assert c.x.cpts.shape == (1, 3) and c.z is None and np.array_equal(c.y.cpts, [[3, 4, 5]])
    assert c.tau.shape == (1001,) and c.tau[-1] == 2.5
    d = Bezier(np.array([1.0, 2.0, 4.0]))                # 1-D input is promoted (bezier.py:58-61)
    assert d.dim == 1 and d.deg == 2
    s = c - Bezier(np.ones((2, 3)), tf=2.5)
    assert np.array_equal(s.cpts, np.array([[-1, 0, 1], [2, 3, 4]], float)) and s.tf == 2.5
    assert np.array_equal((c + c).cpts, 2 * c.cpts)
    with pytest.raises(NotImplementedError):
        c - Bezier(np.ones((2, 3)), tf=1.0)
    with pytest.raises(TypeError):
        c.mul(3)
    with pytest.raises(ValueError):
        c.mul(d)
    with pytest.raises(ValueError):
        d.minDist(d)
    r = RationalBezier(np.ones((1, 3)), np.ones((1, 3)))
    assert r.deg == 2


def test_example_drivers_constructor_forms(golden_dir):
    """The constructor forms of the three drivers added in round 5 (drivers.npz, written by the reference): degree 8 with
    one-element lists (Examples/DubinsCarTimeOptimal.py:70-96, DubinsCarExample2.py:83-104 with a fixed-tf keyword that the
    time-optimal goal overrides) and DrivingOnATrack.py:25-39's scalars / bare tuples (atleast_1d / atleast_2d promote them)."""
    from optimalbeziertrajectorygeneration_amd.optimization import BezOptimization
    g = np.load(golden_dir + "/drivers.npz")
    tt = BezOptimization(numVeh=1, dimension=2, degree=8, minimizeGoal='TimeOpt', maxSep=1, maxSpeed=5, maxAngRate=1,
                         initPoints=[(3, 0)], finalPoints=[(7, 10)], initSpeeds=[1], finalSpeeds=[1], initAngs=[np.pi / 2],
                         finalAngs=[np.pi / 2], pointObstacles=[[3, 2], [6, 7]])
    e2 = BezOptimization(numVeh=1, dimension=2, degree=8, minimizeGoal='TimeOpt', maxSep=1, maxSpeed=3, maxAngRate=np.pi / 2,
                         initPoints=[(0, 0)], finalPoints=[(12, 8)], initSpeeds=[1], finalSpeeds=[1], tf=8, initAngs=[np.pi / 2],
                         finalAngs=[0], pointObstacles=[(3, 2), (7, 6), (9, 9), (4, 5), (5, 8), (3, 7), (7, 3)])
    for pre, bo in (("tt", tt), ("e2", e2)):
        assert np.array_equal(bo.generateGuess(std=0), g[pre + "_x"][0])
        assert np.array_equal(bo.generateGuess(std=0.3, seed=7), g[pre + "_x"][1])
        for k in range(2):
            assert np.array_equal(bo.reshapeVector(g[pre + "_x"][k]), g[pre + "_y"][k])
        for a in range(1, len(g[pre + "_R0_flow_x0"])):            # the retry loop's seeded, noisier guesses
            assert np.array_equal(bo.generateGuess(std=a, seed=100 + a), g[pre + "_R0_flow_x0"][a])
    tr = BezOptimization(numVeh=1, dimension=2, degree=10, minimizeGoal='TimeOpt', maxSep=0.5, maxSpeed=5, maxAngRate=0.5,
                         initPoints=(2, 1), finalPoints=(12, 9), initSpeeds=1, finalSpeeds=1, initAngs=np.pi / 2,
                         finalAngs=np.pi / 2, shapeObstacles=[])
    xg = tr.generateGuess()
    xg[-1] = 10
    assert np.array_equal(xg, g["tr_x"]) and np.array_equal(tr.reshapeVector(xg), g["tr_y"])


def test_default_device_follows_the_launcher(monkeypatch):
    """One process per GPU: when nothing names a device, the scratch context, the shape contexts and BezOptimization take
    OBTG_DEVICE, else torchrun's LOCAL_RANK (wrapped to the devices present, so that several ranks can rehearse on one card)."""
    from optimalbeziertrajectorygeneration_amd import _capi
    n = max(_capi.device_count(), 1)
    monkeypatch.delenv("OBTG_DEVICE", raising=False)
    monkeypatch.delenv("LOCAL_RANK", raising=False)
    assert _capi.default_device() == 0
    monkeypatch.setenv("LOCAL_RANK", "5")
    assert _capi.default_device() == 5 % n
    monkeypatch.setenv("OBTG_DEVICE", "3")
    assert _capi.default_device() == 3 % n
    monkeypatch.setenv("OBTG_DEVICE", "not a number")
    assert _capi.default_device() == 5 % n


def test_spatial_jacobian_call_plan_is_the_loop_form():
    """BezOptimization.spatialSeparationJacobian's one-call plan (optimization._spatial_jac_plan, array form) against the
    loops it replaced: same curve list, same pair list, same (variable, base pair, position) triples -- for rows that move one
    vehicle each, a row that moves all of them (a trailing tf) and a row that moves none."""
    from optimalbeziertrajectorygeneration_amd.optimization import _spatial_jac_plan
    rng = np.random.default_rng(5)
    for (numVeh, dim, K, nobs) in ((5, 2, 6, 3), (4, 3, 4, 0), (1, 2, 11, 2), (7, 2, 9, 1), (1, 2, 5, 0)):      # (the last: no pair at all)
        nvar = numVeh * dim * K
        Y0 = rng.normal(size=(numVeh * dim, K))
        rows = [Y0]
        for k in range(nvar):                        # one coefficient of one vehicle each
            Yk = Y0.copy(); Yk.reshape(-1)[k] += 1e-8
            rows.append(Yk)
        rows.append(Y0 + 1e-8)                       # every vehicle moves
        rows.append(Y0.copy())                       # nothing moves
        two = Y0.copy(); two[0, 0] += 1e-8; two[-1, -1] += 1e-8
        rows.append(two)                             # the first and the last vehicle (the same one when numVeh == 1)
        Y = np.stack(rows)
        obs = [np.vstack((rng.normal(size=(dim, K)), np.zeros((3 - dim, K)))) for _ in range(nobs)]
        stack, pa, pb, P, col, row, pos = _spatial_jac_plan(Y, numVeh, dim, obs)
        # the loop form
        def pad(c):
            out = np.zeros((3, K)); out[:dim] = c
            return out
        n = numVeh + nobs
        curves = [pad(Y[0, i * dim:(i + 1) * dim]) for i in range(numVeh)] + obs
        lpa, lpb = [], []
        for i in range(n):
            for j in range(i + 1, n):
                lpa.append(i); lpb.append(j)
        LP = len(lpa)
        pair_index = {(lpa[q], lpb[q]): q for q in range(LP)}
        trip = []
        Yv = Y.reshape(Y.shape[0], numVeh, dim, K)
        changed = np.any(Yv[1:] != Yv[0], axis=(2, 3))
        for k in range(Y.shape[0] - 1):
            mine = {}
            for v in np.nonzero(changed[k])[0]:
                mine[int(v)] = len(curves)
                curves.append(pad(Y[k + 1, v * dim:(v + 1) * dim]))
            for (i, j), q in pair_index.items():
                if i in mine or j in mine:
                    trip.append((k, q, len(lpa)))
                    lpa.append(mine.get(i, i)); lpb.append(mine.get(j, j))
        assert P == LP
        assert np.array_equal(stack, np.stack(curves))
        assert np.array_equal(pa, lpa) and np.array_equal(pb, lpb)
        assert [tuple(t) for t in zip(col.tolist(), row.tolist(), pos.tolist())] == trip


def test_fd_serving_logic_without_a_device(monkeypatch):
    """BezOptimization._serve on the host alone (the closure and its batch are stand-ins): a sweep of SciPy's forward differences is
    answered from one batch; a step SciPy turned around at a bound, or any other one-variable change, is evaluated directly without
    dropping the batch; a point that differs in several variables is a new base; a changed model entry drops everything; the
    switch and the size limit are honoured."""
    from optimalbeziertrajectorygeneration_amd import optimization as opt
    from optimalbeziertrajectorygeneration_amd.optimization import BezOptimization, FD_STEP
    monkeypatch.delenv("OBTG_FD_BATCHING", raising=False)
    kws = dict(numVeh=2, dimension=2, degree=5, minimizeGoal='Euclidean', maxSep=0.9, initPoints=[[0, 0], [1, 5]], finalPoints=[[9, 9], [8, 0]])
    bo = BezOptimization(**kws)
    calls = {'direct': 0, 'batch': 0}

    def direct(x):
        calls['direct'] += 1
        return np.array([x.sum(), (x * x).sum(), x[0] - x[-1]])

    def fake_fd_values(x, family):
        calls['batch'] += 1
        X, dx = bo._fd_rows(x)
        return np.stack([np.array([r.sum(), (r * r).sum(), r[0] - r[-1]]) for r in X]), dx
    monkeypatch.setattr(bo, "_fd_values", fake_fd_values)
    x0 = bo.generateGuess(std=0.1, seed=1)
    nx = x0.size
    assert np.array_equal(bo._serve('tsep', x0, direct), direct(x0))
    calls['direct'] = 0
    for k in range(nx):                                      # the sweep: one batch, no direct call
        xk = x0.copy(); xk[k] += FD_STEP
        want = np.array([xk.sum(), (xk * xk).sum(), xk[0] - xk[-1]])
        assert np.array_equal(bo._serve('tsep', xk, direct), want)
    assert calls == {'direct': 0, 'batch': 1} and bo.fdBatchingStats['served'] == nx
    got = bo._serve('tsep', x0, direct)                      # the base again: kept
    got[0] = 1e9                                             # (a caller may scribble on what it gets)
    assert calls['direct'] == 0 and bo._serve('tsep', x0, direct)[0] != 1e9
    xb = x0.copy(); xb[3] -= FD_STEP                         # a step turned around at a bound: direct, the batch stays
    bo._serve('tsep', xb, direct)
    xk = x0.copy(); xk[2] += FD_STEP
    bo._serve('tsep', xk, direct)
    assert calls == {'direct': 1, 'batch': 1}
    x1 = x0 + 0.01                                           # the next iterate: a new base, then its own batch
    bo._serve('tsep', x1, direct)
    xk = x1.copy(); xk[0] += FD_STEP
    bo._serve('tsep', xk, direct)
    assert calls == {'direct': 2, 'batch': 2}
    bo.model['maxSep'] = 1.1                                 # a driver edits the model between solves: nothing stale is served
    bo._serve('tsep', xk, direct)
    assert calls['direct'] == 3
    # the limit: a batch that would be too large is not formed
    monkeypatch.setenv("OBTG_FD_BATCH_MB", "0.00001")
    bo2 = BezOptimization(**kws)
    monkeypatch.setattr(bo2, "_fd_values", fake_fd_values)
    bo2._serve('tsep', x0, direct)
    n0 = calls['batch']
    for k in range(3):
        xk = x0.copy(); xk[k] += FD_STEP
        bo2._serve('tsep', xk, direct)
    assert calls['batch'] == n0 and bo2.fdBatchingStats['served'] == 0
    # the switch
    monkeypatch.setenv("OBTG_FD_BATCHING", "0")
    assert BezOptimization(**kws).fdBatching is False and BezOptimization(fdBatching=True, **kws).fdBatching is False
    monkeypatch.delenv("OBTG_FD_BATCHING")
    assert BezOptimization(fdBatching=False, **kws).fdBatching is False


def test_counter_files_carry_their_provenance_and_stale_ones_are_refused(tmp_path):
    """bench.py reports hardware-counter figures (HBM bytes per launch, VALU wave-instructions) from committed rocprofv3
    passes.  VERDICT r5: a line must say where they come from and must not report figures of other kernels.  Every entry of
    profiles/counters.json names the obtg_source_hash of the compile unit it was taken on; bench.counters_for returns an
    entry only for the hash of the RUNNING library, a figure-free {"stale": True} for any other, None where nothing exists.
    And the library's own hashes are those of the sources in the tree (build.unit_hashes)."""
    import importlib.util
    import json
    import os
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(repo, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    path = str(tmp_path / "counters.json")
    entry = {"workload": "C3", "kernel": "pair_sweep", "unit": "gjk_kernels", "source_hash": "00aa11bb22cc33dd", "tree": "deadbeef",
             "source": "profiles/r06_pmc_C3.txt", "hbm_bytes_per_launch": 631815675, "valu_wave_insts": 76.4e6, "valu_busy_frac": 0.75}
    json.dump({"entries": [entry]}, open(path, "w"))
    hit = bench.counters_for("C3", "pair_sweep", "00aa11bb22cc33dd", path=path)
    assert hit["matches_running_library"] is True and hit["hbm_bytes_per_launch"] == 631815675 and hit["source"] == "profiles/r06_pmc_C3.txt"
    stale = bench.counters_for("C3", "pair_sweep", "ffffffffffffffff", path=path)
    assert stale["stale"] is True and stale["matches_running_library"] is False and stale["taken_on_source_hash"] == "00aa11bb22cc33dd"
    assert "hbm_bytes_per_launch" not in stale and "valu_wave_insts" not in stale and "valu_busy_frac" not in stale
    assert bench.counters_for("C3", "pair_sweep", None, path=path)["stale"] is True          # a library that cannot say what it is: nothing reported
    assert bench.counters_for("C4", "pair_sweep", "00aa11bb22cc33dd", path=path) is None
    assert bench.counters_for("C3", "pair_sweep", "00aa11bb22cc33dd", path=str(tmp_path / "absent.json")) is None
    # the committed file: well formed, every entry with its provenance
    committed = os.path.join(repo, "profiles", "counters.json")
    if os.path.exists(committed):
        for e in json.load(open(committed))["entries"]:
            assert e["workload"] and e["kernel"] and len(e["source_hash"]) == 16 and e["unit"] and e["tree"] and e["source"].startswith("profiles/"), e
            assert os.path.exists(os.path.join(repo, e["source"])), e["source"]
    # the library says which sources it was built from, and they are the tree's
    from optimalbeziertrajectorygeneration_amd import _capi, build
    build.build()
    want = build.unit_hashes()
    for unit in ("gjk_kernels", "bern_kernels", "capi", "all"):
        assert _capi.source_hash(unit) == want[unit], unit
    assert _capi.source_hash("no such unit") is None
    assert _capi.libm_pow_matches() is True          # this image's glibc 2.35 is what csrc/libm_pow2.h restates (obtg_libm_pow_matches)
