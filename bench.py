#!/usr/bin/env python3
"""Benchmark of the constraint-evaluation hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C3] [--no-cpu]

Metric (BASELINE.json): constraint-evals/s.  One "constraint-eval" = one evaluation, for
one optimisation vector x, of every constraint family of the configuration: the temporal
separation sweep over all vehicle pairs, max-speed, max-angular-rate, and one gjkNew per
hull pair (vehicle<->vehicle and vehicle<->obstacle; for the curve-obstacle configuration C5
every pair of the 64 + 32 objects, as spatialSeparationConstraints pairs them).  One "step" =
one SLSQP iteration's worth of evaluations: the finite-difference batch B = n_x + 1 rows is
built on the device from x's control points (obtg_fd_batch_dev) and every family is evaluated
on it, all resident in HBM.

Workload (config.workload): C3 = 64 vehicles, 2-D, degree 10, DEG_ELEV 0, 8 polygon
obstacles (BASELINE.json configs[2], the configuration the north-star target is quoted
on).  Multi-GPU (`--gpus N`): this script starts N rank processes itself (or runs as one
rank under torchrun when RANK/WORLD_SIZE are set); swarm instances shard over the ranks with
no data-path collective (each rank evaluates the FD batch of its own seeded swarm) -> "weak"
scaling.  `--mode pairs` is the partitioned form: ONE batch, the pair lists split over the
ranks, one RCCL all-gather of the per-pair minima.

Output: ONE JSON line from rank 0 (contract in the task description) carrying also
  roofline     -- for the kernel that dominates the step's device time: algorithmic bytes
                  per launch / mean launch duration measured with HIP events on the
                  launch stream inside the timed region, against the 8 TB/s HBM peak and
                  against this box's measured copy / fill bandwidth;
  kernels      -- the same figures for every kernel of the step;
  parity_check -- the device buffers of the last timed step against the CPU oracle on a sample
                  of rows (first, last, one per XCD residue);
  cpu_baseline -- the CPU oracle (oracle/, a C port of the reference path) timed on this
                  box's host on a bounded sample of the same workload.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--separate", action="store_true",
                    help="temporal separation and the gjkNew sweep as two launches instead of the one-launch pair sweep")
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="C3", choices=["C2", "C2_file", "C3", "C4", "C5"])
    ap.add_argument("--batch", type=int, default=0, help="rows per step (default n_x + 1)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline and parity_check legs")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-gjk", action="store_true", help="Bernstein sweeps only")
    ap.add_argument("--event-period", type=int, default=0,
                    help="HIP events on every n-th launch of the dominant kernel inside the timed region (default 0 = every "
                         "launch when --steps < 100, so that a 20-step run has 20 samples, every 4th otherwise).  The events ride "
                         "on the dispatch itself (hipExtLaunchKernelGGL start / stop timestamps, no barrier packets of their "
                         "own); a timed launch still does not overlap its neighbours: every launch 0.1886 ms per step, every "
                         "4th 0.1865, none 0.1863 at C3")
    ap.add_argument("--no-variants", action="store_true",
                    help="skip the `variants` legs after the timed region (history off, moving x, structured FD step): "
                         "profile runs, where their launches of the same kernel would blur its average")
    ap.add_argument("--fd-dedup", action="store_true",
                    help="reuse row 0's gjkNew results for bit-identical hull pairs (obtg_ctx_set_fd_dedup); "
                         "NOT the headline number")
    ap.add_argument("--mode", default="batch", choices=["batch", "rows", "pairs", "mindist", "c1text"],
                    help="batch: every rank evaluates its own FD batch, no collective (default, the headline); "
                         "rows: ONE swarm, ONE SLSQP iteration -- its n_x + 1 rows split over the ranks by distributed.shard_rows "
                         "(SURVEY.md 8(e).1), every rank an obtg_fd_view_begin_rows over its range, results left per rank, no "
                         "collective on the data path: `scaling` strong; "
                         "pairs: ONE evaluation (B = 1 unless --batch says otherwise: a line-search / callback evaluation of a swarm "
                         "too large for one GPU's latency budget), the pair lists (temporal separation AND gjkNew hull pairs) "
                         "partitioned over the ranks and the per-pair separation minima all-gathered (RCCL; --gather-gjk adds "
                         "gjkNew's dist / flag) -- the partitioned form of BASELINE.json's 256-vehicle case; an iteration's FD "
                         "batch of that swarm shards by ROWS (--mode rows), DESIGN.md 6")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--materialise", action="store_true",
                    help="write the finite-difference batch to HBM every step (obtg_fd_batch_dev) even when the sweeps "
                         "can form its rows while staging them (obtg_*_fd_dev)")
    ap.add_argument("--streams", type=int, default=1, choices=[1, 2],
                    help="1 (default): every launch on one stream, so that the dominant kernel's HIP-event duration is that "
                         "kernel alone; 2: the dynamics launch (speed + angular rate, latency bound) runs on a second HIP "
                         "stream beside the pair sweep (VALU bound) -- about 5 %% more evals/s, but the sweep's measured "
                         "duration then includes the co-running launch")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal: initialise torch.distributed and run the collectives even with one rank "
                         "(exercises the RCCL path on a one-GPU box)")
    ap.add_argument("--gather-minima", nargs="?", const="sparse", default=None, choices=["sparse", "dense"],
                    help="--mode rows: after every step all-gather the per-(row, pair) separation minima to every rank.  sparse "
                         "(default when the flag is given): per row only the N-1 pairs of the vehicle that row advances, plus row "
                         "0's minima broadcast once -- B x (N-1) x 8 bytes per step (C4: 14.6 MB; distributed.SparseMinimaGather); "
                         "dense: the whole [rows][P] block (C4: 1.87 GB per step)")
    ap.add_argument("--gather-gjk", action="store_true",
                    help="--mode pairs: all-gather gjkNew's (dist, flag) of the partitioned hull pair list as well (12 bytes per "
                         "hull pair and row); the default gathers what north_star names, the separation minima (8 bytes per pair and row)")
    ap.add_argument("--ang-order", default="fast", choices=["fast", "elevate_first", "reference", "exact"],
                    help="DEG_ELEV > 0: obtg_ctx_set_ang_rate_order (fast = the headline; exact = plus the double-double "
                         "recompute of near-stop vehicles' rows)")
    ap.add_argument("--no-proxy", action="store_true", help="skip the strong_scaling_proxy leg of a 1-GPU batch-mode run")
    ap.add_argument("--one-device", action="store_true",
                    help="rehearsal only: every rank uses device 0 (needs --backend gloo)")
    ap.add_argument("--mindist-legs", default=None,
                    help="--mode mindist: comma-separated legs to run (reference_algorithm, jacobian_list, robust, jacobian_list_robust, "
                         "curve_polygon_reference_algorithm, curve_polygon_robust); default all (profile runs pick the kernels they want)")
    ap.add_argument("--no-configs", action="store_true",
                    help="1-GPU C3 batch-mode run: skip the `configs` block (every other BASELINE.json configuration measured by a "
                         "child process of this run: C1_text, C2, C2_file, C4, C5, C5_mindist -- ms per step, the dominant kernel's "
                         "HIP-event duration, algorithmic bytes, fraction of the HBM peak, parity against the oracle on sampled rows)")
    ap.add_argument("--config-leg", action="store_true",
                    help="this process IS one leg of another run's `configs` block: no variants, no proxy, no NumPy baseline, no "
                         "measured-peaks leg, bounded CPU legs")
    ap.add_argument("--all-modes", action="store_true", default=os.environ.get("OBTG_BENCH_ALL_MODES") == "1",
                    help="batch mode with --gpus > 1: after the weak line's own timed region, run `--mode rows --gather-minima` (ONE "
                         "SLSQP iteration's rows over the ranks, sparse minima all-gather: `scaling` strong) for C3 and for C4 in the "
                         "same processes and carry both in the line's `modes` -- one command, both curves (also OBTG_BENCH_ALL_MODES=1)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` starts the N ranks itself.  The parent never touches the GPU
# (no torch import, no HIP call) and never re-execs: it only starts fresh children and waits.
# ---------------------------------------------------------------------------------------------
def spawn_ranks(args):
    import signal
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []

    def stop_all(grace=5.0):
        """terminate every live rank, then kill what ignores it: a rank left in a barrier would hold its GPU"""
        live = [p for p in procs if p.poll() is None]
        for p in live:
            try:
                os.killpg(p.pid, signal.SIGTERM)
            except (ProcessLookupError, PermissionError):
                pass
        t_end = time.time() + grace
        while time.time() < t_end and any(p.poll() is None for p in live):
            time.sleep(0.1)
        for p in live:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except (ProcessLookupError, PermissionError):
                    pass
        for p in live:
            try:
                p.wait(timeout=5)
            except subprocess.TimeoutExpired:
                pass

    def on_signal(signum, _frame):
        stop_all()
        sys.exit(128 + signum)

    old = {sig: signal.signal(sig, on_signal) for sig in (signal.SIGTERM, signal.SIGINT)}
    rc = 0
    try:
        for r in range(args.gpus):
            env = dict(os.environ)
            env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            # own process group per rank: a signal to the launcher reaches the ranks through stop_all() only
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          start_new_session=True))
        live = list(procs)
        while live and rc == 0:
            time.sleep(0.2)
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0:
                    rc = code               # the FIRST failing rank's code is what the launcher returns
                    stop_all()              # one rank failed: the others would wait in a barrier forever
                    break
    finally:
        stop_all()
        for sig, h in old.items():
            signal.signal(sig, h)
    if rc:
        sys.stderr.write("bench.py: a rank exited with code %d\n" % rc)
    return rc


def algorithmic_bytes(N, d, n, R, P_t, P_s, sumK):
    """bytes one evaluation must move (inputs read once, outputs written once)."""
    L = 2 * n + R + 1
    by = {
        "temporal_sep": 8 * N * d * (n + 1) + 8 * P_t * L,
        "speed": 8 * N * d * (n + 1) + 8 * N * L,
        "ang_rate": 8 * N * d * (n + 1) + 8 * N * (4 * (n + R) + 1),
        "gjk": 24 * sumK + 64 * P_s,
        "fd_batch": 8 * N * d * (n + 1),      # one row written per evaluation (the one source row stays in cache)
    }
    # the survey's per-eval total counts the control points once
    total = 8 * N * d * (n + 1) + 8 * (P_t * L + N * L + N * (4 * (n + R) + 1)) + 24 * sumK + 64 * P_s
    return by, total


COUNTERS_PATH = os.path.join(REPO, "profiles", "counters.json")
_KERNEL_UNIT = {"pair_sweep": "gjk_kernels", "gjk": "gjk_kernels", "min_dist": "gjk_kernels", "temporal_sep": "bern_kernels",
                "speed": "bern_kernels", "ang_rate": "bern_kernels", "fd_batch": "bern_kernels"}


def _source_hash(_capi, unit):
    """obtg_source_hash of the LOADED library for a compile unit (what the running kernels were built from)."""
    try:
        return _capi.source_hash(unit)
    except Exception:
        return None


def counters_for(workload, kernel, running_hash, path=None):
    """The committed hardware-counter figures of (workload, kernel) -- HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE
    passes, VALU wave-instructions and busy fraction from the SQ passes (profiles/counters.json, written by
    tools/r06_counters.py from rocprofv3 --pmc output) -- IF they were taken on the kernels that are running now: every
    entry records the obtg_source_hash of its compile unit, the tree it was taken on and the file the raw pass is kept in.
    An entry whose hash is not `running_hash` is NOT reported: {"stale": True, ...} without figures (the line then says
    that counters exist but belong to other kernels); no entry at all: None."""
    path = path or COUNTERS_PATH
    try:
        entries = json.load(open(path)).get("entries", [])
    except (OSError, ValueError):
        return None
    hits = [e for e in entries if e.get("workload") == workload and e.get("kernel") == kernel]
    if not hits:
        return None
    for e in hits:
        if running_hash is not None and e.get("source_hash") == running_hash:
            out = dict(e)
            out["matches_running_library"] = True
            return out
    e = hits[-1]
    return {"stale": True, "matches_running_library": False, "taken_on_source_hash": e.get("source_hash"), "running_source_hash": running_hash,
            "tree": e.get("tree"), "source": e.get("source"),
            "note": "counter figures exist but were taken on other kernel sources: not reported"}


def measured_peaks(torch, dev):
    """This box's achievable HBM rates with stock kernels (tools/bw_probe.py): device copy (read + write
    bytes) and fill (write only), 1 GiB each -- well past the 256 MB Infinity Cache."""
    n = (1 << 30) // 8
    x = torch.empty(n, dtype=torch.float64, device=dev)
    y = torch.empty_like(x)

    def t(f, k=10):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(k):
            f()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e-3 / k
    tc, tw = t(lambda: y.copy_(x)), t(lambda: x.fill_(1.0))
    del x, y
    return 2 * n * 8 / tc / 1e9, n * 8 / tw / 1e9


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("OBTG_BENCH_REHEARSE_EXIT"):      # launcher rehearsal (tests/): "rank:code" -- that rank exits with
        r_, c_ = os.environ["OBTG_BENCH_REHEARSE_EXIT"].split(":")    # the code at once, the others wait as in a barrier
        if rank == int(r_):
            sys.exit(int(c_))
        time.sleep(120)
        sys.exit(0)
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to print a %d-GPU line from %d rank(s)"
                         % (args.gpus, world, args.gpus, world))

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: process group of %d ranks for --gpus %d" % (dist.get_world_size(), args.gpus))
    env = dict(rank=rank, world=world, local_rank=local_rank, use_dist=use_dist)

    if args.mode == "mindist":
        mindist_mode(args, rank)
    elif args.mode == "c1text":
        c1_text_mode(args, rank)
    else:
        line = step_bench(args, env)
        if args.mode == "batch" and args.all_modes and world > 1:          # (every rank: only rank 0 holds a line)
            # the other curve, in the same processes: ONE iteration's rows over the ranks with the sparse minima gather, C3 and C4.
            # Every rank takes part (the step has a collective); only rank 0 holds the lines.
            import copy
            modes = {}
            for wl in ("C3", "C4"):
                a2 = copy.copy(args)
                a2.mode, a2.workload, a2.gather_minima, a2.no_cpu, a2.batch = "rows", wl, "sparse", True, 0
                a2.steps = args.steps if wl == "C3" else max(3, min(args.steps, 10))
                a2.warmup = args.warmup if wl == "C3" else max(1, min(args.warmup, 2))
                l2 = step_bench(a2, env)
                if l2 is not None and l2.get("value") is not None:
                    modes["rows_" + wl] = {k: l2.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling",
                                                                  "config", "roofline", "kernels")}
            if rank == 0 and line is not None:
                line["modes"] = modes
        if rank == 0 and line is not None and line.get("value") is not None:
            if (args.workload == "C3" and args.mode == "batch" and world == 1 and not args.no_configs and not args.config_leg
                    and not args.no_cpu and not args.batch):
                line["configs"] = config_legs(args)
            print(json.dumps(line))
            sys.stdout.flush()
        if use_dist:
            dist.destroy_process_group()
        par = (line or {}).get("parity_check")
        if par is not None and not par["ok"]:
            raise SystemExit("bench.py: the timed step's device buffers DISAGREE with the oracle: %s" % json.dumps(par))
        return
    if use_dist:
        dist.destroy_process_group()


def step_bench(args, env):
    """One timed run of the step (modes batch / rows / pairs) inside the process group main() set up; rank 0 gets the line
    (a dict), the other ranks None."""
    import torch
    import torch.distributed as dist
    from optimalbeziertrajectorygeneration_amd import _capi, synth
    rank, world, local_rank, use_dist = env["rank"], env["world"], env["local_rank"], env["use_dist"]

    cfg = dict(synth.CONFIGS[args.workload])
    N, d, n, R = cfg["N"], cfg["d"], cfg["n"], cfg["R"]
    n_x = N * d * (n - 1)
    B = args.batch or (1 if args.mode == "pairs" else n_x + 1)
    B_total, row_begin = B, 0
    if args.mode == "rows":          # one iteration's rows over the ranks: contiguous balanced blocks
        from optimalbeziertrajectorygeneration_amd import distributed as _dd
        row_begin, B = _dd.shard_rows(B_total, world, rank)
        if B_total < world:          # (every rank decides alike: nobody is left waiting in a barrier)
            raise SystemExit("bench.py --mode rows: %d rows over %d ranks would leave ranks without a row" % (B_total, world))
        if args.materialise:
            raise SystemExit("bench.py --mode rows evaluates row ranges of a view (obtg_fd_view_begin_rows): not with --materialise")
    seed = 1234 + (1000 * rank if args.mode == "batch" else 0)   # batch mode: every rank its own swarm instance
    Y = synth.swarm_control_points(N, d, n, seed=seed)
    statics, pa, pb = synth.config_hull_sweep(args.workload, seed=1234)
    M = len(statics)
    ppts, poff = synth.pack_polys(statics) if M else (None, [0])
    use_gjk = (not args.no_gjk) and d >= 2
    max_sep, vmax, wmax, tfv = 0.9, 5.0, 1.0, 10.0

    ctx = _capi.Context(N, d, n, R, device=local_rank)
    # The library stays on the context's OWN stream (created non-blocking; obtg_ctx_use_own_stream is the default after
    # create): every hand-over between torch's work and the library's below is a torch.cuda.synchronize().
    # (obtg_ctx_set_stream(torch.cuda.current_stream().cuda_stream) would order the launches with torch's stream instead.)
    stream_note = "context's own non-blocking stream; hand-over by torch.cuda.synchronize()"
    if args.ang_order != "fast":
        ctx.set_ang_rate_order({"elevate_first": 1, "reference": 1, "exact": 2}[args.ang_order])      # ("reference": the old name of elevate_first)
    if use_gjk:
        ctx.set_polygons(ppts, poff)
        ctx.set_hull_pairs(pa, pb)
        ctx.set_fd_dedup(args.fd_dedup)
    P_t, L = ctx.num_pairs, 2 * n + R + 1
    P_s = len(pa) if use_gjk else 0

    dev = torch.device("cuda", local_rank)
    f64 = torch.float64
    d0 = torch.from_numpy(Y).to(dev)
    dY = torch.empty((B, N * d, n + 1), dtype=f64, device=dev)
    ctx.fd_batch_dev(d0.data_ptr(), 1, synth.FD_STEP, B, dY.data_ptr())     # inputs resident in HBM

    def barrier():
        if use_dist:
            dist.barrier()

    if args.mode == "pairs":
        return pairs_mode(args, ctx, dist, world, rank, dev, dY, B, max_sep, barrier, pa, pb, use_gjk, use_dist)
    lean = args.config_leg          # a leg of another run's `configs` block: the headline figures only

    d_tf = torch.full((B,), tfv, dtype=f64, device=dev)
    o_sep = torch.empty((B, P_t * L), dtype=f64, device=dev)
    o_sp = torch.empty((B, N * L), dtype=f64, device=dev)
    o_an = torch.empty((B, N * (4 * (n + R) + 1)), dtype=f64, device=dev) if d == 2 else None
    if use_gjk:
        g_flag = torch.empty((B, P_s), dtype=torch.int32, device=dev)
        g_p1 = torch.empty((B, P_s, 3), dtype=f64, device=dev)
        g_p2 = torch.empty((B, P_s, 3), dtype=f64, device=dev)
        g_dist = torch.empty((B, P_s), dtype=f64, device=dev)
        g_stat = torch.empty((B, P_s), dtype=torch.int32, device=dev)

    one_launch = use_gjk and not args.separate   # the N x N pair sweeps (temporal separation + gjkNew) as one grid
    # B0, the FD batch of this SLSQP iteration: never written when every sweep of the step can form its rows while it
    # stages them -- an obtg_fd_view over x's control points, the sweeps called with dY = NULL; kernels without that
    # form make the library write the batch once per step (what --materialise forces for all of them)
    fly_sweep, fly_dyn = ctx.fd_forms_on_the_fly()
    use_view = not args.materialise and row_begin + B <= n_x + 1
    on_the_fly = use_view and one_launch and o_an is not None and fly_sweep and fly_dyn      # two launches, nothing written
    # the dynamics launch is latency bound (one or two wavefronts per SIMD), the pair sweep VALU bound: on two streams
    # the first hides under the second.  A context owns one stream, so the dynamics launch gets a context of its own.
    two_streams = args.streams == 2 and on_the_fly
    ctx_dyn = ctx
    if two_streams:
        stream2 = torch.cuda.Stream(device=dev)
        ctx_dyn = _capi.Context(N, d, n, R, device=local_rank)
        ctx_dyn.set_stream(stream2.cuda_stream)
    Yp = None if use_view else dY.data_ptr()

    everything = one_launch and not two_streams       # (3-D rows: no angular rate; the 3-D sweep's launch then takes the speed rows too)

    def sweeps(B=B):
        if everything:       # all four families through one call (two launches at C3: pair sweep, dynamics)
            ctx.constraint_sweep_dev(Yp, d_tf.data_ptr(), B, max_sep, o_sep.data_ptr(), vmax, True, wmax, o_sp.data_ptr(),
                                     o_an.data_ptr() if o_an is not None else None, g_flag.data_ptr(), g_p1.data_ptr(),
                                     g_p2.data_ptr(), g_dist.data_ptr(),
                                     None, g_stat.data_ptr(), 128, 256)
            return
        if one_launch:
            ctx.pair_sweep_dev(Yp, B, max_sep, o_sep.data_ptr(), g_flag.data_ptr(), g_p1.data_ptr(),
                               g_p2.data_ptr(), g_dist.data_ptr(), None, g_stat.data_ptr(), 128, 256)
        else:
            ctx.temporal_sep_dev(Yp, B, max_sep, o_sep.data_ptr())
        if o_an is not None:    # speed + angular rate share their derivative curves: one launch
            ctx_dyn.dynamics_dev(Yp, d_tf.data_ptr(), B, vmax, True, wmax, o_sp.data_ptr(), o_an.data_ptr())
        else:
            ctx.speed_dev(Yp, d_tf.data_ptr(), B, vmax, True, o_sp.data_ptr())
        if use_gjk and not one_launch:
            ctx.gjk_swarm_dev(Yp, B, g_flag.data_ptr(), g_p1.data_ptr(), g_p2.data_ptr(),
                              g_dist.data_ptr(), None, g_stat.data_ptr(), 128, 256)

    d_min = None
    sparse_gather, gather_bytes = None, None
    if args.mode == "rows" and args.gather_minima:
        counts = [c for _, c in _dd.partition(B_total, world)]
        d_min = torch.zeros((max(counts), P_t), dtype=f64, device=dev)
        if args.gather_minima == "dense":
            d_min_all = torch.empty((world, max(counts), P_t), dtype=f64, device=dev)
            gather_bytes = 8 * world * max(counts) * P_t
        else:
            sparse_gather = _dd.SparseMinimaGather(B_total, N, ctx.n_veh + ctx.n_obs, d, n - 1, world=world, rank=rank)
            gather_bytes = sparse_gather.bytes_per_step
            sg_send = sparse_gather.send_rows(dev)
            sg_row0 = sparse_gather.send_row0(dev)

    gathered = {}

    def step():
        if not use_view:
            ctx.fd_batch_dev(d0.data_ptr(), 1, synth.FD_STEP, B, dY.data_ptr())
            return sweeps()
        for cx in ([ctx, ctx_dyn] if ctx_dyn is not ctx else [ctx]):
            cx.fd_view_begin(d0.data_ptr(), 1, synth.FD_STEP, B, row_begin=row_begin)
        sweeps()
        if sparse_gather is not None:
            # what this rank sends, straight from x's control points: per owned row the minima of the N - 1 pairs its vehicle
            # touches (obtg_temporal_sep_fd_min_rows_dev: B (N - 1) pair evaluations, no [B][P] block is formed or read back),
            # and, on the rank that owns it, row 0's P minima
            first = max(row_begin, 1)
            if row_begin == 0:
                ctx.temporal_sep_min_dev(d0.data_ptr(), 1, max_sep, sg_row0.data_ptr())
            if row_begin + B > first:
                ctx.temporal_sep_fd_min_rows_dev(d0.data_ptr(), 1, synth.FD_STEP, first, row_begin + B - first, max_sep,
                                                 sg_send[first - row_begin:].data_ptr())
        elif d_min is not None:
            # dense: the per-pair minima of this rank's rows from the library's reduced kernel, inside the same view (8 B P
            # bytes written instead of the 8 B P L of the full block read back by torch.amin: 58 GB at C4)
            ctx.temporal_sep_min_dev(None, B, max_sep, d_min.data_ptr())
        for cx in ([ctx, ctx_dyn] if ctx_dyn is not ctx else [ctx]):
            cx.fd_view_end()
        if d_min is not None:         # (torch's collectives on its own stream: hand over with the library's sync)
            ctx.sync()
            if sparse_gather is not None:
                gathered["sparse"], gathered["row0"] = sparse_gather.exchange_compact(sg_send[:B], sg_row0 if row_begin == 0 else None,
                                                                                            force=args.force_dist)
            elif use_dist:
                dist.all_gather_into_tensor(d_min_all.view(-1), d_min.view(-1))

    # spin-up: the clocks of an idle MI355X need a few hundred ms of work to settle (at C3 a step reads 0.257 ms
    # straight after start and 0.205 ms once they have); untimed, before the W warm-up steps
    # (with a collective inside step() -- the rows mode's gathers -- every rank must run the SAME number of steps: the ranks
    # agree after each batch whether to go on.  A time-based loop per rank deadlocks as soon as two ranks read the clock on
    # either side of the limit: round 5 met it with three gloo ranks, two in the all-gather of a 21st batch and one in the
    # barrier behind the loop.)
    t_spin = time.perf_counter()
    while True:
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        more = time.perf_counter() - t_spin < 0.4
        if use_dist:
            flag = torch.tensor([1 if more else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            more = bool(flag.item())
        if not more:
            break
    ctxs = [ctx] + ([ctx_dyn] if ctx_dyn is not ctx else [])

    def prof(on, only=None, period=1):
        for cx in ctxs:
            cx.set_profiling(on, only=only) if only else cx.set_profiling(on)
            cx.set_profile_period(period)
            cx.reset_kernel_stats()

    def kstats():
        out = {}
        for cx in ctxs:
            cx.sync()
            for kname, (ms, cnt) in cx.kernel_stats().items():
                a = out.get(kname, (0.0, 0))
                out[kname] = (a[0] + ms, a[1] + cnt)
        return out

    # warm-up, with events around every launch: finds the dominant kernel of this workload
    prof(True)
    for _ in range(max(args.warmup, 1)):
        step()
    torch.cuda.synchronize()
    wstats = kstats()
    KNAMES = ("pair_sweep", "temporal_sep", "speed", "ang_rate", "gjk", "fd_batch")
    dom_name = max(KNAMES, key=lambda k: wstats.get(k, (0.0, 0))[0])
    # timed region: HIP events (launch stream) on the dominant kernel only: every launch of a short run (the driver's
    # --steps 20 then has 20 samples), every 4th of a long one (a timed launch does not overlap its neighbours: 1 % of the step)
    prof(True, only=dom_name, period=args.event_period or (1 if args.steps < 100 else 4))
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=f64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    stats_timed = kstats()
    # sample rows of the buffers the last timed step left behind, for the parity check (before anything rewrites them)
    status_nonok = int((g_stat != 0).sum().item()) if use_gjk else 0
    snap = None
    if rank == 0 and not args.no_cpu and args.gpus == 1:
        dev_out = dict(Y=dY, sep=o_sep, speed=o_sp, ang=o_an)
        if use_gjk:
            dev_out.update(flag=g_flag, dist=g_dist, p1=g_p1)
        snap = parity_snapshot(dev_out, B)
    # the other kernels: the same steps once more with events on every launch, outside the timed region
    prof(True)
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    stats = kstats()
    stats[dom_name] = stats_timed[dom_name]
    prof(False)

    # ---- variants, beside -- never as -- `value`: the same step (a) with the trip-count history off (pairs swept in
    # list order), (b) with x MOVING between steps the way SLSQP moves it: every step evaluates the FD batch around a
    # new x, one N(0, 1e-2) step of a random walk on the free control points away from the previous one, so the history
    # the sweep orders its pairs by comes from a different point each time (identical replays are its best case),
    # (c) the structured finite-difference step.  Every variant: its own 0.2 s of untimed steps, then three timed runs of
    # max(--steps, 200) steps each; the line carries the median and the spread (max - min) / median.
    variants = None
    proxy = None
    _sumK = N * (n + 1) + (int(poff[-1]) if M else 0)
    total_bytes_ = algorithmic_bytes(N, d, n, R, P_t, P_s, _sumK)[1]

    def timed(nsteps, fn, reps=3, spin=0.2):
        t_s = time.perf_counter()
        k = 0
        while time.perf_counter() - t_s < spin:
            for _ in range(10):
                fn(k)
                k += 1
            torch.cuda.synchronize()
        runs = []
        for _ in range(reps):
            torch.cuda.synchronize()
            a = time.perf_counter()
            for i in range(nsteps):
                fn(i)
            torch.cuda.synchronize()
            runs.append(1e3 * (time.perf_counter() - a) / nsteps)
        runs.sort()
        med = runs[len(runs) // 2]
        return med, (runs[-1] - runs[0]) / med

    def entry(ms_spread, rows, **kw):
        ms, spread = ms_spread
        e = {"ms_per_step": round(ms, 4), "spread": round(spread, 4), "evals_per_s": round(rows / (ms * 1e-3), 1)}
        e.update(kw)
        return e

    def step_at(ptr, Bs=B):
        for cx in ctxs:
            cx.fd_view_begin(ptr, 1, synth.FD_STEP, Bs, row_begin=row_begin)
        sweeps(Bs)
        for cx in ctxs:
            cx.fd_view_end()

    def structured(Bs=B):
        ctx.constraint_sweep_fd_structured_dev(d0.data_ptr(), 1, synth.FD_STEP, d_tf.data_ptr(), Bs, max_sep,
                                               o_sep.data_ptr(), vmax, True, wmax, o_sp.data_ptr(), o_an.data_ptr(),
                                               g_flag.data_ptr(), g_p1.data_ptr(), g_p2.data_ptr(), g_dist.data_ptr(),
                                               None, g_stat.data_ptr(), 128, 256)

    nv = max(args.steps, 200)
    nv = max(20, min(nv, int(2000.0 / max(1e3 * elapsed / args.steps, 1e-3))))      # (a 26 ms step: 76 steps per run, not 500)
    can_structured = everything and o_an is not None and row_begin == 0
    if use_view and use_gjk and args.mode == "batch" and not args.no_variants and not lean:
        variants = {}
        ctx.set_gjk_history(False)
        variants["history_off"] = entry(timed(nv, lambda i: step_at(d0.data_ptr())), B,
                                        what="obtg_ctx_set_gjk_history(0): pairs swept in list order")
        ctx.set_gjk_history(True)
        rng = np.random.default_rng(99)
        walk = [Y.copy()]
        for _ in range(31):
            nxt = walk[-1].copy()
            nxt[:, 1:-1] += rng.normal(0.0, 1e-2, size=(N * d, n - 1))
            walk.append(nxt)
        d_walk = [torch.from_numpy(w).to(dev) for w in walk]
        seq = list(range(32)) + list(range(30, 0, -1))          # there and back: consecutive steps are one move apart
        variants["moving_x"] = entry(timed(nv, lambda i: step_at(d_walk[seq[i % len(seq)]].data_ptr())), B,
                                     what="x advanced between steps by N(0, 1e-2) on every free control point (random walk, "
                                          "32 positions there and back); history from the previous position")
        ctx.set_gjk_history(False)
        variants["moving_x_history_off"] = entry(timed(nv, lambda i: step_at(d_walk[seq[i % len(seq)]].data_ptr())), B)
        ctx.set_gjk_history(True)
        # (c) the structured finite-difference step: ONE launch that evaluates row 0 in full and, per perturbed row, only
        # the pairs and the vehicle its advanced control point touches (obtg_constraint_sweep_fd_structured_dev); same
        # output buffers, bit for bit (tests/), a different evaluation strategy -- hence a variant, not `value`.  Its own
        # parity figure: the buffers IT left behind against the oracle, on the same sample of rows as `parity_check`.
        if can_structured:
            try:
                for t_ in (o_sep, o_sp, o_an, g_dist):
                    t_.fill_(float("nan"))
                g_flag.fill_(-7)
                torch.cuda.synchronize()
                structured()
                torch.cuda.synchronize()
                s_par = None
                if rank == 0 and not args.no_cpu and args.gpus == 1:
                    s_snap = parity_snapshot(dict(Y=dY, sep=o_sep, speed=o_sp, ang=o_an, flag=g_flag, dist=g_dist, p1=g_p1), B)
                    s_par = parity_check(s_snap, N, d, n, R, statics, pa, pb, use_gjk, max_sep, vmax, wmax, tfv)
                    s_par = {k: s_par[k] for k in ("rows", "max_rel", "max_rel_elementwise", "flags_equal", "gjk_dist_max_rel", "ok")}
                t_ms = timed(nv, lambda i: structured())
                prof(True, only="pair_sweep")
                for i in range(50):
                    structured()
                kms, kcnt = kstats().get("pair_sweep", (0.0, 0))
                prof(False)
                gbs = B * total_bytes_ / ((kms / max(kcnt, 1)) * 1e-3) / 1e9 if kcnt else None
                s_cnt = counters_for(args.workload + "_fd_structured", "pair_sweep", _source_hash(_capi, "gjk_kernels"))
                variants["fd_structured"] = entry(
                    t_ms, B, kernel_avg_ms=round(kms / max(kcnt, 1), 5), launches_per_step=1, counters=s_cnt,
                    bound="hbm (stores): the launch moves its bytes at the box's write-only rate, DESIGN.md 4.10",
                    alg_bytes_per_launch=B * total_bytes_, achieved_gbs=round(gbs, 2) if gbs else None,
                    frac=round(gbs / HBM_PEAK_GBS, 5) if gbs else None, parity=s_par,
                    what="row 0 evaluated in full and streamed into all rows; per row only the N-1 separation pairs, "
                         "the hull pairs and the vehicle its advanced control point touches; outputs identical to the "
                         "brute-force sweep")
            except RuntimeError as e:
                variants["fd_structured"] = {"unsupported": str(e)}

    # ---- strong_scaling_proxy (1 GPU): what ONE SLSQP iteration's rows cost when they are split G ways -- the step on
    # the first ceil(B / G) rows of the same view, G = 1, 2, 4, 8 (the rows every rank of `--mode rows` would own, up to
    # which vehicle they perturb).  efficiency = t(B) / (G t(B / G)): the speed-up G GPUs would show over one, divided by
    # G, if nothing but the kernels' small-batch behaviour stood in the way (no collective on this path).
    if use_view and use_gjk and args.mode == "batch" and world == 1 and not args.no_proxy and not args.no_variants and not lean:
        proxy = {"rows": [], "what": "the step on the first ceil(B / G) rows of the view; efficiency = t(B) / (G t(B/G))"}
        base = {}
        for G in (1, 2, 4, 8):
            Bg = -(-B // G)
            e = {"G": G, "rows": Bg}
            ms, sp = timed(nv, lambda i: step_at(d0.data_ptr(), Bg))
            e["brute_force_ms"] = round(ms, 4)
            if G == 1:
                base["bf"] = ms
            e["brute_force_efficiency"] = round(base["bf"] / (G * ms), 4)
            if can_structured:
                try:
                    ms2, sp2 = timed(nv, lambda i: structured(Bg))
                    e["structured_ms"] = round(ms2, 4)
                    if G == 1:
                        base["st"] = ms2
                    e["structured_efficiency"] = round(base["st"] / (G * ms2), 4)
                except RuntimeError:
                    pass
            proxy["rows"].append(e)
        step_at(d0.data_ptr())          # leave the buffers as a full step left them
        torch.cuda.synchronize()

    evals = (B_total if args.mode == "rows" else world * B) * args.steps
    value = evals / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    sumK = N * (n + 1) + (int(poff[-1]) if M else 0)
    by, total_bytes = algorithmic_bytes(N, d, n, R, P_t, P_s, sumK)
    kernels = []
    if o_an is not None:   # the fused dynamics launch is booked under "ang_rate"
        by["ang_rate"] = by["ang_rate"] + 8 * N * L
    by["pair_sweep"] = by["temporal_sep"] + by["gjk"]    # when a shape falls back to two launches they report separately
    if stats.get("pair_sweep", (0.0, 0))[1] > 0 and stats.get("speed", (0.0, 0))[1] == 0 and o_an is None:
        by["pair_sweep"] += by["speed"]             # 3-D rows: the sweep's launch wrote the speed rows as well
    one_launch_step = (stats.get("pair_sweep", (0.0, 0))[1] > 0 and stats.get("ang_rate", (0.0, 0))[1] == 0 and
                       stats.get("speed", (0.0, 0))[1] == 0 and o_an is not None)
    if one_launch_step:
        # planar rows: the speed / angular-rate groups ran as the grid's last workgroups -- the launch IS the evaluation,
        # its algorithmic bytes are SURVEY.md 8(d)'s per-eval figure (control points counted once) x B
        by["pair_sweep"] = total_bytes
    sep_with_dynamics = (not one_launch_step and o_an is not None and stats.get("temporal_sep", (0.0, 0))[1] > 0 and
                         stats.get("ang_rate", (0.0, 0))[1] == 0 and stats.get("speed", (0.0, 0))[1] == 0)
    if sep_with_dynamics:
        # DEG_ELEV > 0: the separation rows and the speed / angular-rate rows share a launch (k_sep_dynamics_elev, booked as
        # "temporal_sep"): both families' bytes, the control points counted once
        by["temporal_sep"] += by["ang_rate"] - 8 * N * d * (n + 1)
    for name in KNAMES:
        ms, cnt = stats.get(name, (0.0, 0))
        if cnt == 0:
            continue
        avg_ms = ms / cnt
        gbs = B * by[name] / (avg_ms * 1e-3) / 1e9
        kernels.append(dict(kernel=name, launches=cnt, avg_ms=round(avg_ms, 5), in_timed_region=(name == dom_name),
                            alg_bytes_per_launch=B * by[name], achieved_gbs=round(gbs, 2),
                            frac=round(gbs / HBM_PEAK_GBS, 5)))
    dom = max(kernels, key=lambda k: k["avg_ms"])
    # hardware counters of the dominant kernel, from the committed passes -- only when they were taken on THESE kernels
    cnt = counters_for(args.workload, dom["kernel"], _source_hash(_capi, _KERNEL_UNIT.get(dom["kernel"], "all")))
    traffic = traffic_source = issue = None
    bound = "hbm"
    if cnt is not None and not cnt.get("stale"):
        traffic = cnt.get("hbm_bytes_per_launch")
        traffic_source = {k: cnt.get(k) for k in ("source", "tree", "unit", "source_hash", "matches_running_library", "kernel_symbol")}
        if cnt.get("valu_wave_insts") is not None:
            issue = {"valu_wave_insts": cnt.get("valu_wave_insts"), "busy_frac": cnt.get("valu_busy_frac"),
                     "lds_bank_conflict_frac": cnt.get("lds_bank_conflict_frac"),
                     "busy_frac_what": "SQ_INSTS_VALU x 4 clocks / (1024 SIMDs x the launch's clocks): the share of the chip's FP64-rate issue slots the launch fills",
                     "source": cnt.get("source_issue", cnt.get("source")), "tree": cnt.get("tree"), "source_hash": cnt.get("source_hash")}
            # which roof binds: the counters decide.  A launch that fills more of its issue slots than of the HBM peak is
            # bound by issue (the bit-exact gjkNew state machine), not by bytes
            if cnt.get("valu_busy_frac") is not None and cnt["valu_busy_frac"] > max(dom["frac"], 0.5):
                bound = "valu_issue"
    elif cnt is not None:
        traffic_source = cnt
    note = None
    if dom["kernel"] == "pair_sweep":
        note = ("one launch: the gjkNew sweep's workgroups (VALU bound) also write their row's temporal-separation "
                "block (HBM-write bound); bytes = both families' algorithmic bytes; see DESIGN.md 4.7")
        if one_launch_step:
            note = ("the whole step is this one launch: the gjkNew sweep's workgroups also write their row's temporal-"
                    "separation block, and the speed / angular-rate groups run as the grid's last workgroups; bytes = "
                    "all four families' algorithmic bytes; see DESIGN.md 4.7")
    if dom["kernel"] == "gjk":
        note = ("gjkNew sweep: VALU-issue bound by nature (PMC at C3: ~85 % VALU busy, LDS 34 %), reported against "
                "HBM as the contract asks; see DESIGN.md 4.3")
    copy_gbs, fill_gbs = measured_peaks(torch, dev) if not lean else (float("nan"), float("nan"))
    roofline = dict(bound=bound, kernel=dom["kernel"], achieved=dom["achieved_gbs"], peak=HBM_PEAK_GBS,
                    unit="GB/s", frac=dom["frac"], traffic=traffic, traffic_source=traffic_source, issue=issue,
                    bound_note=("achieved / peak / frac are the HBM figures the contract asks for; `bound` names the roof the "
                                "counters say binds this launch (valu_issue: the launch fills a larger share of the chip's VALU "
                                "issue slots than of the HBM peak -- DESIGN.md 4.7)"),
                    peak_measured=round(copy_gbs, 1) if not lean else None,
                    frac_measured=round(dom["achieved_gbs"] / copy_gbs, 5) if not lean else None,
                    peak_measured_write_only=round(fill_gbs, 1) if not lean else None,
                    peak_measured_note="this box, torch copy_ (read+write bytes) / fill_ of 1 GiB, HIP events",
                    note=note, step_achieved=round(B * total_bytes / (ms_per_step * 1e-3) / 1e9, 2))

    cpu = cpu_np = parity = None
    if rank == 0 and not args.no_cpu and args.gpus == 1:
        parity = parity_check(snap, N, d, n, R, statics, pa, pb, use_gjk, max_sep, vmax, wmax, tfv)
        parity["status_nonok"] = status_nonok
        cpu = cpu_baseline(args, N, d, n, R, Y, statics, pa, pb, use_gjk, max_sep, vmax, wmax, tfv)
        if not lean:
            cpu_np = cpu_baseline_numpy(N, d, n, R, Y, statics, pa, pb, use_gjk, max_sep, vmax, wmax, tfv)

    # --mode rows: a checksum over EVERY rank's rows (each rank sums its own, the sums travel as objects): the same figures
    # from one rank and from G ranks say that the row ranges tile the iteration
    checksum = None
    if args.mode == "rows":
        mine = {"rows": B, "sep_min_sum": float(torch.amin(o_sep.view(B, P_t, L), dim=2).sum().item()) if B else 0.0,
                "speed_sum": float(o_sp.sum().item()) if B else 0.0,
                "gjk_flag_sum": int(g_flag.sum().item()) if (use_gjk and B) else 0}
        parts = [mine]
        if use_dist:
            parts = [None] * dist.get_world_size()
            dist.all_gather_object(parts, mine)
        checksum = {k: sum(q[k] for q in parts) for k in mine}
        checksum["rows_per_rank"] = [q["rows"] for q in parts]
    gather_check = None
    if sparse_gather is not None and B and "sparse" in gathered:
        # what the exchange returned rebuilds the dense minima: this rank's first / last rows against its own reduction
        mine_rows = [row_begin, row_begin + B - 1]
        rebuilt = sparse_gather.dense_rows(gathered["sparse"], gathered["row0"], mine_rows)
        own = torch.amin(o_sep.view(B, P_t, L), dim=2)[[0, B - 1]]
        gather_check = bool(torch.equal(rebuilt, own)) and tuple(gathered["sparse"].shape) == (B_total, ctx.n_veh + ctx.n_obs - 1)
    # --mode rows, the same ranges through the STRUCTURED step (obtg_constraint_sweep_fd_structured_rows_dev: every rank
    # evaluates the unperturbed row, the source of its streams, and per row of its range only what that row's vehicle
    # touches): same barrier + MAX-over-ranks timing, the checksums over every rank's rows must be the brute-force ones
    rows_structured = None
    if args.mode == "rows" and everything and o_an is not None and use_view and use_gjk and not args.no_variants and B > 0:
        try:
            def structured_rows():
                ctx.constraint_sweep_fd_structured_dev(d0.data_ptr(), 1, synth.FD_STEP, d_tf.data_ptr(), B, max_sep,
                                                       o_sep.data_ptr(), vmax, True, wmax, o_sp.data_ptr(), o_an.data_ptr(),
                                                       g_flag.data_ptr(), g_p1.data_ptr(), g_p2.data_ptr(), g_dist.data_ptr(),
                                                       None, g_stat.data_ptr(), 128, 256, row_begin=row_begin)
            for t_ in (o_sep, o_sp):
                t_.fill_(float("nan"))
            g_flag.fill_(-7)
            torch.cuda.synchronize()      # torch filled on ITS stream; the library launches on the context's own
            for _ in range(max(args.warmup, 3)):
                structured_rows()
            barrier()
            torch.cuda.synchronize()
            ts0 = time.perf_counter()
            for _ in range(args.steps):
                structured_rows()
            torch.cuda.synchronize()
            barrier()
            el_s = time.perf_counter() - ts0
            if use_dist:
                tt = torch.tensor([el_s], dtype=f64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                el_s = float(tt.item())
            mine_s = {"sep_min_sum": float(torch.amin(o_sep.view(B, P_t, L), dim=2).sum().item()), "speed_sum": float(o_sp.sum().item()),
                      "gjk_flag_sum": int(g_flag.sum().item())}
            parts_s = [mine_s]
            if use_dist:
                parts_s = [None] * dist.get_world_size()
                dist.all_gather_object(parts_s, mine_s)
            sums = {k: sum(q[k] for q in parts_s) for k in mine_s}
            ms_s = 1e3 * el_s / args.steps
            rows_structured = {"ms_per_step": round(ms_s, 4), "evals_per_s": round(B_total / (ms_s * 1e-3), 1), "checksum": sums,
                               "checksum_equals_brute_force": all(sums[k] == checksum[k] for k in sums),
                               "what": "every rank's row range through the structured step (row 0 of the view evaluated by every "
                                       "rank, per row only what its vehicle touches); never `value`"}
        except RuntimeError as e:
            rows_structured = {"unsupported": str(e)}
    # proof of ranks: what the process group itself reports, and the device every rank ran on
    ranks_seen = dist.get_world_size() if use_dist else 1
    devices = [torch.cuda.get_device_name(local_rank) + " #%d" % local_rank]
    if use_dist:
        gathered = [None] * ranks_seen
        dist.all_gather_object(gathered, "%s #%d pid %d" % (torch.cuda.get_device_name(local_rank), local_rank, os.getpid()))
        devices = gathered
    status_note = None
    if status_nonok:
        status_note = ("%d of the last step's %d gjkNew evaluations ended with a status other than OK (cycle detected / "
                       "minimumDistance cap): inputs on which the reference's own gjkNew never returns; they are "
                       "evaluated, flagged per pair and counted in evals/s" % (status_nonok, B * P_s))
    if rank == 0:
        obst = ("%d curve obstacles (shapeObstacles)" % M) if cfg.get("n_curve_obs") else ("%d polygon obstacles" % M)
        line = {
            "metric": "constraint-evals/s (full swarm pairwise min-dist + dynamics) per SLSQP iter",
            "value": round(value, 2), "unit": "constraint-evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong" if args.mode == "rows" else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %d vehicles, %d-D, degree %d, DEG_ELEV %d, %s; "
                                   "FD batch B=%d rows per GPU per step (%s); families: "
                                   "temporal_sep(%d pairs)+max_speed+%sgjkNew(%d hull pairs)" % (
                                       args.workload, N, d, n, R, obst, B,
                                       "formed from x's control points inside the sweeps" if on_the_fly
                                       else ("a view over x's control points: formed inside the sweeps that can, written "
                                             "once per step for the others" if use_view
                                             else "written to HBM by obtg_fd_batch_dev each step"), P_t,
                                       "max_ang_rate+" if d == 2 else "", P_s),
                       "ang_rate_order": args.ang_order, "ang_rate_order_in_effect": (("fast", "elevate_first", "exact")[ctx.ang_rate_order_in_effect()] if d == 2 else None), "mode": args.mode, "rows_per_step_all_ranks": B_total if args.mode == "rows" else world * B,
                       "row_range_of_rank0": [row_begin, B] if args.mode == "rows" else None, "rows_structured": rows_structured,
                       "gather_minima": args.gather_minima if d_min is not None else None, "allgather_bytes": gather_bytes,
                       "gather_check": gather_check, "checksum": checksum,
                       "launches_per_step": len(kernels), "streams": 2 if two_streams else 1, "stream": stream_note,
                       "ranks_seen": ranks_seen, "backend": (args.backend if use_dist else None), "devices": devices,
                       "gjk_status_note": status_note,
                       "evals_per_step_per_gpu": B, "alg_bytes_per_eval": total_bytes,
                       "gjk_fd_dedup": bool(args.fd_dedup), "gjk_status_nonok_last_step": status_nonok},
            "roofline": roofline,
            "kernels": kernels,
            "variants": variants,
            "strong_scaling_proxy": proxy,
            "parity_check": parity,
            "cpu_baseline": cpu,
            "cpu_baseline_numpy": cpu_np,
        }
        for cx in ctxs:
            cx.close()
        return line
    for cx in ctxs:
        cx.close()
    return None


def parity_snapshot(dev_out, B):
    """Host copies of a sample of rows of the step's device buffers: first, last, and rows of every residue
    mod 8 (consecutive rows of a launch go to the 8 XCDs round-robin)."""
    rows = sorted(set([0, B - 1] + [min(B - 1, (k * (B - 1)) // 9 // 8 * 8 + k % 8) for k in range(1, 9)]))
    snap = {k: (v[rows].cpu().numpy() if v is not None else None) for k, v in dev_out.items()}
    snap["rows"] = rows
    return snap


def parity_check(dev_out, N, d, n, R, statics, pa, pb, use_gjk, max_sep, vmax, wmax, tfv):
    """The sampled rows of the buffers the LAST TIMED step left on the device against the CPU oracle.  The oracle
    sees the device's own FD batch rows.  Bars as in tests/: 1e-9 scale-aware on constraint values, flags
    identical, closest-point distances 1e-12."""
    from oracle import oracle as O
    from optimalbeziertrajectorygeneration_amd import synth
    O.build()
    rows = dev_out["rows"]
    Yr = dev_out["Y"]
    o_sep, o_sp, o_an = O.eval_batch(Yr, tfv, N, d, R, max_sep, vmax, wmax, nthreads=1)

    def rel(got, ref):
        fin = np.isfinite(ref)
        if (np.isfinite(got) != fin).any():
            return float("inf")
        if not fin.any():
            return 0.0
        return float((np.abs(got[fin] - ref[fin]) / np.maximum(np.abs(ref[fin]), np.abs(ref[fin]).max())).max())

    def rel_elem(got, ref):
        """largest |got - ref| / |ref| over the elements that are not tiny against their vector (|ref| > 1e-6 x scale)"""
        fin = np.isfinite(ref)
        if not fin.any():
            return 0.0
        scale = np.abs(ref[fin]).max()
        m = fin & (np.abs(ref) > 1e-6 * scale)
        return float((np.abs(got[m] - ref[m]) / np.abs(ref[m])).max()) if m.any() else 0.0

    worst = {"temporal_sep": rel(dev_out["sep"], o_sep), "speed": rel(dev_out["speed"], o_sp)}
    elem = {"temporal_sep": rel_elem(dev_out["sep"], o_sep), "speed": rel_elem(dev_out["speed"], o_sp)}
    if dev_out.get("ang") is not None:
        worst["ang_rate"] = rel(dev_out["ang"], o_an)
        elem["ang_rate"] = rel_elem(dev_out["ang"], o_an)
    flags_equal, dist_rel = True, 0.0
    if use_gjk:
        fl, di, p1 = dev_out["flag"], dev_out["dist"], dev_out["p1"]
        for k, r in enumerate(rows):
            hp, ho = synth.pack_polys(synth.hulls_from_Y(Yr[k], d) + statics)
            o = O.gjk_pairs(hp, ho, pa, pb, md_cap=256)
            flags_equal = flags_equal and bool((fl[k] == o["flag"]).all())
            sep = o["flag"] == 1
            if sep.any():
                dist_rel = max(dist_rel, float((np.abs(di[k][sep] - o["dist"][sep]) / np.maximum(1.0, np.abs(o["dist"][sep]))).max()),
                               float((np.abs(p1[k][sep] - o["c1"][sep]) / np.maximum(1.0, np.abs(o["c1"][sep]))).max()))
    max_rel = max(worst.values())
    ok = bool(max_rel <= 1e-9 and flags_equal and dist_rel <= 1e-12)
    return {"rows": len(rows), "row_ids": rows, "max_rel": max_rel, "per_family": worst,
            "max_rel_elementwise": max(elem.values()), "per_family_elementwise": elem,
            "elementwise_note": "|got - ref| / |ref| over elements with |ref| > 1e-6 x the vector's largest magnitude",
            "flags_equal": flags_equal,
            "gjk_dist_max_rel": dist_rel, "ok": ok,
            "against": "oracle/obtg_oracle.c on the device's own FD rows; buffers of the LAST timed step"}


def pairs_mode(args, ctx, dist, world, rank, dev, dY, B, max_sep, barrier, pa, pb, use_gjk, use_dist=False):
    """Pair-partitioned evaluation of ONE batch (same swarm on every rank): each rank sweeps a contiguous block of
    the lexicographic temporal-separation pair list and of the gjkNew hull pair list; ONE all-gather returns the
    per-pair separation minima and gjkNew's (dist, flag) to every rank."""
    import torch
    from optimalbeziertrajectorygeneration_amd.distributed import (GpuHullPairSweep, PairPartitionedSweep,
                                                                   all_gather_pair_blocks, gpu_temporal_sep_evaluator)
    sweep = PairPartitionedSweep(ctx.num_pairs, 1)
    evaluate = gpu_temporal_sep_evaluator(ctx, dY, B, max_sep, min_only=True)
    hull = GpuHullPairSweep(ctx, pa, pb) if use_gjk else None

    gather_gjk = hull is not None and args.gather_gjk

    def step():
        parts = [(evaluate(*sweep.my_block), sweep.blocks, 1)]
        if hull is not None:
            hp = hull.parts(dY, B)              # the rank's block of the hull sweep runs either way; its results travel on request
            if gather_gjk:
                parts += hp
        return all_gather_pair_blocks(parts, force=args.force_dist)

    # bytes one step's all-gather delivers to EVERY rank (blocks padded to the largest), per evaluation row and in all
    per_eval = 8 * world * max(c for _, c in sweep.blocks)
    if gather_gjk:
        per_eval += 12 * world * max(c for _, c in hull.blocks)
    allgather_bytes = {"per_evaluation": per_eval, "per_step": per_eval * B, "sent_per_rank_per_step": per_eval * B // world,
                       "what": "separation minima (8 B per pair)" + (" + gjkNew dist, flag (12 B per hull pair)" if gather_gjk else "")}

    for _ in range(max(args.warmup, 1)):
        out = step()
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # proof of ranks: what the process group itself reports, and the device every rank ran on
    ranks_seen = dist.get_world_size() if use_dist else 1
    me = "%s #%d pid %d" % (torch.cuda.get_device_name(dev), dev.index if dev.index is not None else 0, os.getpid())
    devices = [me]
    if use_dist:
        devices = [None] * ranks_seen
        dist.all_gather_object(devices, me)
    # every rank must hold the full, identical result
    chk = {"sep_min_sum": out[0].sum().item()}
    if gather_gjk:
        chk["gjk_dist_nansum"] = torch.nansum(out[1]).item()
        chk["gjk_flag_sum"] = int(out[2].sum().item())
    elif hull is not None:                    # not gathered: the sums over every rank's own block must still be the one-rank figures
        d_dist, d_flag = hull._buf[0], hull._buf[1]
        mine = [float(torch.nansum(d_dist).item()), int(d_flag.sum().item())]
        allm = [mine]
        if use_dist:
            allm = [None] * dist.get_world_size()
            dist.all_gather_object(allm, mine)
        chk["gjk_dist_nansum"] = sum(m[0] for m in allm)
        chk["gjk_flag_sum"] = sum(m[1] for m in allm)
    if rank == 0:
        return ({
            "metric": "pair-partitioned evals/s (one evaluation batch across all GPUs: pair lists split over the ranks, "
                      "separation minima in one RCCL all-gather)",
            "value": round(B * args.steps / elapsed, 2), "unit": "constraint-evals/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %d temporal-separation pairs + %d gjkNew hull pairs split over %d ranks, "
                                   "B=%d rows" % (args.workload, ctx.num_pairs, len(pa) if use_gjk else 0, world, B),
                       "ranks_seen": ranks_seen, "backend": (args.backend if use_dist else None), "devices": devices,
                       "allgather_bytes": allgather_bytes, "checksum": chk}})
    return None


def mindist_jacobian_plan(N=64, M=32, n=10, seed=1234):
    """The ONE `obtg_min_dist` call spatialSeparationJacobian makes for a C5-sized problem (64 vehicles + 32 curve obstacles,
    degree 10, n_x = 1152): the C(96, 2) = 4560 base pairs plus, per variable, the N + M - 1 = 95 pairs of the vehicle it
    advances -- 1152 x 95 + 4560 = 114 000 searches on 96 + 1152 curves (optimization._spatial_jac_plan, the provider's own
    plan; Examples/ComplexObstacles.py:49-63 hands the bare constraint to SLSQP: n_x + 1 all-pairs sweeps per iteration)."""
    from optimalbeziertrajectorygeneration_amd import synth
    from optimalbeziertrajectorygeneration_amd.optimization import _spatial_jac_plan
    Y = synth.swarm_control_points(N, 2, n, seed=seed)
    Yb = synth.fd_batch(Y)                                      # [n_x + 1][N * 2][n + 1]
    obs = synth.curve_obstacles(M, 2, n, seed=seed).reshape(M, 2, n + 1)
    obs3 = np.zeros((M, 3, n + 1))
    obs3[:, :2, :] = obs
    stack, pa, pb, P, col, row, pos = _spatial_jac_plan(Yb, N, 2, list(obs3))
    return dict(curves=stack, pa=pa, pb=pb, P=P, col=col, row=row, pos=pos, n_x=Yb.shape[0] - 1)


def mindist_parity(O, curves, pa, pb, r, sample, kw):
    """Sampled pairs of the device's call against the oracle's search of the same pair: (distance, t1, t2) IDENTICAL (bit for
    bit; NaN where NaN), node counts, gjkNew-call counts, depths and statuses equal."""
    o = O.min_dist_pairs(curves, pa[sample], pb[sample], nthreads=1, **kw)
    # (distance, t1, t2) of the searches that END; a search stopped by a budget (node / depth / inner-gjkNew cap: inputs on which
    # the reference itself does not return) has no result to compare -- its status and its counts up to the stop are compared
    ended = o["status"] == 0
    got, ref = r["res"][sample][ended], o["res"][ended]
    same_res = bool(np.array_equal(got, ref, equal_nan=True))
    eq = {k: bool(np.array_equal(np.asarray(r[k])[sample], np.asarray(o[k]))) for k in ("nodes", "gjk_calls", "depth", "status")}
    return {"pairs": int(len(sample)), "pairs_that_end": int(ended.sum()), "results_identical": same_res, "node_counts_equal": eq["nodes"],
            "gjk_call_counts_equal": eq["gjk_calls"], "depths_equal": eq["depth"], "statuses_equal": eq["status"],
            "ok": bool(same_res and all(eq.values())),
            "against": "oracle/obtg_oracle.c min_dist_rec (bezier.py:1283-1408) on the same curves, same budgets"}


def mindist_cpu_baseline(O, curves, pa, pb, kw, seconds, what):
    """The oracle's pair loop on a bounded, strided sample of the same pair list: 1 core, then OpenMP on the box's share."""
    n = len(pa)
    probe = np.arange(0, n, max(1, n // 64))[:64]
    t0 = time.perf_counter()
    O.min_dist_pairs(curves, pa[probe], pb[probe], nthreads=1, **kw)
    per = (time.perf_counter() - t0) / len(probe)
    take = int(max(64, min(n, seconds / max(per, 1e-9))))
    samp = np.arange(0, n, max(1, n // take))[:take]
    t0 = time.perf_counter()
    o = O.min_dist_pairs(curves, pa[samp], pb[samp], nthreads=1, **kw)
    dt = time.perf_counter() - t0
    out = {"value": round(len(samp) / dt, 2), "unit": "pair searches/s", "evals_per_s": round(len(samp) / dt / n, 5), "cores": 1, "kind": "port",
           "nodes_per_s": round(float(o["nodes"].sum()) / dt, 1),
           "sample": "%d of the %d pairs of %s (every %d-th), %.2f s, oracle/obtg_oracle.c -O2" % (len(samp), n, what, max(1, n // take), dt)}
    try:
        ncores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        ncores = os.cpu_count() or 1
    ncores = max(1, min(ncores, 16))
    if ncores > 1:
        t0 = time.perf_counter()
        O.min_dist_pairs(curves, pa[samp], pb[samp], nthreads=ncores, **kw)
        dtm = time.perf_counter() - t0
        out["all_cores"] = {"value": round(len(samp) / dtm, 2), "evals_per_s": round(len(samp) / dtm / n, 5), "cores": ncores,
                            "sample": "the same %d pairs, OpenMP (dynamic over pairs), %.2f s" % (len(samp), dtm)}
    return out


def mindist_mode(args, rank):
    """The `_minDist` variant of the pair sweep, reported separately: one evaluation of
    spatialSeparationConstraints (optimization.py:109-133) at the ComplexObstacles-style size of C5 -- all
    C(96, 2) pairs of 64 vehicles and 32 curve obstacles -- through the host-buffer entry points, in the
    reference's own algorithm (obtg_min_dist, node budget 2000 per pair: a fifth of the pairs would not
    finish in the reference either) and in the robust one (obtg_min_dist_robust); and, since round 6, at the size an SLSQP
    ITERATION asks for: the one call of spatialSeparationJacobian (`jacobian_list`: 114 000 searches)."""
    legs = ("reference_algorithm", "jacobian_list", "curve_polygon_reference_algorithm", "provider_end_to_end") if args.config_leg else None
    if args.mindist_legs:
        legs = tuple(x.strip() for x in args.mindist_legs.split(",") if x.strip())
    line = mindist_line(args, cpu=not args.no_cpu, legs=legs)
    if rank == 0:
        print(json.dumps(line))
        sys.stdout.flush()
    bad = [k for k, v in line["variants"].items() if isinstance(v, dict) and v.get("parity_check") and not v["parity_check"]["ok"]]
    if bad:
        raise SystemExit("bench.py --mode mindist: the device's searches DISAGREE with the oracle's: %s" % bad)


def mindist_line(args, cpu=True, legs=None, cpu_seconds=None):
    from optimalbeziertrajectorygeneration_amd import _capi, synth
    N, M, n = 64, 32, 10
    Yc = np.vstack((synth.swarm_control_points(N, 2, n, seed=1234), synth.curve_obstacles(M, 2, n, seed=1234)))
    curves = np.zeros((N + M, 3, n + 1))
    curves[:, :2, :] = Yc.reshape(N + M, 2, n + 1)
    pa, pb = synth.all_pairs(N + M)
    ctx = _capi.scratch_context()
    out = {}
    cpu_seconds = cpu_seconds if cpu_seconds is not None else args.cpu_seconds
    # VALU-busy fractions from the committed counter passes -- only those taken on THIS tree's kernels (counters_for)
    # curve <-> polygon (`_minDist2Poly`, bezier.py:1411-1496): the 64 vehicles against 64 polygon obstacles, 4096 pairs
    polys = synth.polygon_obstacles(64, seed=1234)
    ppts, poff = synth.pack_polys(polys)
    pc = np.repeat(np.arange(N), len(polys)).astype(np.int32)
    pp = np.tile(np.arange(len(polys)), N).astype(np.int32)
    kw_ref = dict(eps=1e-9, max_depth=128, max_nodes=2000)
    plan = mindist_jacobian_plan(N, M, n)
    all_legs = (("reference_algorithm", lambda: ctx.min_dist(curves, pa, pb, **kw_ref)),
                ("jacobian_list", lambda: ctx.min_dist(plan["curves"], plan["pa"], plan["pb"], **kw_ref)),
                ("robust", lambda: ctx.min_dist_robust(curves, pa, pb, eps=1e-9, max_nodes=400000)),
                ("jacobian_list_robust", lambda: ctx.min_dist_robust(plan["curves"], plan["pa"], plan["pb"], eps=1e-9, max_nodes=400000)),
                ("curve_polygon_reference_algorithm", lambda: ctx.min_dist2poly(curves[:N], ppts, poff, pc, pp, eps=1e-6, max_depth=128, max_nodes=2000)),
                ("curve_polygon_robust", lambda: ctx.min_dist2poly_robust(curves[:N], ppts, poff, pc, pp, eps=1e-9, max_nodes=200000)))
    O = None
    if cpu:
        from oracle import oracle as O
        O.build()
    for name, f in all_legs:
        if legs is not None and name not in legs:
            continue
        big = name.startswith("jacobian_list")
        ctx.set_profiling(True, only="min_dist")
        ctx.reset_kernel_stats()
        t0 = time.perf_counter()
        r = f()                                    # the first evaluation of this pair list: no node-count history yet
        first_ms = 1e3 * (time.perf_counter() - t0)
        for _ in range(1 if big else max(args.warmup // 10, 2)):
            r = f()
        ctx.reset_kernel_stats()
        reps = max(args.steps // 100, 3) if big else max(args.steps // 50, 5)
        t0 = time.perf_counter()
        for _ in range(reps):
            r = f()
        ms = 1e3 * (time.perf_counter() - t0) / reps
        kms, kcnt = ctx.kernel_stats().get("min_dist", (0.0, 0))
        ctx.set_profiling(False)
        nodes = int(r["nodes"].sum())
        npairs = len(r["status"])
        out[name] = dict(ms_per_eval=round(ms, 3), first_eval_ms=round(first_ms, 3), evals_per_s=round(1e3 / ms, 2), pairs=npairs,
                         kernel_avg_ms=round(kms / kcnt, 4) if kcnt else None,
                         pairs_per_s=round(npairs * 1e3 / ms, 1), nodes_per_eval=nodes, nodes_per_s=round(nodes * 1e3 / ms, 1),
                         status_counts=np.bincount(r["status"], minlength=4).tolist(),
                         result_checksum=float(np.nansum(r["res"][:, 0])))
        if "gjk_calls" in r:
            calls = int(r["gjk_calls"].sum())
            out[name]["gjk_calls_per_eval"] = calls
            out[name]["gjk_calls_per_s"] = round(calls * 1e3 / ms, 1)
        if big:
            out[name]["what"] = ("the ONE call of spatialSeparationJacobian at C5 size: %d base pairs + %d variables x %d pairs of the advanced "
                                 "vehicle = %d searches on %d curves" % (plan["P"], plan["n_x"], N + M - 1, npairs, len(plan["curves"])))
            out[name]["iteration_equivalent"] = ("one SLSQP iteration of Examples/ComplexObstacles.py:49-63 (the bare constraint, n_x + 1 = %d "
                                                 "all-pairs sweeps = %d searches) from %d" % (plan["n_x"] + 1, (plan["n_x"] + 1) * plan["P"], npairs))
        cnt = counters_for("C5_mindist", name, _source_hash(_capi, "gjk_kernels"))
        if cnt is not None:
            out[name]["counters"] = cnt
        if O is not None and name == "curve_polygon_reference_algorithm":
            # sampled (curve, polygon) pairs against the oracle's search (bezier.py:1411-1496): (alpha, t1, closest point) identical where the
            # search ends, node / call counts, depths and statuses equal; and the oracle's pair loop timed on the same sample
            rng = np.random.default_rng(6)
            samp = np.sort(rng.choice(npairs, size=min(npairs, 400), replace=False))
            kw2 = dict(eps=1e-6, max_depth=128, max_nodes=2000, md_cap=4096)
            t0 = time.perf_counter()
            oo = [O.min_dist2poly(curves[pc[k]], polys[pp[k]], **kw2) for k in samp]
            dt = time.perf_counter() - t0
            ended = np.array([o_["status"] == 0 for o_ in oo])
            same = all(np.array_equal(r["res"][k], o_["res"], equal_nan=True) for k, o_, e_ in zip(samp, oo, ended) if e_)
            cnt_ok = all(r["nodes"][k] == o_["nodes"] and r["gjk_calls"][k] == o_["gjk_calls"] and r["depth"][k] == o_["depth"] and r["status"][k] == o_["status"]
                         for k, o_ in zip(samp, oo))
            out[name]["parity_check"] = {"pairs": int(len(samp)), "pairs_that_end": int(ended.sum()), "results_identical": bool(same),
                                         "counts_and_statuses_equal": bool(cnt_ok), "ok": bool(same and cnt_ok),
                                         "against": "oracle/obtg_oracle.c min_dist_poly_rec (bezier.py:1411-1496), same budgets"}
            out[name]["cpu_baseline"] = {"value": round(len(samp) / dt, 1), "unit": "pair searches/s", "cores": 1, "kind": "port",
                                         "sample": "%d of the %d pairs, one ctypes call each (call overhead included), %.2f s" % (len(samp), npairs, dt)}
        if O is not None and name in ("reference_algorithm", "jacobian_list"):
            cs, pas, pbs = (plan["curves"], plan["pa"], plan["pb"]) if big else (curves, pa, pb)
            rng = np.random.default_rng(5)
            # sampled pairs: a stride through the list plus the longest searches that END (capped ones are budget-bound in the oracle too)
            ok_ids = np.nonzero(r["status"] == 0)[0]
            longest = ok_ids[np.argsort(r["nodes"][ok_ids])[-40:]]
            samp = np.unique(np.concatenate((rng.choice(npairs, size=min(npairs, 260), replace=False), longest)))
            out[name]["parity_check"] = mindist_parity(O, cs, pas, pbs, r, samp, kw_ref)
            out[name]["cpu_baseline"] = mindist_cpu_baseline(O, cs, pas, pbs, kw_ref, cpu_seconds, name)
    # The PROVIDER end to end: BezOptimization.spatialSeparationJacobian(x, robust=True, column=0) on the same problem -- the 1-D
    # distance constraint's (P, n_x) Jacobian a driver hands to SLSQP as `jac`: the host's plan of the one call, the call, the
    # assembly of the dense matrix.  With the robust search, whose every pair ends (the reference's own search raises on this
    # swarm's crossing pairs, as the reference does: optimization.py:127-131 never comes back from them).
    if legs is None or "provider_end_to_end" in legs:
        from optimalbeziertrajectorygeneration_amd import optimization as opt, bezier as bez
        Yv = synth.swarm_control_points(N, 2, n, seed=1234)
        Yo = synth.curve_obstacles(M, 2, n, seed=1234).reshape(M, 2, n + 1)
        bo = opt.BezOptimization(numVeh=N, dimension=2, degree=n, minimizeGoal='Euclidean', maxSep=0.9,
                                 initPoints=Yv.reshape(N, 2, n + 1)[:, :, 0], finalPoints=Yv.reshape(N, 2, n + 1)[:, :, -1],
                                 shapeObstacles=[bez.Bezier(np.ascontiguousarray(o)) for o in Yo])
        x = np.ascontiguousarray(Yv[:, 1:-1]).reshape(-1)
        assert np.array_equal(bo.reshapeVector(x), Yv)
        t0 = time.perf_counter()
        J = bo.spatialSeparationJacobian(x, robust=True, column=0)
        first_ms = 1e3 * (time.perf_counter() - t0)
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            J = bo.spatialSeparationJacobian(x, robust=True, column=0)
        ms = 1e3 * (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        F = bo.spatialSeparationConstraints(x, robust=True)
        ms_f = 1e3 * (time.perf_counter() - t0)
        out["provider_end_to_end"] = dict(
            ms_per_jacobian=round(ms, 2), first_ms=round(first_ms, 2), ms_per_constraint_evaluation=round(ms_f, 2),
            jacobian_shape=list(J.shape), nonzeros=int(np.count_nonzero(J)), finite=bool(np.isfinite(J).all() and np.isfinite(F).all()),
            what="BezOptimization.spatialSeparationJacobian(x, robust=True, column=0) at C5 size, host arrays in and out: plan of the ONE "
                 "call (114 000 searches), obtg_min_dist_robust, dense (4560 x 1152) matrix; and one spatialSeparationConstraints(x, robust=True); "
                 "what ONE SLSQP iteration of Examples/ComplexObstacles.py:49-63 costs through the provider instead of n_x + 1 sweeps")
    head = out.get("jacobian_list") or out.get("reference_algorithm") or next(iter(out.values()))
    return {"metric": "spatial-separation (_minDist) evals/s, host buffers in and out", "value": (out.get("reference_algorithm") or head).get("evals_per_s"),
            "unit": "constraint-evals/s", "n_gpus": 1, "higher_is_better": True, "dtype": "f64", "data": "synthetic",
            "vs_baseline": None, "config": {"workload": "C5-style: 64 vehicles + 32 curve obstacles, degree 10, 4560 curve pairs per evaluation; jacobian_list: the 114 000-search call of one SLSQP iteration",
                                            "note": "a branch-and-bound search per pair, not an HBM stream: the launch lasts as long as the dependent chain of its slowest pair "
                                                    "(profiles/r05_experiments/mindist_quad_phases.txt) -- nodes/s, gjkNew calls/s and the VALU-busy fraction of the launch stand in for a roofline"},
            "variants": out}


def c1_text_mode(args, rank):
    """BASELINE.json configs[0] as its text reads -- Examples/Example1: ONE vehicle, degree 10, 4 point obstacles, time-optimal,
    through the class path (optimization.py:86-98: the obstacles join the pair loop as constant curves: P = C(5, 2) = 10 pairs,
    210 / 21 / 41 values per evaluation) -- at the reference's own point x_r of tests/golden/c1_text.npz.  One step = what ONE
    SLSQP iteration asks of the three constraint closures: each family on x and its n_x = 15 forward-difference neighbours
    (16 rows, the trailing tf among the variables), through the drop-in BezOptimization's host-buffer path (three calls, three
    launches).  Launch-latency sized: the figure of interest is ms per iteration against the reference's, not bytes."""
    from optimalbeziertrajectorygeneration_amd import optimization as opt
    g = np.load(os.path.join(REPO, "tests", "golden", "c1_text.npz"))
    bo = opt.BezOptimization(numVeh=1, dimension=2, degree=10, minimizeGoal='TimeOpt', maxSep=1, maxSpeed=5, maxAngRate=1,
                             initPoints=[(0, 0)], finalPoints=[(10, 10)], initSpeeds=[1], finalSpeeds=[1],
                             initAngs=[0], finalAngs=[np.pi / 2], pointObstacles=g["obs"].tolist())
    x = g["x_r"]
    fams = ("tsep", "vmax", "ang")

    def step():
        return [bo._fd_values(x, f)[0] for f in fams]

    t_s = time.perf_counter()
    while time.perf_counter() - t_s < 0.3:
        step()
    for _ in range(max(args.warmup, 1)):
        step()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        F = step()
    elapsed = time.perf_counter() - t0
    B = x.size + 1
    ms = 1e3 * elapsed / args.steps
    # the kernels behind it: HIP events on every launch of a replay
    ctxs = [bo._ctx(True), bo._ctx(False)]
    for cx in ctxs:
        cx.set_profiling(True)
        cx.reset_kernel_stats()
    for _ in range(args.steps):
        step()
    kernels = []
    N, d, n, R, P, L = 1, 2, 10, 0, 10, 21
    by = {"temporal_sep": 8 * 5 * d * (n + 1) + 8 * P * L, "speed": 8 * N * d * (n + 1) + 8 * N * L,
          "ang_rate": 8 * N * d * (n + 1) + 8 * N * (4 * n + 1)}
    for cx in ctxs:
        cx.sync()
        for kname, (kms, cnt) in cx.kernel_stats().items():
            if cnt and kname in by:
                avg = kms / cnt
                gbs = B * by[kname] / (avg * 1e-3) / 1e9
                kernels.append(dict(kernel=kname, launches=cnt, avg_ms=round(avg, 5), alg_bytes_per_launch=B * by[kname],
                                    achieved_gbs=round(gbs, 3), frac=round(gbs / HBM_PEAK_GBS, 7)))
        cx.set_profiling(False)
    dom = max(kernels, key=lambda k: k["avg_ms"]) if kernels else None
    parity = cpu = None
    if not args.no_cpu:
        from oracle import oracle as O
        O.build()
        X, _ = bo._fd_rows(x)
        Yr = bo.reshapeVectors(X)
        tfr = X[:, -1]

        def rel(got, ref):
            return float((np.abs(got - ref) / np.maximum(np.abs(ref), np.abs(ref).max())).max())
        ref_fix = {"tsep": g["tsep_r"], "vmax": g["maxspeed_r"], "ang": g["angrate_r"]}
        vs_reference = {f: rel(F[i][0], ref_fix[f]) for i, f in enumerate(fams)}       # row 0 against the REFERENCE's own values
        orc = {"tsep": [], "vmax": [], "ang": []}
        for b in range(B):
            yo = np.vstack([Yr[b]] + [np.full((1, n + 1), v) for o in g["obs"] for v in o])
            orc["tsep"].append(O.temporal_sep(yo, 5, 2, 0, 1.0))
            orc["vmax"].append(O.speed(Yr[b], 1, 2, 0, tfr[b], 5.0, 1))
            orc["ang"].append(O.ang_rate(Yr[b], 1, 0, tfr[b], 1.0))
        vs_oracle = {f: max(rel(F[i][b], orc[f][b]) for b in range(B)) for i, f in enumerate(fams)}
        worst = max(max(vs_reference.values()), max(vs_oracle.values()))
        parity = {"rows": B, "max_rel": worst, "row0_vs_reference_fixture": vs_reference, "all_rows_vs_oracle": vs_oracle, "ok": bool(worst <= 1e-9),
                  "against": "tests/golden/c1_text.npz (written by the reference's own closures) on row 0; oracle/obtg_oracle.c on all %d rows" % B}
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < min(args.cpu_seconds, 2.0):
            for b in range(B):
                yo = np.vstack([Yr[b]] + [np.full((1, n + 1), v) for o in g["obs"] for v in o])
                O.temporal_sep(yo, 5, 2, 0, 1.0); O.speed(Yr[b], 1, 2, 0, tfr[b], 5.0, 1); O.ang_rate(Yr[b], 1, 0, tfr[b], 1.0)
            reps += 1
        dt = time.perf_counter() - t0
        cpu = {"value": round(B * reps / dt, 1), "unit": "constraint-evals/s", "cores": 1, "kind": "port",
               "sample": "the same %d rows x %d passes through the oracle's one-row calls (ctypes overhead included), %.2f s" % (B, reps, dt)}
    if rank == 0:
        print(json.dumps({
            "metric": "constraint-evals/s (full swarm pairwise min-dist + dynamics) per SLSQP iter", "value": round(B * args.steps / elapsed, 1),
            "unit": "constraint-evals/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "reference fixture point (tests/golden/c1_text.npz)",
            "config": {"workload": "C1_text: 1 vehicle, 2-D, degree 10, 4 point obstacles (class path: 10 pairs), time-optimal; one step = the three "
                                   "constraint closures on x and its 15 forward-difference neighbours (B=16 rows) through the host-buffer path",
                       "launches_per_step": 3, "evals_per_step_per_gpu": B,
                       "note": "launch-latency bound: a step is three host calls of one launch each; the reference takes 0.45 s for its whole "
                               "22-iteration solve (BASELINE.md 2)"},
            "roofline": (dict(bound="launch latency", kernel=dom["kernel"], achieved=dom["achieved_gbs"], peak=HBM_PEAK_GBS, unit="GB/s",
                              frac=dom["frac"], traffic=None) if dom else None),
            "kernels": kernels, "parity_check": parity, "cpu_baseline": cpu}))
        sys.stdout.flush()
    if parity is not None and not parity["ok"]:
        raise SystemExit("bench.py --mode c1text: the closures DISAGREE with the reference fixture / the oracle: %s" % json.dumps(parity))


def config_legs(args):
    """`configs`: every other BASELINE.json configuration, each measured by a child process of this run (fresh process, the
    same script with --config-leg: its own context, spin-up, timed region with HIP events on the dominant kernel, parity of
    the timed step's buffers against the oracle, bounded CPU legs) -- so that the ONE line the driver records carries them
    all.  The parent has finished its own timing; parent + one child use the GPU at a time."""
    legs = (("C1_text", ["--mode", "c1text", "--steps", "200", "--warmup", "20"]),
            ("C2", ["--workload", "C2", "--steps", "300", "--warmup", "30"]),
            ("C2_file", ["--workload", "C2_file", "--steps", "300", "--warmup", "30"]),
            ("C4", ["--workload", "C4", "--steps", "5", "--warmup", "2"]),
            ("C5", ["--workload", "C5", "--steps", "60", "--warmup", "10"]),
            ("C5_mindist", ["--mode", "mindist", "--steps", "100", "--warmup", "20"]))
    out = {}
    t_all = time.perf_counter()
    for name, extra in legs:
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--config-leg", "--cpu-seconds", "2.5"] + extra
        t0 = time.perf_counter()
        try:
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=150, text=True)
        except subprocess.TimeoutExpired:
            out[name] = {"error": "timed out after 150 s"}
            continue
        wall = time.perf_counter() - t0
        lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
        if not lines:
            out[name] = {"error": "exit code %d, no line" % p.returncode, "stderr_tail": p.stderr[-400:]}
            continue
        ln = json.loads(lines[-1])
        e = {"exit_code": p.returncode, "wall_s": round(wall, 1), "value": ln.get("value"), "unit": ln.get("unit"),
             "ms_per_step": ln.get("ms_per_step"), "workload": (ln.get("config") or {}).get("workload")}
        if name == "C5_mindist":
            v = ln["variants"]
            for leg in ("reference_algorithm", "jacobian_list", "curve_polygon_reference_algorithm"):
                if leg in v:
                    e[leg] = {k: v[leg].get(k) for k in ("ms_per_eval", "kernel_avg_ms", "pairs", "pairs_per_s", "nodes_per_s", "gjk_calls_per_s",
                                                         "status_counts", "counters", "parity_check", "cpu_baseline")}
            if "provider_end_to_end" in v:
                e["provider_end_to_end"] = v["provider_end_to_end"]
            e["ms_per_step"] = v["jacobian_list"]["ms_per_eval"] if "jacobian_list" in v else v["reference_algorithm"]["ms_per_eval"]
            e["parity"] = {"ok": all((v[leg].get("parity_check") or {"ok": True})["ok"] for leg in v if isinstance(v[leg], dict))}
        else:
            rf = ln.get("roofline") or {}
            dom = next((k for k in (ln.get("kernels") or []) if k["kernel"] == rf.get("kernel")), None)
            e["kernel"] = {"name": rf.get("kernel"), "avg_ms": dom["avg_ms"] if dom else None, "launches": dom["launches"] if dom else None,
                           "alg_bytes_per_launch": dom["alg_bytes_per_launch"] if dom else None, "frac": rf.get("frac"), "bound": rf.get("bound"),
                           "traffic": rf.get("traffic"), "timing": "HIP events on the launch stream, inside the leg's timed region"}
            e["launches_per_step"] = (ln.get("config") or {}).get("launches_per_step")
            pc = ln.get("parity_check") or {}
            e["parity"] = {k: pc.get(k) for k in ("rows", "max_rel", "max_rel_elementwise", "flags_equal", "gjk_dist_max_rel", "ok", "against")}
            cb = ln.get("cpu_baseline") or {}
            e["cpu_baseline"] = {"value": cb.get("value"), "cores": cb.get("cores"), "kind": cb.get("kind"), "sample": cb.get("sample"),
                                 "all_cores": cb.get("all_cores")}
            if (ln.get("config") or {}).get("gjk_status_note"):
                e["gjk_status_note"] = ln["config"]["gjk_status_note"]
        out[name] = e
    out["_wall_s"] = round(time.perf_counter() - t_all, 1)
    out["_what"] = ("each entry: a child process of this run (python bench.py --config-leg ...), its own timed region; kernel.avg_ms from "
                    "HIP events on the launch stream; parity against oracle/ on rows sampled from the timed step's buffers")
    return out


def cpu_baseline(args, N, d, n, R, Y, statics, pa, pb, use_gjk, max_sep, vmax, wmax, tfv):
    """The CPU oracle (C port of the reference path, single thread -- the reference is
    single-threaded) on a bounded sample of the same FD batch."""
    from oracle import oracle as O
    from optimalbeziertrajectorygeneration_amd import synth
    O.build()

    def run(rows, nthreads, passes=1):
        """time only the oracle calls; inputs (FD rows, packed hulls) are prepared before"""
        Yb = synth.fd_batch(Y, B=rows)
        hulls = [synth.pack_polys(synth.hulls_from_Y(Yb[b], d) + statics) for b in range(rows)] if use_gjk else []
        t0 = time.perf_counter()
        for _ in range(passes):
            O.eval_batch(Yb, tfv, N, d, R, max_sep, vmax, wmax, nthreads=nthreads)
            for hp, ho in hulls:
                O.gjk_pairs(hp, ho, pa, pb, md_cap=256, nthreads=nthreads)
        return time.perf_counter() - t0

    probe = run(8, 1)
    want_rows = args.cpu_seconds / (probe / 8)
    rows = int(max(16, min(4000, want_rows)))              # 4000 rows = 1.4 GB of oracle output at C3: the sample is
    passes = int(max(1, min(8, round(want_rows / rows))))   # walked several times rather than made larger
    dt = run(rows, 1, passes)
    out = {"value": round(rows * passes / dt, 3), "unit": "constraint-evals/s", "cores": 1, "kind": "port",
           "sample": "%d FD-batch rows of the same %s workload (all families) x %d passes, %.1f s, oracle/obtg_oracle.c -O2"
                     % (rows, args.workload, passes, dt)}
    # the same port with OpenMP over rows / pairs on every host core (the reference itself is serial)
    # the GPU box exposes every host CPU but grants a share of 16 per GPU: never oversubscribe OpenMP
    try:
        ncores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        ncores = os.cpu_count() or 1
    ncores = max(1, min(ncores, 16))
    if ncores > 1:
        dtm = run(rows, ncores)
        out["all_cores"] = {"value": round(rows / dtm, 3), "cores": ncores,
                            "sample": "%d rows, OpenMP, %.2f s" % (rows, dtm)}
    return out


def cpu_baseline_numpy(N, d, n, R, Y, statics, pa, pb, use_gjk, max_sep, vmax, wmax, tfv):
    """Reference-shaped NumPy port (oracle/numpy_port.py: the reference's per-pair Python loops, dense
    coefficient matrices and dict-simplex gjkNew, minus its Bezier objects), EVERY family of the evaluation, one
    thread -- the figure the >= 50x target is quoted against."""
    from oracle import numpy_port as P
    from optimalbeziertrajectorygeneration_amd import synth

    def one(Yr):
        P.temporal_sep(Yr, N, d, R, max_sep)
        P.speed(Yr, N, d, R, tfv, vmax, True)
        if d == 2:
            P.ang_rate(Yr, N, R, tfv, wmax)
        if use_gjk:
            polys = synth.hulls_from_Y(Yr, d) + statics
            for a, b in zip(pa, pb):
                P.gjk_new(polys[a], polys[b])

    Yb = synth.fd_batch(Y, B=16)
    P.temporal_sep(Yb[0], N, d, R, max_sep)      # builds the coefficient-matrix caches, like the drivers' warm-up
    P.speed(Yb[0], N, d, R, tfv, vmax, True)
    if d == 2:
        P.ang_rate(Yb[0], N, R, tfv, wmax)
    t0 = time.perf_counter()
    rows = 0
    while rows < 16 and (rows == 0 or time.perf_counter() - t0 < 6.0):
        one(Yb[rows])
        rows += 1
    dt = time.perf_counter() - t0
    return {"value": round(rows / dt, 3), "unit": "constraint-evals/s", "cores": 1, "kind": "port",
            "sample": "%d rows, all families (Bernstein sweeps%s), %.1f s, oracle/numpy_port.py; the real "
                      "reference measured in the survey container is ~5.6x slower than this port on the Bernstein "
                      "families (BASELINE.md section 2)" % (rows, " + one gjkNew per hull pair" if use_gjk else "", dt)}


if __name__ == "__main__":
    main()
