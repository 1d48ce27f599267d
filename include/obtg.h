/*
 * obtg.h -- C ABI of libobtg_hip.so: MI355X (gfx950) evaluation of the constraint /
 * cost hot path that SciPy SLSQP calls on every iteration of the reference
 * (caslabuiowa/OptimalBezierTrajectoryGeneration).
 *
 * The reference has no FFI layer (it is pure Python); the seam this ABI replaces is
 * the set of Python callables SLSQP invokes.  Each entry point names the reference
 * function(s) it stands in for (paths relative to the reference checkout).  The
 * Python look-alikes in optimalbeziertrajectorygeneration_amd/ bind these symbols
 * with ctypes; INTEGRATION.md shows the reference-side stub.
 *
 * Conventions
 *   - plain C symbols, POD arguments, no exceptions, no global state outside the ctx;
 *   - all floating point is IEEE binary64; all index / flag arrays are int32 (traces int16);
 *   - return value: 0 = OBTG_OK, negative = argument / device error (obtg_strerror).
 *     Per-item algorithmic outcomes (collision, iteration caps) go to flag/status
 *     arrays, never to the return code;
 *   - "host" entry points take caller-allocated HOST buffers, block until the result
 *     is in `out`, and leave the device idle.  "_dev" entry points take DEVICE
 *     pointers, enqueue on the context's stream and return immediately
 *     (obtg_sync() or the caller's own stream sync completes them);
 *   - a context is used by one thread at a time (SLSQP is serial); distinct contexts
 *     are independent.  There is NO CPU fallback: obtg_ctx_create fails when no
 *     gfx950 device is usable.
 *
 * Shapes: N = n_veh vehicles, d = dim, n = deg (n+1 control points), R = deg_elev,
 * M = n_point_obs.  One "evaluation row" of control points is
 *     Y[(N*d)][(n+1)]  row-major float64  (the `y` of reshapeVector, optimization.py:242-285)
 * and batched calls take B such rows back to back (B = n_x+1 for one SLSQP Jacobian).
 */
#ifndef OBTG_H
#define OBTG_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

typedef struct obtg_ctx obtg_ctx;

enum {
    OBTG_OK = 0,
    OBTG_ERR_ARG = -1,        /* bad argument (null pointer, negative size, unsupported dim ...) */
    OBTG_ERR_DEVICE = -2,     /* HIP runtime error (see obtg_last_error) */
    OBTG_ERR_NO_DEVICE = -3,  /* no usable gfx950 device */
    OBTG_ERR_OOM = -4,        /* device or host allocation failed */
    OBTG_ERR_UNSUPPORTED = -5 /* degree / size outside what the kernels support */
};

/* per-item GJK status (status[] arrays) */
enum {
    OBTG_ST_OK = 0,
    OBTG_ST_MD_CAP = 1,  /* minimumDistance's `while True` (gjk/gjk.py:277) exceeded md_cap rounds */
    OBTG_ST_MAXITER = 2, /* gjkNew exhausted maxIter (flag = -1, gjk/gjk.py:269-270) */
    OBTG_ST_CYCLE = 3    /* minimumDistance returned exactly to an earlier (simplex, direction) state: the
                          * reference's loop depends on nothing else, so it never exits on this input.
                          * Reported by the 3-D state machine (any z != 0) and by the curve-distance entry points
                          * (obtg_min_dist, obtg_min_dist2poly: inner calls, seen as OBTG_MD_GJK_CAP; since round 6 their
                          * planar calls carry the checkpoint too); the planar hull SWEEPS are guarded by md_cap alone. */
};

/* per-item minDist status */
enum {
    OBTG_MD_OK = 0,
    OBTG_MD_NODE_CAP = 1,  /* node budget exhausted (reference: runs for seconds / forever) */
    OBTG_MD_DEPTH_CAP = 2, /* deeper than max_depth (reference: RecursionError / cnt > 1000, bezier.py:1310) */
    OBTG_MD_GJK_CAP = 3    /* an inner gjkNew never returns: a proven cycle of its minimumDistance loop, or md_cap rounds */
};

const char* obtg_strerror(int code);

/* ABI revision.  Bumped whenever an EXISTING symbol changes meaning (new symbols alone do not bump it), so that a caller
 * bound against an older header can notice at load time.
 *   4  round 4: obtg_ctx_set_stream(ctx, NULL) selects the NULL (legacy default) stream -- the handle is taken as given.
 *      Until then NULL meant "the context's own non-blocking stream"; that meaning moved to obtg_ctx_use_own_stream.
 *      A caller written against revision <= 3 that passes NULL to get the private stream back still links and runs, but
 *      now serialises with the legacy stream: call obtg_ctx_use_own_stream instead.
 *   5  round 5: nothing changed meaning; new: obtg_abi_version, obtg_fast_kernels, obtg_ctx_ang_rate_order_in_effect, obtg_temporal_sep_active[_dev],
 *      obtg_comm_* and obtg_temporal_sep_min_gather_dev (the collective behind the C ABI); 9 control points (degree 8)
 *      joined the specialised counts.
 *   6  round 6: nothing changed meaning; new: obtg_source_hash, obtg_libm_pow_matches. */
#define OBTG_ABI_VERSION 6
int obtg_abi_version(void);

/* Which sources this library was built from: the first 16 hex digits of a sha256 over a compile unit's source, the headers of
 * csrc/ and its compiler flags.  unit: "gjk_kernels", "bern_kernels", "capi", "tables", "comm", "libm_check", or "all" (NULL =
 * "all"); NULL is returned for a name that is none of these.  Counter files under profiles/ record the hash of the kernels they
 * were taken on; bench.py drops a counter whose hash is not the running library's instead of reporting it as measured. */
const char* obtg_source_hash(const char* unit);

/* 1 when this host's libm rounds pow(x, 2.0) exactly as the device's restatement of it does (csrc/libm_pow2.h: glibc 2.35,
 * the FMA build) on a few thousand inputs, 0 when not.  `a**2` in the reference (gjk/gjk.py:460; optimization.py:343-459 on the
 * bounds) is libm's pow; with 0 the device's closest points / distances can differ from a reference run on THIS host by one
 * ulp (and `_minDist` searches that feed them back can take another path), although they still equal the committed fixtures.
 * Evaluated once per process, no device needed. */
int obtg_libm_pow_matches(void);

/* Which specialised (template-instantiated) kernel families exist for curves of `deg` in `dim` dimensions -- a bit mask:
 *   1  separation / speed rows, one-vs-many, structured separation Jacobian blocks (2-D and 3-D)
 *   2  angular rate at DEG_ELEV = 0, the hull sweeps on deg + 1 points per object, the one-launch steps
 *   4  DEG_ELEV > 0: separation + dynamics in one launch, the elevated structured step
 * 0: every family of that shape runs on the any-degree kernels (one wave per item).  The list itself lives in ONE place,
 * csrc/obtg_internal.h (OBTG_NC_*); callers that pick a code path by degree ask here instead of repeating it. */
int obtg_fast_kernels(int dim, int deg);
/* last HIP error string seen by this context (empty when none) */
const char* obtg_last_error(const obtg_ctx*);
/* number of usable gfx950 devices (0 when none / no HIP runtime) */
int obtg_device_count(void);
/* every exported symbol, NUL-separated list terminated by an empty string (for load tests) */
const char* obtg_abi_symbols(void);

/* Pinned (page-locked) host memory.  The host-buffer entry points accept any host pointer; buffers from
 * obtg_host_alloc are DMA targets as they are (PCIe rate), pageable buffers (e.g. a plain NumPy array) are staged
 * through a pinned ring inside the context, chunk by chunk, overlapped with the transfer. */
int obtg_host_alloc(size_t bytes, void** out);
int obtg_host_free(void* p);

/* ---- context -------------------------------------------------------------------------
 * Replaces: BezOptimization.__init__ (optimization.py:21-63) as far as the constraint
 * closures need it, plus the class-level matrix caches of bezier.py:48-52 and the
 * "warm the caches" idiom of the drivers (Examples/Example1_DubinsCarTimeOptimal.py:133-136):
 * coefficient tables for (deg, deg_elev) are built once here and stay device-resident.
 * point_obs[M][d] are the pointObstacles that temporalSeparationConstraints appends as
 * constant curves (optimization.py:86-98).  device = HIP ordinal (>= 0). */
int obtg_ctx_create(obtg_ctx** out, int n_veh, int dim, int deg, int deg_elev,
                    int n_point_obs, const double* point_obs, int device);
void obtg_ctx_destroy(obtg_ctx*);
/* Streams.  A context starts on a stream of its own, created non-blocking: NOT ordered with the null stream.
 * obtg_ctx_set_stream: every later call goes to the caller's HIP stream, the handle taken as given -- NULL is the null
 * stream itself, which is torch's default stream: a torch user who passes torch.cuda.current_stream().cuda_stream gets
 * launches ordered with torch's own work whichever stream that is.  obtg_ctx_use_own_stream: back to the private stream.
 * (Until round 4 NULL meant "own stream", and torch's default stream, handle 0, silently un-ordered a caller's work.) */
int obtg_ctx_set_stream(obtg_ctx*, void* hip_stream);
int obtg_ctx_use_own_stream(obtg_ctx*);
/* DEG_ELEV is a module constant read at call time (optimization.py:17): allow changing it */
int obtg_ctx_set_deg_elev(obtg_ctx*, int deg_elev);
/* Angular rate with DEG_ELEV > 0 (optimization.py:578-611).  The reference elevates the position by R first and
 * forms every product at degree n+R (a degree-4(n+R) square: 441 coefficients from degree-220 operands at R = 100).
 * Elevation commutes with diff/mul/add, so by default (elevate_first = 0) the library forms numerator and
 * denominator at degree 4n from the original control points and elevates both by 4R before the element-wise
 * quotient -- the same control points up to rounding (within the 1e-9 bar on every reference fixture), an order of
 * magnitude less arithmetic.  elevate_first = 1 keeps the reference's order of operations (generic kernel).
 * 2 = "exact": the default order, then the rows of vehicles that nearly stop -- a control point of |v|^2 three orders
 * below the curve's largest, where every float64 evaluation of the quotient, the reference's included, loses up to
 * 1e-8 to cancellation -- once more in double-double arithmetic, rounded once: those rows are then within a few 1e-16
 * of the exact rational value.  One more (small) launch behind the dynamics launch; not in the structured step. */
int obtg_ctx_set_ang_rate_order(obtg_ctx*, int elevate_first);
/* The order that is IN EFFECT for the context's present shape -- 0, 1 or 2 as above, negative = error.  A request holds
 * only where its kernels exist: DEG_ELEV = 0 has one order (0); without a products-then-elevation kernel for (deg, R) --
 * deg + 1 outside obtg_fast_kernels & 2, deg > 15, 4 R > 1000 -- the any-degree kernel runs, which elevates first (1),
 * and the double-double pass of order 2 is not run.  Callers that report which order produced their numbers ask here. */
int obtg_ctx_ang_rate_order_in_effect(obtg_ctx*);
int obtg_sync(obtg_ctx*);

/* sizes of one evaluation row's outputs, in doubles */
int obtg_len_temporal_sep(const obtg_ctx*); /* C(N+M,2) * (2n+R+1) */
int obtg_len_speed(const obtg_ctx*);        /* N * (2n+R+1)        */
int obtg_len_ang_rate(const obtg_ctx*);     /* N * (4(n+R)+1)      */
int obtg_num_pairs(const obtg_ctx*);        /* C(N+M,2)            */

/* ---- Bernstein constraint sweeps (host buffers) ----------------------------------------
 * obtg_temporal_sep: _temporalSeparationConstraints (optimization.py:311-346) through the
 *   closure of optimization.py:83-107: for every pair i<j of the N+M objects, in
 *   lexicographic order, ((v_i - v_j).normSquare().elev(R)).cpts - max_sep^2.
 *   normSquare keeps the reference's (d/2) factor (bezier.py:884).
 * obtg_speed: _maxSpeedConstraints / _minSpeedConstraints (optimization.py:349-422):
 *   per vehicle diff() [derivative then elev(1), bezier.py:497-519], normSquare, elev(R);
 *   is_max ? bound^2 - cpts : cpts - bound^2.  tf[B]: final time of each row.
 * obtg_ang_rate: _maxAngularRateConstraints -> _angularRateSqr (optimization.py:425-459,
 *   578-611), dim must be 2: max_rate^2 - num.cpts/den.cpts element-wise (inf/nan kept). */
int obtg_temporal_sep(obtg_ctx*, const double* Y, int B, double max_sep, double* out);
int obtg_speed(obtg_ctx*, const double* Y, const double* tf, int B, double bound, int is_max, double* out);
int obtg_ang_rate(obtg_ctx*, const double* Y, const double* tf, int B, double max_rate, double* out);
/* fused per-pair minimum of the elevated separation control points minus max_sep^2
 * (the `dv.normSquare().min()` variant commented at optimization.py:338 and used by
 * Examples/SequentialSwarm.py:65): out[B][C(N+M,2)]. */
int obtg_temporal_sep_min(obtg_ctx*, const double* Y, int B, double max_sep, double* out);
/* SURVEY.md 8(f) item 4 as the survey words it -- "feed SLSQP only active / near-active constraint rows": per pair the k
 * SMALLEST of its 2n+R+1 elevated separation control points (minus max_sep^2) -- among equal values the lower
 * control-point index is taken first --, listed in ascending control-point INDEX.  1 <= k <= 4 (and k <= 2n+R+1):
 * out_val[B][P][k], out_idx[B][P][k] (nullable) = which control points they are.  k is fixed, so the constraint vector
 * has a constant length P k -- what SLSQP needs -- and in index order a row follows one control point, a polynomial in
 * x, for as long as the membership of the set stands: the rows kink only where the k-th and (k+1)-th smallest cross (in
 * value order they would kink at every crossing inside the set as well).  k = 1 is obtg_temporal_sep_min.  The specialised kernels select in the epilogue of their reduced form (one lane holds a pair's
 * control points: a branch-free insertion into four registers per value); other degrees write the rows to a workspace and
 * select in a second launch.  Same values as the corresponding entries of obtg_temporal_sep, bit for bit.
 * optimization.py:337-338 is where the reference builds the full rows / leaves the minimum commented out. */
int obtg_temporal_sep_active(obtg_ctx*, const double* Y, int B, double max_sep, int k, double* out_val, int* out_idx);
/* the same restricted to pairs [pair_begin, pair_begin+pair_count) of the lexicographic list:
 * out[B][pair_count].  With pair_begin = 0, pair_count = N-1 this is the one-vs-many constraint
 * of Examples/SequentialSwarm.py:43-70 (vehicle 0 against every other one). */
int obtg_temporal_sep_min_range(obtg_ctx*, const double* Y, int B, double max_sep,
                                int pair_begin, int pair_count, double* out);

/* Structured finite differences of the temporal-separation family (the `jac`
 * entry a driver adds next to optimization.py:83-107).  Perturbation t sets element (pert_row[t],
 * pert_col[t]) of the evaluation row Y0[(n_veh*dim)][deg+1] to pert_val[t] -- one control-point
 * coordinate of vehicle v_t = pert_row[t] / dim -- so only the n_obj-1 pairs containing v_t change.
 * out_blk[n_pert][n_obj-1][2n+R+1]: block t holds, for the partners u = 0..n_obj-1, u != v_t in
 * increasing order, the constraint values of pair {v_t, u} under perturbation t; they equal the
 * corresponding entries of obtg_temporal_sep on the fully perturbed row bit for bit, the rest of
 * that row equals the unperturbed evaluation.  n_x (N-1) pair evaluations instead of n_x C(N,2).
 * OBTG_ERR_UNSUPPORTED for shapes outside the specialised kernels (use the batch form then). */
int obtg_temporal_sep_fd(obtg_ctx*, const double* Y0, int n_pert, const int* pert_row, const int* pert_col,
                         const double* pert_val, double max_sep, double* out_blk);
int obtg_temporal_sep_fd_dev(obtg_ctx*, const double* dY0, int n_pert, const int* d_pert_row,
                             const int* d_pert_col, const double* d_pert_val, double max_sep, double* d_out_blk);

/* ---- one curve against K others (Examples/SequentialSwarm.py:43-70, the sequential planner's constraint) ----------
 * out[b][k] = min over the elevated control points of |one_b - many_k|^2 (normSquare's (d/2) factor kept), minus
 * max_sep^2: `dv = vehTraj - tempTraj; dv.normSquare().elev(DEG_ELEV).cpts.min() - maxSep**2` (the example hard-codes
 * elev(10); here DEG_ELEV is the context's).  one[B][dim][deg+1] are B candidates of the one curve (B = 1 for a plain
 * callback, n_x + 1 for a finite-difference batch of the vehicle being planned), many[K][dim][deg+1] the curves it is
 * checked against.  No pair table: the context fixes (dim, deg, DEG_ELEV) only -- its vehicle count is not used -- and K
 * may differ from call to call (the planner's K grows by one per vehicle).  Degrees with a specialised kernel
 * (obtg_fast_kernels(dim, deg) & 1), others OBTG_ERR_UNSUPPORTED. */
int obtg_one_vs_many_min(obtg_ctx*, const double* one, int B, const double* many, int K, double max_sep, double* out /*[B][K]*/);
int obtg_one_vs_many_min_dev(obtg_ctx*, const double* d_one, int B, const double* d_many, int K, double max_sep, double* d_out);

/* ---- same sweeps on DEVICE pointers, asynchronous on the context's stream ---------------
 * pair_begin/pair_count select a contiguous block of the lexicographic pair list (the
 * pair-partitioned multi-GPU mode); out rows then hold only that block:
 * out[B][pair_count*(2n+R+1)].  Pass 0, obtg_num_pairs() for everything. */
int obtg_temporal_sep_dev(obtg_ctx*, const double* dY, int B, double max_sep,
                          int pair_begin, int pair_count, double* d_out);
int obtg_temporal_sep_min_dev(obtg_ctx*, const double* dY, int B, double max_sep,
                              int pair_begin, int pair_count, double* d_out);
int obtg_temporal_sep_active_dev(obtg_ctx*, const double* dY, int B, double max_sep, int k,
                                 int pair_begin, int pair_count, double* d_out_val, int* d_out_idx /*nullable*/);
int obtg_speed_dev(obtg_ctx*, const double* dY, const double* d_tf, int B, double bound, int is_max,
                   double* d_out);
int obtg_ang_rate_dev(obtg_ctx*, const double* dY, const double* d_tf, int B, double max_rate,
                      double* d_out);
/* speed and angular-rate constraints of the same rows in one launch (they share the derivative
 * curves: for d = 2 the speed curve is the angular rate's denominator before squaring,
 * optimization.py:380 vs 605).  Either output may be NULL. */
int obtg_dynamics_dev(obtg_ctx*, const double* dY, const double* d_tf, int B, double speed_bound,
                      int speed_is_max, double max_rate, double* d_out_speed, double* d_out_ang);
/* BOTH speed bounds from one pass.  The reference exposes minSpeedConstraints and maxSpeedConstraints
 * (optimization.py:135-169, 349-422): the same elevated |v|^2 curve, once as cpts - vmin^2 and once as vmax^2 - cpts.
 * While d_out2 is set (device, [B][N*(2n+R+1)]; NULL = off), every pass that writes speed rows through
 * obtg_dynamics[_fd]_dev or obtg_constraint_sweep_dev also writes the rows of this second bound from the curve it
 * has in registers -- no second launch, no second evaluation of the derivative curves. */
int obtg_ctx_set_second_speed_bound(obtg_ctx*, double bound, int is_max, double* d_out2);

/* ---- finite-difference batch on the device --------------------------------------------
 * Replaces the n_x+1 serial calls SciPy's approx_derivative makes: builds
 * dY[B][N*d][n+1] with row 0 = Y and row k = Y with the k-th free control point
 * (interior columns, row-major as x.reshape(numRows,numCols), optimization.py:283)
 * advanced by h.  n_fixed_cols = columns pinned at each end (1: end points; 2: + speed
 * columns).  B <= N*d*(n+1-2*n_fixed_cols) + 1. */
int obtg_fd_batch_dev(obtg_ctx*, const double* dY0, int n_fixed_cols, double h, int B, double* dY);

/* The same batch WITHOUT writing it: a VIEW.  Between obtg_fd_view_begin and obtg_fd_view_end every `_dev` sweep
 * (temporal_sep[_min], speed, ang_rate, dynamics, gjk_swarm, pair_sweep) may be called with dY = NULL and the view's B:
 * it then evaluates the virtual batch whose row 0 is dY0 and whose row b >= 1 is dY0 with its (b-1)-th free control point
 * advanced by h -- bit for bit the rows obtg_fd_batch_dev(dY0, n_fixed_cols, h, B) writes.  Kernels with an on-the-fly
 * form build the rows while staging their inputs (the B x 11 KB batch never travels through HBM); for the others the
 * batch is written to a context buffer, once per view.  Results are identical either way.  dY0 must stay valid until the
 * view ends.  obtg_pair_sweep_fd_dev / obtg_dynamics_fd_dev are the one-call forms (a view around a single sweep);
 * obtg_fd_forms_on_the_fly: bit 0 = the pair sweep is one launch forming its rows itself, bit 1 = the dynamics launch. */
int obtg_fd_view_begin(obtg_ctx*, const double* dY0, int n_fixed_cols, double h, int B);
/* A RANGE of that batch's rows: the view's local row b is batch row row_begin + b (B rows).  What one of G processes
 * opens when an SLSQP iteration's n_x + 1 rows are sharded over G GPUs (distributed.shard_rows): no rank
 * writes or reads rows it does not own.  The structured step's form of it: obtg_constraint_sweep_fd_structured_rows_dev. */
int obtg_fd_view_begin_rows(obtg_ctx*, const double* dY0, int n_fixed_cols, double h, int row_begin, int B);
int obtg_fd_view_end(obtg_ctx*);
int obtg_fd_forms_on_the_fly(const obtg_ctx*);
int obtg_pair_sweep_fd_dev(obtg_ctx*, const double* dY0, int n_fixed_cols, double h, int B, double max_sep,
                           double* d_out_sep, int max_iter, int md_cap, int* d_flag, double* d_p1, double* d_p2,
                           double* d_dist, int* d_nsup, int* d_status);
int obtg_dynamics_fd_dev(obtg_ctx*, const double* dY0, int n_fixed_cols, double h, const double* d_tf, int B,
                         double speed_bound, int speed_is_max, double max_rate, double* d_out_speed, double* d_out_ang);

/* ---- GJK (gjk/gjk.py:230-270 gjkNew, 273-360 minimumDistance) ---------------------------
 * Generic point sets: pts[n_pts][3], poly_off[n_poly+1]; pair k = (pair_a[k], pair_b[k]).
 * Outputs per pair: flag in {-1,0,1} (gjk.py:234-237); p1/p2 = closest points on poly1/poly2
 * and dist when flag == 1 (NaN otherwise); n_support = number of supportPts calls;
 * support_trace (nullable) [n_pairs][trace_cap][2] int16 = (index into poly1, index into
 * poly2) of every supportPts call in order (bit-exact parity target); status as OBTG_ST_*.
 * The reference's decision sequence is followed exactly, including its early exits. */
int obtg_gjk_pairs(obtg_ctx*, const double* pts, int n_pts, const int* poly_off, int n_poly,
                   const int* pair_a, const int* pair_b, int n_pairs, int max_iter, int md_cap,
                   int* flag, double* p1, double* p2, double* dist,
                   short* support_trace, int trace_cap, int* n_support, int* status);

/* Batched swarm sweep: hulls of the N vehicles of each of the B rows (control polygons,
 * 2-D padded with z = 0 as bezier.py:1294-1308 does) plus the static polygons registered
 * with obtg_ctx_set_polygons (object ids N .. N+n_poly-1).  Pair list is device-resident
 * after obtg_ctx_set_hull_pairs.  Outputs are DEVICE arrays:
 * d_flag[B][n_pairs] int32, d_p1/d_p2[B][n_pairs][3], d_dist[B][n_pairs],
 * d_nsup[B][n_pairs] int32 (nullable), d_status[B][n_pairs] int32 (nullable).
 * obtg_ctx_set_polygons drops the pair list (object ids change meaning): the sweeps return OBTG_ERR_ARG until
 * obtg_ctx_set_hull_pairs has been called again. */
int obtg_ctx_set_polygons(obtg_ctx*, const double* pts, int n_pts, const int* poly_off, int n_poly);
int obtg_ctx_set_hull_pairs(obtg_ctx*, const int* pair_a, const int* pair_b, int n_pairs);
int obtg_gjk_swarm_dev(obtg_ctx*, const double* dY, int B, int max_iter, int md_cap,
                       int* d_flag, double* d_p1, double* d_p2, double* d_dist,
                       int* d_nsup, int* d_status);
/* The N x N pair sweep of a batch in ONE launch: obtg_temporal_sep_dev (all pairs) and
 * obtg_gjk_swarm_dev for the same B rows.  The workgroups of the planar gjkNew sweep have the row's
 * control points in LDS anyway; each also writes its share of the row's temporal-separation block
 * (part before, part after its gjkNew phases, staggered between workgroups), whose HBM stores then
 * drain under the VALU-bound gjkNew work of the other wavefronts.  Outputs are those of the two
 * separate calls, bit for bit.  Rows whose hulls exceed 48 KB of LDS (256 vehicles of degree 15) take the
 * tiled form of the same idea when the hull pair list holds every vehicle pair: a chunk of the tiled sweep
 * stages one TA x 64 tile of the pair matrix and writes that tile's separation rows.  3-D rows: the 3-D
 * sweep's workgroups run their part of the separation block first (obtg_constraint_sweep_dev: and of the
 * speed rows), one launch as well.  Point obstacles (constant curves in the separation pairs, optimization.py:86-98; no
 * part of the hull sweep) are staged behind the hull objects by the planar grid: one launch too, for rows within 48 KB.
 * Shapes without such a kernel (DEG_ELEV > 0, de-duplication on, point obstacles with large or 3-D rows) fall back to the
 * two launches. */
int obtg_pair_sweep_dev(obtg_ctx*, const double* dY, int B, double max_sep, double* d_out_sep,
                        int max_iter, int md_cap, int* d_flag, double* d_p1, double* d_p2,
                        double* d_dist, int* d_nsup, int* d_status);
/* EVERY constraint family of the batch in one call: obtg_pair_sweep_dev (temporal separation + gjkNew hull sweep) and
 * obtg_dynamics_dev (max/min speed + angular rate; d_out_ang may be NULL) of the same B rows -- what one evaluation of an
 * SLSQP step needs, as the library's best launch sequence for the shape (ONE launch for planar DEG_ELEV = 0 rows and 3-D
 * rows; two for planar DEG_ELEV > 0: the gjkNew sweep, and the separation rows with the speed / angular-rate groups).  Outputs are those of the two separate calls, bit for bit.  dY may be NULL inside an obtg_fd_view. */
int obtg_constraint_sweep_dev(obtg_ctx*, const double* dY, const double* d_tf, int B, double max_sep, double* d_out_sep,
                              double speed_bound, int speed_is_max, double max_rate, double* d_out_speed,
                              double* d_out_ang, int max_iter, int md_cap, int* d_flag, double* d_p1, double* d_p2,
                              double* d_dist, int* d_nsup, int* d_status);
/* ---- the collective behind the C ABI: one process per GPU, RCCL over xGMI ------------------------
 * For callers that bind this library without PyTorch (INTEGRATION.md route 2); the Python layer's distributed.py does the
 * same through torch.distributed.  RCCL is loaded at run time (librccl.so.1, or the file OBTG_RCCL_LIB names): without it
 * these calls return OBTG_ERR_UNSUPPORTED and nothing else in the library is affected.
 *   rank 0:      obtg_comm_unique_id(id)             -- 128 bytes, handed to the other ranks by the caller's own means
 *                                                       (a file, a socket, MPI: the library opens no connection of its own)
 *   every rank:  obtg_comm_create(&comm, n_ranks, rank, id, device)      (collective: returns when all ranks have called)
 * obtg_comm_all_gather_dev: bytes_per_rank bytes of d_send from every rank into d_recv[n_ranks][bytes_per_rank], on the
 * CONTEXT's stream (ordered with its launches; obtg_sync completes it).
 * obtg_temporal_sep_min_gather_dev: what north_star names -- the swarm's pair list partitioned over the ranks (contiguous
 * balanced blocks of the lexicographic list, the first P mod G ranks one pair more), each rank evaluating the per-pair
 * separation minima of ITS block for the B rows (every rank holds all control points: dY, or an open view with dY = NULL),
 * ONE all-gather of 8 B P bytes in all, and d_min_all[B][P] complete on every rank.  The reference loops over every pair
 * in one process (optimization.py:311-346); the minimum per pair is its commented form at optimization.py:338. */
typedef struct obtg_comm obtg_comm;
int obtg_comm_unique_id(unsigned char* id /*[128]*/);
int obtg_comm_create(obtg_comm** out, int n_ranks, int rank, const unsigned char* id /*[128]*/, int device);
void obtg_comm_destroy(obtg_comm*);
int obtg_comm_size(const obtg_comm*);
int obtg_comm_rank(const obtg_comm*);
const char* obtg_comm_last_error(const obtg_comm*);   /* NULL: the calling thread's last failure without a communicator (obtg_comm_unique_id, obtg_comm_create) */
int obtg_comm_all_gather_dev(obtg_comm*, obtg_ctx*, const void* d_send, void* d_recv, size_t bytes_per_rank);
int obtg_temporal_sep_min_gather_dev(obtg_ctx*, obtg_comm*, const double* dY, int B, double max_sep, double* d_min_all /*[B][P]*/);
/* The two pieces of it for callers with a collective of their own (MPI, a host-staged exchange): rank's block of the pair
 * list, and rank blocks [n_ranks][B * ceil(P / n_ranks)] (rank r's `count_r` minima of row 0, then of row 1, ...: what
 * obtg_temporal_sep_min_dev(pair_begin, pair_count) writes, padded to the largest block) -> rows d_rows[B][P]. */
/* What a rank of a ROW-sharded finite-difference step sends when the per-pair minima of every row are wanted in one place
 * (distributed.SparseMinimaGather): batch row b >= 1 differs from row 0 only in the n_obj - 1 pairs of the vehicle
 * it advances, so for the rows [row_begin, row_begin + n_rows) of the batch over dY0 (obtg_fd_view_begin's rows, row_begin >= 1)
 * d_out[n_rows][n_obj - 1] holds, per row, the minima of exactly those pairs, partners ascending -- evaluated from dY0
 * directly (n_rows (n_obj - 1) pair evaluations; no [B][P] block is formed), the same bits as the corresponding entries of
 * obtg_temporal_sep_min_dev inside the view.  Row 0's P minima are obtg_temporal_sep_min_dev(dY0, 1, ...). */
int obtg_temporal_sep_fd_min_rows_dev(obtg_ctx*, const double* dY0, int n_fixed_cols, double h, int row_begin, int n_rows,
                                      double max_sep, double* d_out);
int obtg_pair_block(const obtg_ctx*, int n_ranks, int rank, int* begin, int* count);
int obtg_unpack_pair_blocks_dev(obtg_ctx*, const double* d_blocks, int B, int n_ranks, double* d_rows);

/* The whole finite-difference step as ONE launch that does not repeat row 0's work (SURVEY.md 8(f) item 1 for every
 * family; what SciPy's approx_derivative needs from optimization.py:83-187 per SLSQP iteration).  The rows of the view
 * (dY0, n_fixed_cols, h, B) differ from row 0 in ONE vehicle each: the launch evaluates row 0 in full and streams its
 * results into all B rows, while one workgroup per row evaluates only the N-1 separation pairs, the hull pairs and the
 * vehicle its advanced control point touches.  Outputs are those of obtg_constraint_sweep_dev inside the same view,
 * bit for bit (the same device functions evaluate every pair); the launch is bound by its stores instead of by gjkNew.
 * Planar shapes (obtg_fast_kernels & 4: deg + 1 in {4, 6, 8, 9, 11}, with or without point obstacles -- constant curves of the separation pair table,
 * optimization.py:86-98 --, angular rate wanted, a row's objects within 40 KB of LDS), any
 * DEG_ELEV with 2 deg + DEG_ELEV + 1 <= 512 -- for DEG_ELEV > 0 (where the brute-force step is two launches) the separation
 * streams are the elevated rows and the dynamics groups are the elevated kernel's; at DEG_ELEV = 0 also deg + 1 = 16 and, for
 * deg + 1 in {11, 16}, rows of up to 158 KB (256 vehicles of degree 15 are 70 KB: two workgroups per CU):
 * OBTG_ERR_UNSUPPORTED otherwise -- the brute-force call gives the same numbers.  d_tf is read per row: the speed /
 * angular-rate rows of row 0 are copied only into rows whose tf equals tf[0] bit for bit, any other row is evaluated in full.  A different evaluation strategy from
 * "every row in full": bench.py reports it as variants.fd_structured, never as its headline value.  The call is a view
 * of its own (begin .. end around its launch): with a view of the caller's open on the context it returns OBTG_ERR_ARG
 * instead of replacing and closing that view.
 * ..._rows_dev: the same for the row RANGE [row_begin, row_begin + B) of the batch (what obtg_fd_view_begin_rows views; d_tf and
 * every output hold the range's B rows): what one of G processes calls when an SLSQP iteration's n_x + 1 rows are sharded over
 * G GPUs.  The unperturbed row is evaluated by every rank (it is the source of the streams); a range that
 * does not hold the batch's row 0 has a perturbed row as its local row 0.  Bit for bit the rows of the whole-batch call. */
int obtg_constraint_sweep_fd_structured_dev(obtg_ctx*, const double* dY0, int n_fixed_cols, double h, const double* d_tf,
                                            int B, double max_sep, double* d_out_sep, double speed_bound, int speed_is_max,
                                            double max_rate, double* d_out_speed, double* d_out_ang, int max_iter, int md_cap,
                                            int* d_flag, double* d_p1, double* d_p2, double* d_dist, int* d_nsup,
                                            int* d_status);
int obtg_constraint_sweep_fd_structured_rows_dev(obtg_ctx*, const double* dY0, int n_fixed_cols, double h, int row_begin,
                                                 const double* d_tf, int B, double max_sep, double* d_out_sep, double speed_bound,
                                                 int speed_is_max, double max_rate, double* d_out_speed, double* d_out_ang,
                                                 int max_iter, int md_cap, int* d_flag, double* d_p1, double* d_p2, double* d_dist,
                                                 int* d_nsup, int* d_status);
/* Finite-difference de-duplication (off by default).  The rows of one
 * SLSQP Jacobian differ from row 0 in ONE vehicle, so all pairs not involving it have row 0's
 * inputs bit for bit.  When on, obtg_gjk_swarm[_dev] compares every row with row 0 (bitwise, per
 * vehicle), runs gjkNew only for pairs with a changed hull and copies row 0's outputs for the
 * rest.  Results are identical to the brute-force sweep for any input. */
int obtg_ctx_set_fd_dedup(obtg_ctx*, int on);
/* Trip-count history of the planar sweep (on by default).  Each obtg_gjk_swarm[_dev] call keeps one
 * byte per (row, pair): the number of support scans gjkNew took.  The next call evaluates every
 * workgroup's pairs in descending order of that count, so that the lanes of a wavefront finish and
 * refill together (SLSQP's consecutive calls see nearly the same geometry).  Scheduling only:
 * every pair is evaluated in full and the outputs do not depend on the order.  Off = list order. */
int obtg_ctx_set_gjk_history(obtg_ctx*, int on);
/* host-buffer form of the same sweep (B rows of Y in, arrays out) */
int obtg_gjk_swarm(obtg_ctx*, const double* Y, int B, int max_iter, int md_cap,
                   int* flag, double* p1, double* p2, double* dist, int* nsup, int* status);

/* Robust curve <-> curve minimum distance (NOT the reference's algorithm): breadth-first
 * branch & bound on the parameter square with bounds that are valid for the curves themselves -- upper: distances
 * between end points of sub-curves; lower: gap of the two control polygons projected on the direction between the
 * sub-curves' mid points.  res[n_pairs][3] = (dist, t1, t2) with dist within a relative eps of the true minimum, or
 * below eps x (largest coordinate) when the curves touch, when status == OBTG_MD_OK; OBTG_MD_NODE_CAP (node budget max_nodes or the internal frontier of 1024 nodes per
 * pair exhausted) and OBTG_MD_DEPTH_CAP (level 48) return the best distance found so far, an upper bound.
 * info[n_pairs][4] = (nodes, levels, largest frontier, status).  Use where `_minDist`'s known defects matter
 * (bezier.py:1283-1408 on gjkNew: non-minimal hull distances used as lower bounds, unbounded loops). */
int obtg_min_dist_robust(obtg_ctx*, const double* curves, int n_curves, int K, const int* pair_a, const int* pair_b,
                         int n_pairs, double eps, int max_nodes, double* res, int* info, int* status);
/* The curve <-> polygon form of the robust search (NOT the reference's `_minDist2Poly`, bezier.py:1411-1496): branch &
 * bound on the curve parameter with bounds that hold for the curve itself -- upper: true distances from sub-curve end
 * points to the polygon's convex hull; lower: the certified lower bound of the true distance between the sub-curve's
 * control hull and the polygon's hull (obtg_gjk_true_pairs' algorithm).  res[n_pairs][5] = (dist, t, closest point on the
 * polygon's hull[3]); info / status as obtg_min_dist_robust. */
int obtg_min_dist2poly_robust(obtg_ctx*, const double* curves, int n_curves, int K, const double* pts, int n_pts,
                              const int* poly_off, int n_poly, const int* pair_curve, const int* pair_poly, int n_pairs,
                              double eps, int max_nodes, double* res, int* info, int* status);
/* True distance between the convex hulls of two point sets (opt-in; NOT gjkNew, which stops at a non-minimal distance
 * on ~30 % of separated pairs and never exits on some 3-D inputs -- SURVEY.md 8(a) G2): a textbook GJK whose exit test
 * is a certificate, dist >= hull distance >= lower with dist - lower <= eps * dist.  Same point-set / pair conventions as
 * obtg_gjk_pairs.  flag 1 separated (p1 / p2 = closest points, dist), 0 intersecting or touching (dist 0);
 * lower, iters, status are nullable; status 0 = converged with that certificate, 1 = iteration cap, 2 = stalled at
 * rounding level before the certificate closed (dist is still a distance between hull points, `lower` still a proven
 * lower bound: compare the two). */
int obtg_gjk_true_pairs(obtg_ctx*, const double* pts, int n_pts, const int* poly_off, int n_poly,
                        const int* pair_a, const int* pair_b, int n_pairs, double eps, int max_iter,
                        int* flag, double* p1, double* p2, double* dist, double* lower, int* iters, int* status);
/* ---- curve <-> curve / curve <-> polygon minimum distance --------------------------------
 * Bezier.minDist -> _minDist (bezier.py:840-852, 1283-1408) and Bezier.minDist2Poly ->
 * _minDist2Poly (bezier.py:854-857, 1411-1496): branch and bound over de Casteljau
 * subdivisions with gjkNew lower bounds.  curves[n_curves][3][K] (2-D curves carry a zero
 * z row).  res[n_pairs][3] = (alpha, t1, t2); for the polygon form res[n_pairs][5] =
 * (alpha, t1, closest point on polygon[3]).  info[n_pairs][4] (nullable) = nodes visited,
 * gjkNew calls, max depth, status; status[n_pairs] as OBTG_MD_*. */
int obtg_min_dist(obtg_ctx*, const double* curves, int n_curves, int K,
                  const int* pair_a, const int* pair_b, int n_pairs,
                  double eps, int max_iter, int md_cap, int max_depth, int max_nodes,
                  double* res, int* info, int* status);
int obtg_min_dist2poly(obtg_ctx*, const double* curves, int n_curves, int K,
                       const double* pts, int n_pts, const int* poly_off, int n_poly,
                       const int* pair_curve, const int* pair_poly, int n_pairs,
                       double eps, int max_iter, int md_cap, int max_depth, int max_nodes,
                       double* res, int* info, int* status);

/* ---- single-curve Bernstein algebra (the Bezier object's methods, batched over rows) ----
 * obtg_bern_elev:   Bezier.elev(R)      bezier.py:469-495   in[rows][n+1]   -> out[rows][n+R+1]
 * obtg_bern_diff:   Bezier.diff()       bezier.py:497-519   in[rows][n+1]   -> out[rows][n+1]  (T = tf-t0)
 * obtg_bern_mul:    Bezier.mul          bezier.py:376-432   a[rows][m+1], b[rows][n+1] -> out[rows][m+n+1]
 * obtg_bern_normsq: Bezier.normSquare() bezier.py:869-889   x[d][n+1]       -> out[2n+1]  ((d/2) quirk kept)
 * obtg_bern_split:  Bezier.split(tDiv)  bezier.py:533-572 -> deCasteljauSplit 985-1027   in[rows][n+1] ->
 *                   left[rows][n+1], right[rows][n+1] at z = (tDiv - t0)/(tf - t0); `right` is in the curve's own
 *                   orientation (the reference reverses deCasteljauSplit's second array, bezier.py:563) */
int obtg_bern_elev(obtg_ctx*, const double* in, int rows, int n, int R, double* out);
int obtg_bern_diff(obtg_ctx*, const double* in, int rows, int n, double T, double* out);
int obtg_bern_mul(obtg_ctx*, const double* a, const double* b, int rows, int m, int n, double* out);
int obtg_bern_normsq(obtg_ctx*, const double* x, int d, int n, double* out);
int obtg_bern_split(obtg_ctx*, const double* in, int rows, int n, double z, double* left, double* right);
/* Bezier.__call__ / Bezier.curve (bezier.py:184-199, 233-258) -> deCasteljauCurve (bezier.py:945-982): every row of
 * cpts[rows][n+1] at every tau[k]: out[rows][n_tau], T = (tau - t0) / (tf - t0), the de Casteljau triangle per sample. */
int obtg_bern_eval(obtg_ctx*, const double* cpts, int rows, int n, const double* tau, int n_tau, double t0, double tf, double* out);

/* ---- objectives (optimization.py:462-489 _euclideanObjective, 503-519 _minAccelObjective,
 * 522-539 _minJerkObjective): sums over vehicles of the elevated |d^k pos/dt^k|^2 control
 * points, k = 2 (accel) or 3 (jerk); tf[B] must hold ONE final time B times, as the reference has a single
 * model['tf'] for these objectives (a batch with differing tf is OBTG_ERR_ARG). */
int obtg_euclidean_obj(obtg_ctx*, const double* Y, int B, double* out /*[B]*/);
int obtg_accel_obj(obtg_ctx*, const double* Y, const double* tf, int B, double* out /*[B]*/);
int obtg_jerk_obj(obtg_ctx*, const double* Y, const double* tf, int B, double* out /*[B]*/);

/* ---- instrumentation -------------------------------------------------------------------
 * When enabled every kernel launch is bracketed by HIP events on the launch stream;
 * obtg_kernel_stats returns the accumulated device time and launch count per kernel id. */
enum {
    OBTG_K_TEMPORAL_SEP = 0, OBTG_K_SPEED = 1, OBTG_K_ANG_RATE = 2, OBTG_K_GJK = 3,
    OBTG_K_MIN_DIST = 4, OBTG_K_FD_BATCH = 5, OBTG_K_BERN = 6, OBTG_K_PAIR_SWEEP = 7, OBTG_K_COUNT = 8
};
/* on: 0 = off, 1 = every kernel, OBTG_PROFILE_ONLY(id) = only launches of that kernel id (two
 * events per launch drain the queue between kernels: 15 % of a 0.25 ms step when all three
 * launches of the step carry them, hence the selective form for timed regions). */
#define OBTG_PROFILE_ONLY_FLAG 0x100
#define OBTG_PROFILE_ONLY(kernel_id) (OBTG_PROFILE_ONLY_FLAG | (kernel_id))
int obtg_set_profiling(obtg_ctx*, int on);
/* sample: events on every `every`-th eligible launch of each kernel id (1 = all, the default) */
int obtg_set_profile_period(obtg_ctx*, int every);
int obtg_kernel_stats(obtg_ctx*, int kernel_id, double* total_ms, long long* launches);
int obtg_reset_kernel_stats(obtg_ctx*);
const char* obtg_kernel_name(int kernel_id);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* OBTG_H */
