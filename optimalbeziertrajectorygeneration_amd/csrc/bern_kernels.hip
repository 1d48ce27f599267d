// Bernstein constraint sweeps for gfx950 (MI355X).
//
// What is computed (reference file:line in the reference checkout):
//   temporal separation  optimization.py:311-346   (v_i - v_j).normSquare().elev(R) - maxSep^2
//   max / min speed      optimization.py:349-422   diff() -> normSquare() -> elev(R)
//   angular rate         optimization.py:425-459, 578-611
// through  bezier.py:347-374 (sub), 497-519 (diff = derivative then elev(1)), 869-889 +
// 1724-1756 (normSquare with the (d/2) factor), 469-495 + 1127-1147 (elev), 376-432 +
// 1183-1208 (mul).
//
// Kernel shape (fast path, k_normsq_elev): the sweep is OUTPUT-bandwidth bound -- one
// evaluation row at N=64, n=10 reads 11 KB and writes 338 KB -- so the design is about
// the stores.  One wave (64 lanes) owns 64 consecutive items (pairs or vehicles); each
// lane keeps its item's source curve (d x (n+1) doubles) and its 2n+1 product
// coefficients in registers (fully unrolled for the instantiated (n, d)), with the folded
// product weights / elevation band arriving as wave-uniform scalar loads.  Results are
// transposed through a per-wave LDS tile so that the wave writes its 64*(2n+R+1) doubles
// (contiguous in the output because pairs are lexicographic) as 16-byte-per-lane
// coalesced stores.  The control points of the vehicles a workgroup needs are staged once
// into LDS (odd pitch => conflict-free ds_read_b64 across lanes).
//
// Everything here is float64.  Parity target is 1e-9 relative, so fused multiply-adds and
// the symmetric folding of the product are allowed in this translation unit.
#include <algorithm>
#include <type_traits>
#include <cstdio>

#include "obtg_internal.h"
#include "bern_device.h"

// LDS per workgroup of the temporal sweep: room for 64-row transposition tiles (one pass per 64-pair group) at two
// workgroups per CU.  36 KB (16- / 32-row tiles, four workgroups): C3 0.083 ms, C4 15.4 ms; 76 KB: 0.077 / 14.3 ms.
constexpr size_t kNsLdsBudget = 76 * 1024;

namespace obtg {

// (the elevated full rows: B fragments, accumulators and A fragments of a group are ~210 registers, and LDS holds two
// workgroups per CU anyway: two waves per SIMD)
template <int NC, int DIM, int MODE /*0 = pairs, 1 = vehicles*/, bool MINONLY, bool ELEV>
__global__ __launch_bounds__(256, (ELEV && !MINONLY) ? 2 : 1) void k_normsq_elev(const NsParams p)
{
    extern __shared__ double lds[];
    const int b = blockIdx.x / p.wgs_per_row, w = blockIdx.x - b * p.wgs_per_row;
    normsq_elev_body<NC, DIM, MODE, MINONLY, ELEV>(p, b, w, lds);
}

// DEG_ELEV > 0, full separation rows of a batch whose rows stage whole (bern_device.h sep_elev_coop_body): four waves
// share every 16-row tile, the elevation matrix stays in their registers (2 x KS fragments each).  Three workgroups
// per CU: 53 KB of LDS each at C5, <= 168 registers.
constexpr int kCoopNTW = 2;            // output tiles per wave: 2n + R + 1 <= 128
template <int NC, int DIM>
__global__ __launch_bounds__(256, (NC > 11 ? 2 : 3)) void k_sep_elev_coop(const NsParams p)
{
    extern __shared__ double lds[];
    const int b = blockIdx.x / p.wgs_per_row, w = blockIdx.x - b * p.wgs_per_row;
    sep_elev_coop_body<NC, DIM, kCoopNTW>(p, b, w, lds);
}

// LDS bytes of sep_elev_coop_body for a planned launch; 0 = the shape is outside it (normsq_elev_body's form runs)
template <int NC, int DIM>
static size_t sep_elev_coop_lds(const NsParams& p)
{
    using S = NsShape<NC, DIM>;
    const int LR = S::L + p.R;
    if (p.R <= 0 || !p.stage_all || p.tiling || LR > 64 * kCoopNTW || p.n_obj >= 65536) return 0;
    static const bool on = !(getenv("OBTG_ELEV_COOP") && getenv("OBTG_ELEV_COOP")[0] == '0');      // (A/B runs)
    if (!on) return 0;
    const size_t lds = sizeof(double) * ((((size_t)p.stage_slots * S::VP + 1) & ~(size_t)1) + ElevCoop<S::L>::lds_doubles(LR)) +
                       sizeof(unsigned) * kWave * (size_t)p.groups_per_wg;
    return lds <= 160 * 1024 ? lds : 0;
}

// =====================================================================================
//  Structured finite differences of the temporal-separation family.
//  Perturbation t replaces ONE element of the evaluation row, i.e. one control point coordinate of
//  vehicle v_t; only the n_obj - 1 pairs that contain v_t change (optimization.py:311-346 is a sum over
//  pairs).  Item = (t, partner): one lane evaluates that pair exactly as k_normsq_elev does (same
//  difference, same product weights, same elevation sums in the same order), so block t of the
//  output equals the corresponding entries of the brute-force finite-difference row bit for bit.
// =====================================================================================
struct TsepFdParams {
    const double* __restrict__ Y0;     // [n_veh*DIM][NC] the evaluation row
    const double* __restrict__ obs;    // [n_obj - n_veh][DIM] point obstacles (constant curves)
    const double* __restrict__ W2;     // folded product weights
    const double* __restrict__ Td;     // the elevation matrix, dense and transposed (NsParams::Td)
    const int* __restrict__ prow;      // [n_pert] row of Y0 that perturbation t touches
    const int* __restrict__ pcol;      // [n_pert] column
    const double* __restrict__ pval;   // [n_pert] the perturbed value itself (x_k + h as the caller rounds it)
    double* __restrict__ out;          // [n_pert][n_obj-1][L+R]; min_only: [n_pert][n_obj-1]
    int n_veh, n_obj, R, n_pert;
    double sign, offset;
    int min_only;                      // 1: per item only the smallest of its L+R values (obtg_temporal_sep_fd_min_rows_dev)
    int fd_row0, fd_fixed;             // prow == nullptr: perturbation t IS row fd_row0 + t (>= 1) of the finite-difference batch
    double fd_h;                       //   over Y0 (obtg_fd_view_begin's rows: free control point (row - 1) advanced by fd_h)
};

template <int NC, int DIM>
__global__ __launch_bounds__(kWave) void k_tsep_fd(const TsepFdParams p)
{
    using S = NsShape<NC, DIM>;
    constexpr int L = S::L;
    const long item = (long)blockIdx.x * kWave + threadIdx.x;
    const int partners = p.n_obj - 1;
    if (item >= (long)p.n_pert * partners) return;
    const int t = (int)(item / partners), uu = (int)(item - (long)t * partners);
    int r, cc;
    double val;
    if (p.prow) { r = p.prow[t]; cc = p.pcol[t]; val = p.pval[t]; }
    else {
        const int free_cols = NC - 2 * p.fd_fixed, kq = p.fd_row0 + t - 1;
        r = kq / free_cols; cc = p.fd_fixed + (kq - r * free_cols);
        val = p.Y0[(size_t)r * NC + cc] + p.fd_h;              // (as k_fd_batch and the views form it)
    }
    const int v = r / DIM, rq = r - v * DIM;
    const int u = uu < v ? uu : uu + 1;
    double a[DIM][NC];
#pragma unroll
    for (int q = 0; q < DIM; ++q)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            double xv = p.Y0[(size_t)(v * DIM + q) * NC + c];
            if (q == rq && c == cc) xv = val;
            const double xu = u < p.n_veh ? p.Y0[(size_t)(u * DIM + q) * NC + c] : p.obs[(u - p.n_veh) * DIM + q];
            a[q][c] = v < u ? xv - xu : xu - xv;      // pairs are (i, j) with i < j: v_i - v_j
        }
    double cf[L];
    normsq_coeffs<NC, DIM>(a, as_ctab(p.W2), cf);
    const int LR = L + p.R;
    if (p.min_only) {
        double m = INFINITY;
        if (p.R == 0) {
#pragma unroll
            for (int k = 0; k < L; ++k) m = fmin(m, p.sign * cf[k] + p.offset);
        } else {
            const ctab_t Td = as_ctab(p.Td);
            double ch[L];
#pragma unroll
            for (int j = 0; j < L; ++j) ch[j] = p.sign * cf[j];
            for (int k = 0; k < LR; ++k) m = fmin(m, elev_at<L>(ch, Td + k * L, p.offset));
        }
        p.out[item] = m;
        return;
    }
    double* o = p.out + (size_t)item * LR;
    if (p.R == 0) {
#pragma unroll
        for (int k = 0; k < L; ++k) o[k] = p.sign * cf[k] + p.offset;
    } else {
        // the elevation exactly as normsq_elev_body's matrix instructions form it (elev_at: the same chain of fused multiply-adds)
        const ctab_t Td = as_ctab(p.Td);
        double ch[L];
#pragma unroll
        for (int j = 0; j < L; ++j) ch[j] = p.sign * cf[j];
        for (int k = 0; k < LR; ++k) o[k] = elev_at<L>(ch, Td + k * L, p.offset);
    }
}

// =====================================================================================
//  One-vs-many separation minima (Examples/SequentialSwarm.py:43-70): the sequential planner's constraint pairs
//  ONE curve with K others -- `dv = vehTraj - tempTraj; dv.normSquare().elev(10).cpts.min() - maxSep**2` -- so there
//  is no C(N,2) pair table and K grows by one per planned vehicle.  Item = (b, k): candidate b of the `one` curves
//  against curve k of the `many`; one lane evaluates the pair exactly as k_normsq_elev's MINONLY form does (same
//  difference, product weights and elevation sums in the same order) and keeps the minimum in the lane.
// =====================================================================================
struct OneManyParams {
    const double* __restrict__ one;    // [B][DIM][NC]
    const double* __restrict__ many;   // [K][DIM][NC]
    const double* __restrict__ W2;     // folded product weights
    const double* __restrict__ Td;     // the elevation matrix, dense and transposed (NsParams::Td)
    double* __restrict__ out;          // [B][K]
    int B, K, R;
    double sign, offset;
};

template <int NC, int DIM>
__global__ __launch_bounds__(256) void k_one_vs_many(const OneManyParams p)
{
    using S = NsShape<NC, DIM>;
    constexpr int L = S::L;
    const long item = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= (long)p.B * p.K) return;
    const int b = (int)(item / p.K), k = (int)(item - (long)b * p.K);
    const double* o = p.one + (size_t)b * DIM * NC;
    const double* m = p.many + (size_t)k * DIM * NC;
    double a[DIM][NC];
#pragma unroll
    for (int q = 0; q < DIM; ++q)
#pragma unroll
        for (int c = 0; c < NC; ++c) a[q][c] = o[q * NC + c] - m[q * NC + c];
    double cf[L];
    normsq_coeffs<NC, DIM>(a, as_ctab(p.W2), cf);
    double mn;
    if (p.R == 0) {
        mn = cf[0];
#pragma unroll
        for (int j = 1; j < L; ++j) mn = fmin(mn, cf[j]);
        mn = p.sign * mn + p.offset;
    } else {
        const int LR = L + p.R;
        const ctab_t Td = as_ctab(p.Td);
        double ch[L];
#pragma unroll
        for (int j = 0; j < L; ++j) ch[j] = p.sign * cf[j];
        mn = INFINITY;
        for (int kk = 0; kk < LR; ++kk) mn = fmin(mn, elev_at<L>(ch, Td + kk * L, p.offset));
    }
    p.out[item] = mn;
}

// =====================================================================================
//  angular rate, fast path (R == 0): one vehicle per lane
// =====================================================================================
// The degree-4n stage can be split over kDynParts waves per group of 64 vehicles (balanced
// constexpr k-ranges).  Measured at C3 after the tables became scalar loads: 1 part 0.030 ms,
// 2 parts 0.039 ms, 4 parts 0.043 ms -- every part repeats the derivative / degree-2n stage, and
// that redundancy costs more than the extra waves hide.  Kept at 1.
constexpr int kDynParts = 1;
constexpr int dyn_cost(int k, int n2) { return k / 2 - (k - n2 > 0 ? k - n2 : 0) + 1 + 6; }
constexpr int dyn_bound(int n2, int q)
{
    const int L = 2 * n2 + 1;
    long T = 0;
    for (int k = 0; k < L; ++k) T += dyn_cost(k, n2);
    const long target = T * q / kDynParts;
    long cum = 0;
    for (int k = 0; k < L; ++k) {
        if (cum >= target) return k;
        cum += dyn_cost(k, n2);
    }
    return L;
}

template <int NC, int PART>
__device__ __forceinline__ void dyn_final(const AngParams& p, const double (&num1)[2 * NC - 1],
                                          const double (&den1)[2 * NC - 1], double* tile, size_t grow,
                                          int n_valid, int lane)
{
    constexpr int N = NC - 1, L2 = 2 * N + 1, L4 = 4 * N + 1;
    constexpr int K0 = dyn_bound(2 * N, PART), K1 = dyn_bound(2 * N, PART + 1);
    const ctab_t W22n = as_ctab(p.W22n);
    // num = num1^2, den = den1^2 (degree 4n); constraint = w^2 - num.cpts / den.cpts
#pragma unroll
    for (int k = K0; k < K1; ++k) {
        double sn = 0.0, sd = 0.0;
#pragma unroll
        for (int j = (k - 2 * N > 0 ? k - 2 * N : 0); 2 * j <= k; ++j) {
            const double wkj = W22n[k * L2 + j];
            sn = fma(wkj, num1[j] * num1[k - j], sn);
            sd = fma(wkj, den1[j] * den1[k - j], sd);
        }
        tile[lane * L4 + (k - K0)] = p.w2 - sn / sd;
    }
    wave_sync();
    if (K0 == 0 && K1 == L4) flush_full<L4, L4>(tile, p.out, grow, n_valid, lane);   // whole rows: contiguous
    else flush_chunk<L4>(tile, p.out, grow, L4, K0, K1 - K0, n_valid, lane);
}

// angular rate (optimization.py:425-459, 578-611) and, from the same derivatives, the speed
// constraint (optimization.py:349-422): for d = 2 the speed curve (d/2)(xD^2 + yD^2) IS den1.
template <int NC>
__global__ __launch_bounds__(kWave) void k_dynamics(const AngParams p)
{
    constexpr int N = NC - 1, L2 = 2 * N + 1, L4 = 4 * N + 1;
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const int part = blockIdx.x % kDynParts;
    const int it0 = (blockIdx.x / kDynParts) * kWave;
    if (it0 >= p.total) return;
    const bool want_ang = p.out != nullptr;
    if (!want_ang && part != 0) return;
    const int n_valid = min(kWave, p.total - it0);
    const int item = min(it0 + lane, p.total - 1);
    const int b = item / p.n_veh;
    double x[NC], y[NC];
    load_item_xy<NC>(p, item, b, x, y);                // rows (b, veh) are contiguous
    const double val = (double)N / p.tf[b];
    double xD[NC], yD[NC], xDD[NC], yDD[NC];
    diff_elev1<NC>(x, val, xD);
    diff_elev1<NC>(y, val, yD);
    diff_elev1<NC>(xD, val, xDD);
    diff_elev1<NC>(yD, val, yDD);

    // num1 = yDD*xD - xDD*yD,  den1 = xD*xD + yD*yD     (degree 2n, optimization.py:603-605)
    const ctab_t Wn = as_ctab(p.Wn), W2n = as_ctab(p.W2n);
    double num1[L2], den1[L2];
#pragma unroll
    for (int k = 0; k < L2; ++k) {
        double s1 = 0.0, s2 = 0.0, sd = 0.0;
        if (want_ang) {
#pragma unroll
            for (int j = (k - N > 0 ? k - N : 0); j <= (N < k ? N : k); ++j) {
                const double wkj = Wn[k * NC + j];
                s1 = fma(wkj, yDD[j] * xD[k - j], s1);
                s2 = fma(wkj, xDD[j] * yD[k - j], s2);
            }
        }
#pragma unroll
        for (int j = (k - N > 0 ? k - N : 0); 2 * j <= k; ++j)
            sd = fma(W2n[k * NC + j], fma(xD[j], xD[k - j], yD[j] * yD[k - j]), sd);
        num1[k] = s1 - s2;
        den1[k] = sd;
    }
    for (int which = 0; which < ((p.out_speed && part == 0) ? (p.out_speed2 ? 2 : 1) : 0); ++which) {
        const double sgn = which ? p.sp2_sign : p.sp_sign, off = which ? p.sp2_offset : p.sp_offset;
#pragma unroll
        for (int k = 0; k < L2; ++k) lds[lane * L2 + k] = sgn * den1[k] + off;
        wave_sync();
        flush_full<L2, L2>(lds, which ? p.out_speed2 : p.out_speed, (size_t)it0 * L2, n_valid, lane);
        wave_sync();
    }
    if (!want_ang) return;
    const size_t grow = (size_t)it0 * L4;
    switch (part) {
        case 0: dyn_final<NC, 0>(p, num1, den1, lds, grow, n_valid, lane); break;
        case 1: dyn_final<NC, 1>(p, num1, den1, lds, grow, n_valid, lane); break;
        case 2: dyn_final<NC, 2>(p, num1, den1, lds, grow, n_valid, lane); break;
        default: dyn_final<NC, 3>(p, num1, den1, lds, grow, n_valid, lane); break;
    }
}

// Two-wave form of k_dynamics for the angular-rate case.  One lane per (row, vehicle) leaves the
// chip with about one wavefront per SIMD (C3: 1153 groups of 64 items), so every scalar table load
// and every dependent FMA chain is exposed.  Here a workgroup of two waves shares a group: wave 0
// takes the denominator side (den1 = xD^2 + yD^2, the speed constraint, den = den1^2), wave 1 the
// numerator side (num1 = yDD xD - xDD yD, num = num1^2); they swap what the other needs for its
// share of the 4n+1 quotients through LDS.  Only the derivative stage is computed twice.
template <int NC>
__global__ __launch_bounds__(2 * kWave) void k_dynamics2(const AngParams p)
{
    extern __shared__ double lds[];
    dynamics2_group<NC>(p, lds, (int)blockIdx.x);
}

// Angular rate (and speed) with DEG_ELEV = R > 0 (optimization.py:425-459, 578-611 after `pos.elev(R)`).
// The reference elevates the position to degree m = n + R first and forms every product at the elevated
// degree (4m + 1 = 441 coefficients at R = 100, from degree-2m operands).  Degree elevation commutes with diff(),
// mul() and add(): the degree-4m control points of num and den are elev(., 4R) of the degree-4n control points the
// R = 0 kernel forms from the ORIGINAL control points, and the speed rows are elev(den1, R).  So: phase A = the
// degree-4n num / den of k_dynamics2 (lane = item, everything in registers), phase B = two banded elevations and
// the element-wise quotient.  Elevation is a binomially scaled convolution,
//     elev(a, Q)_k = (1 / C(P+Q, k)) sum_j [C(P, j) a_j] C(Q, k-j),
// so all output columns share ONE weight row C(Q, .): a block of 8 columns walks a window of that row with one
// wave-uniform scalar per step (8 + 8 independent FMA chains), instead of fetching 4n+1 weights per column; in the
// quotient num_k / den_k the factor 1 / C(P+Q, k) cancels and is never applied.  2 (4n+1) FMAs per output value
// against the generic kernel's degree-4m convolutions.  A workgroup is four waves on the same 64 items; each
// repeats phase A (8 % of its work) and takes every fourth 32-column chunk, transposed through a per-wave LDS tile
// so that stores are 256-byte runs.
template <int NC>
__global__ __launch_bounds__(4 * kWave, (NC > 11 ? 1 : 3)) void k_dynamics_elev(const AngElevParams q)
{
    extern __shared__ double lds[];
    dynamics_elev_group<NC>(q, lds, (int)blockIdx.x);
}

// =====================================================================================
//  obtg_ctx_set_ang_rate_order(2): the angular-rate rows of near-stop vehicles once more, in double-double
// =====================================================================================
// Where a vehicle nearly stops, den = (xD^2 + yD^2)^2 falls orders of magnitude below its size elsewhere on the curve and
// its elevated control points are sums of terms far larger than themselves: ANY float64 evaluation of
// optimization.py:578-611 -- the reference's own included, 3e-9 from the exact value on the fixture's vehicle -- loses
// what the cancellation takes.  The dynamics kernels list such items (a |v|^2 control point three
// orders below the largest); this kernel evaluates the whole chain for them -- derivatives, the degree-2n curves, their
// squares, the two elevations by 4R, the quotient -- in double-double (two_sum / fma two_prod: ~1e-31 per operation) from
// tables held as (hi, lo) pairs, and rounds once at the end: every element within a few 1e-16 of the exact rational
// value.  One wave per item, lane = output column; the chain up to degree 4n is computed by every lane (the items are
// few: one in a few hundred vehicles).
#pragma clang fp contract(off)
struct dd_t { double hi, lo; };
__device__ __forceinline__ dd_t dd_mk(double hi, double lo = 0.0) { dd_t r; r.hi = hi; r.lo = lo; return r; }
__device__ __forceinline__ dd_t dd_qsum(double a, double b) { const double s = a + b; return dd_mk(s, b - (s - a)); }
__device__ __forceinline__ dd_t dd_sum2(double a, double b)
{
    const double s = a + b, bb = s - a;
    return dd_mk(s, (a - (s - bb)) + (b - bb));
}
__device__ __forceinline__ dd_t dd_add(dd_t a, dd_t b)
{
    dd_t s = dd_sum2(a.hi, b.hi);
    const dd_t t = dd_sum2(a.lo, b.lo);
    s.lo += t.hi;
    s = dd_qsum(s.hi, s.lo);
    s.lo += t.lo;
    return dd_qsum(s.hi, s.lo);
}
__device__ __forceinline__ dd_t dd_neg(dd_t a) { return dd_mk(-a.hi, -a.lo); }
__device__ __forceinline__ dd_t dd_sub(dd_t a, dd_t b) { return dd_add(a, dd_neg(b)); }
__device__ __forceinline__ dd_t dd_mul(dd_t a, dd_t b)
{
    const double p = a.hi * b.hi;
    double e = fma(a.hi, b.hi, -p);
    e += a.hi * b.lo + a.lo * b.hi;
    return dd_qsum(p, e);
}
__device__ __forceinline__ dd_t dd_div(dd_t a, dd_t b)
{
    if (b.hi == 0.0 || !(fabs(b.hi) < INFINITY) || !(fabs(a.hi) < INFINITY)) return dd_mk(a.hi / b.hi);   // 0/0, x/0: as float64 has them
    const double q1 = a.hi / b.hi;
    dd_t r = dd_sub(a, dd_mul(b, dd_mk(q1)));
    const double q2 = r.hi / b.hi;
    r = dd_sub(r, dd_mul(b, dd_mk(q2)));
    const double q3 = r.hi / b.hi;
    dd_t q = dd_qsum(q1, q2);
    return dd_add(q, dd_mk(q3));
}
__device__ __forceinline__ dd_t dd_ld(const double* t, int i) { return dd_mk(t[2 * i], t[2 * i + 1]); }

struct AngDdParams {
    AngParams a;                     // the batch (or the view's row), tf, out, w2
    const int* __restrict__ flags;   // flags[0] = count, flags[1 ..] = items
    const double* __restrict__ tab;  // angrate_dd_tables
    DdTables off;
    int R, cap;
};

// d = elev(1) of the derivative of pp (Bezier.diff(), bezier.py:497-519), all in LDS, lane = control point
template <int NC>
__device__ __forceinline__ void dd_diff_elev1(const dd_t* pp, dd_t val, const double* ratio, dd_t* t, dd_t* d, int lane)
{
    constexpr int N = NC - 1;
    if (lane < N) t[lane] = dd_mul(dd_sub(pp[lane + 1], pp[lane]), val);
    __syncthreads();
    if (lane == 0) d[0] = t[0];
    else if (lane == N) d[N] = t[N - 1];
    else if (lane < N) d[lane] = dd_add(dd_mul(t[lane - 1], dd_ld(ratio, lane)), dd_mul(t[lane], dd_ld(ratio, N - lane)));
    __syncthreads();
}

template <int NC>
__global__ __launch_bounds__(kWave) void k_angrate_dd(const AngDdParams q)
{
    constexpr int N = NC - 1, L2 = 2 * N + 1, L4 = 4 * N + 1;
    static_assert(L2 <= kWave, "one lane per degree-2n coefficient");
    extern __shared__ double lds[];
    dd_t* xs = reinterpret_cast<dd_t*>(lds);            // x, y, xD, yD, xDD, yDD [NC] each, t [NC] scratch
    dd_t* ys = xs + NC; dd_t* xD = ys + NC; dd_t* yD = xD + NC; dd_t* xDD = yD + NC; dd_t* yDD = xDD + NC; dd_t* tt = yDD + NC;
    dd_t* num1 = tt + NC;                               // [L2] degree-2n numerator yDD xD - xDD yD
    dd_t* den1 = num1 + L2;                             // [L2] xD^2 + yD^2
    dd_t* numl = den1 + L2;                             // [L4] num1^2, times C(4n, k)
    dd_t* denl = numl + L4;                             // [L4] den1^2, times C(4n, k)
    const AngParams& p = q.a;
    const int lane = threadIdx.x;
    const int count = min(q.flags[0], q.cap);
    const int L4R = L4 + 4 * q.R;
    const double* wn = q.tab + q.off.wn; const double* w2n = q.tab + q.off.w2n; const double* w22n = q.tab + q.off.w22n;
    const double* ratio = q.tab + q.off.ratio; const double* row4 = q.tab + q.off.row4; const double* sc4 = q.tab + q.off.sc4;
    for (int fi = blockIdx.x; fi < count; fi += gridDim.x) {
        const int item = q.flags[1 + fi];
        const int b = item / p.n_veh;
        __syncthreads();                                 // (the previous item's arrays have been read)
        {
            double x[NC], y[NC];
            load_item_xy<NC>(p, item, b, x, y);
#pragma unroll
            for (int c = 0; c < NC; ++c) if (lane == c) { xs[c] = dd_mk(x[c]); ys[c] = dd_mk(y[c]); }
        }
        __syncthreads();
        const dd_t val = dd_div(dd_mk((double)N), dd_mk(p.tf[b]));
        dd_diff_elev1<NC>(xs, val, ratio, tt, xD, lane);
        dd_diff_elev1<NC>(ys, val, ratio, tt, yD, lane);
        dd_diff_elev1<NC>(xD, val, ratio, tt, xDD, lane);
        dd_diff_elev1<NC>(yD, val, ratio, tt, yDD, lane);
        if (lane < L2) {
            const int k = lane;
            dd_t s = dd_mk(0.0), sd = dd_mk(0.0);
            for (int j = (k - N > 0 ? k - N : 0); j <= (N < k ? N : k); ++j)
                s = dd_add(s, dd_mul(dd_ld(wn, k * NC + j), dd_sub(dd_mul(yDD[j], xD[k - j]), dd_mul(xDD[j], yD[k - j]))));
            for (int j = (k - N > 0 ? k - N : 0); 2 * j <= k; ++j)
                sd = dd_add(sd, dd_mul(dd_ld(w2n, k * NC + j), dd_add(dd_mul(xD[j], xD[k - j]), dd_mul(yD[j], yD[k - j]))));
            num1[k] = s;
            den1[k] = sd;
        }
        __syncthreads();
        for (int k = lane; k < L4; k += kWave) {         // the degree-4n squares, pre-scaled by C(4n, k) for the convolution form
            dd_t sn = dd_mk(0.0), sd = dd_mk(0.0);
            for (int j = (k - 2 * N > 0 ? k - 2 * N : 0); 2 * j <= k; ++j) {
                const dd_t w = dd_ld(w22n, k * L2 + j);
                sn = dd_add(sn, dd_mul(w, dd_mul(num1[j], num1[k - j])));
                sd = dd_add(sd, dd_mul(w, dd_mul(den1[j], den1[k - j])));
            }
            numl[k] = dd_mul(sn, dd_ld(sc4, k));
            denl[k] = dd_mul(sd, dd_ld(sc4, k));
        }
        __syncthreads();
        for (int k = lane; k < L4R; k += kWave) {        // elevation by 4R (the factor 1 / C(4n + 4R, k) cancels in the quotient)
            dd_t sn = dd_mk(0.0), sd = dd_mk(0.0);
            const int j0 = k - 4 * q.R > 0 ? k - 4 * q.R : 0, j1 = k < L4 - 1 ? k : L4 - 1;
            for (int j = j0; j <= j1; ++j) {
                const dd_t w = dd_ld(row4, k - j);
                sn = dd_add(sn, dd_mul(w, numl[j]));
                sd = dd_add(sd, dd_mul(w, denl[j]));
            }
            const dd_t r = dd_sub(dd_mk(p.w2), dd_div(sn, sd));
            p.out[(size_t)item * L4R + k] = r.hi + r.lo;
        }
    }
}
#pragma clang fp contract(fast)

// DEG_ELEV > 0: the elevated separation rows and the elevated speed / angular-rate rows of a batch in ONE launch (the
// brute-force step of such a shape is then two launches: this and the gjkNew sweep).  The separation kernel is bound by
// its stores, the dynamics kernel by its FMAs; their workgroups are interleaved over the block ids (pattern rotated by
// one slot per 16 ids: every XCD sees both kinds) so that each CU has one of each at most times.
struct SepDynElevParams {
    NsParams ts;
    AngElevParams dyn;
    int n_kind[2], per16[2];           // separation workgroups (row, share of the row's groups) | dynamics groups
    unsigned char pat[16], rank[16];
};

template <int NC>
__global__ __launch_bounds__(4 * kWave, (NC > 11 ? 1 : 3)) void k_sep_dynamics_elev(const SepDynElevParams sp)
{
    extern __shared__ double lds[];
    const int grp = (int)blockIdx.x >> 4, slot = ((int)blockIdx.x + grp) & 15;
    const int kind = sp.pat[slot];
    const int id = grp * sp.per16[kind] + sp.rank[slot];
    if (id >= sp.n_kind[kind]) return;
    if (kind == 1) {
        dynamics_elev_group<NC>(sp.dyn, lds, id);
        return;
    }
    const int b = id / sp.ts.wgs_per_row;
    sep_elev_coop_body<NC, 2, kCoopNTW>(sp.ts, b, id - b * sp.ts.wgs_per_row, lds);
}

// =====================================================================================
//  generic path: any degree / elevation, one wave per item, operands in LDS, products as
//  binomially scaled convolutions:  c_k = (1/C(m+n,k)) * sum_j [C(m,j) a_j][C(n,k-j) b_{k-j}]
// =====================================================================================
struct GenParams {
    const double* __restrict__ Y;
    const double* __restrict__ obs;
    const double* __restrict__ tf;
    const int2* __restrict__ pairs;
    const double* __restrict__ bin;   // concatenated binomial rows
    double* __restrict__ out;
    int n_veh, n_obj, dim, n, R;
    int item_begin, item_count, B;
    int o_n, o_2n, o_R, o_2nR;        // offsets of rows C(n,.), C(2n,.), C(R,.), C(2n+R,.)
    int o_m, o_2m, o_4m, o_1m;        // ang-rate: rows for m = n+R, 2m, 4m and C(m-1,.) (unused)
    double sign, offset;
    int min_only;
};

// out[k] (k < L_out, lanes strided) = (1/bo[k]) * sum_j ah[j] * bh[k-j],  ah: la entries, bh: lb entries
__device__ __forceinline__ double conv_at(const double* ah, int la, const double* bh, int lb, int k)
{
    double s = 0.0;
    const int j0 = max(0, k - (lb - 1)), j1 = min(la - 1, k);
    for (int j = j0; j <= j1; ++j) s = fma(ah[j], bh[k - j], s);
    return s;
}

template <int MODE>
__global__ __launch_bounds__(kWave) void k_generic_normsq_elev(const GenParams p)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const int n = p.n, nc = n + 1, L = 2 * n + 1, LR = L + p.R, dim = p.dim;
    const long gi = blockIdx.x;
    const int b = (int)(gi / p.item_count);
    const int item = p.item_begin + (int)(gi - (long)b * p.item_count);
    double* ah = lds;               // [dim][nc]   C(n,j) * a_j
    double* ch = lds + dim * nc;    // [L]         C(2n,j) * c_j
    double* tm = ch + L;            // [dim][nc]   scratch (vehicle mode)
    const double* bn = p.bin + p.o_n;
    const double* b2n = p.bin + p.o_2n;
    const double* Yrow = p.Y + (size_t)b * p.n_veh * dim * nc;

    if (MODE == 0) {
        const int2 ij = p.pairs[item];
        for (int e = lane; e < dim * nc; e += kWave) {
            const int q = e / nc, c = e - q * nc;
            const double vi = ij.x < p.n_veh ? Yrow[(size_t)ij.x * dim * nc + e] : p.obs[(ij.x - p.n_veh) * dim + q];
            const double vj = ij.y < p.n_veh ? Yrow[(size_t)ij.y * dim * nc + e] : p.obs[(ij.y - p.n_veh) * dim + q];
            ah[e] = (vi - vj) * bn[c];
        }
    } else {
        const double val = (double)n / p.tf[b];
        const double* v = Yrow + (size_t)item * dim * nc;
        for (int e = lane; e < dim * nc; e += kWave) {
            const int q = e / nc, c = e - q * nc;
            tm[e] = (c < n) ? v[q * nc + c] * (-val) + v[q * nc + c + 1] * val : 0.0;
        }
        __syncthreads();
        for (int e = lane; e < dim * nc; e += kWave) {
            const int q = e / nc, c = e - q * nc;
            double d;
            if (c == 0) d = tm[q * nc];
            else if (c == n) d = tm[q * nc + n - 1];
            else d = tm[q * nc + c - 1] * ((double)c / (double)n) + tm[q * nc + c] * ((double)(n - c) / (double)n);
            ah[e] = d * bn[c];
        }
    }
    __syncthreads();
    const size_t row = ((size_t)b * p.item_count + (size_t)(item - p.item_begin));
    double mloc = INFINITY;
    for (int k = lane; k < L; k += kWave) {
        double s = 0.0;
        for (int q = 0; q < dim; ++q) s += conv_at(ah + q * nc, nc, ah + q * nc, nc, k);
        const double c = (0.5 * dim) * s / b2n[k];
        if (p.R == 0) {
            if (p.min_only) mloc = fmin(mloc, c);
            else p.out[row * LR + k] = p.sign * c + p.offset;
        } else ch[k] = c * b2n[k];
    }
    if (p.R > 0) {
        __syncthreads();
        const double* bR = p.bin + p.o_R;
        const double* b2nR = p.bin + p.o_2nR;
        for (int k = lane; k < LR; k += kWave) {
            const double s = conv_at(ch, L, bR, p.R + 1, k) / b2nR[k];
            if (p.min_only) mloc = fmin(mloc, s);
            else p.out[row * LR + k] = p.sign * s + p.offset;
        }
    }
    if (p.min_only) {
        for (int o = 32; o > 0; o >>= 1) mloc = fmin(mloc, __shfl_down(mloc, o));
        if (lane == 0) p.out[row] = p.sign * mloc + p.offset;
    }
}

// Register-tiled convolution for the long products of the generic angular-rate kernel.
// Lane owns 8 consecutive outputs k0..k0+7:  s[i] = sum_t ah[t] * bp[Q - t + i],  Q = pad + k0,
// where bp is the zero-padded copy of the second operand (`pad` leading zeros) and ah is padded
// with zeros to a multiple of 8 (la8).  The 8-wide window of bp slides by one element per step and
// lives in registers as a circular buffer (slot (i - t) mod 8), so a step costs one broadcast LDS
// read (ah[t]) and one vector LDS read (the new window element) for 8 FMAs -- 0.25 LDS reads per
// FMA instead of 2 for the one-output-per-lane form.
template <int T = 8>
__device__ __forceinline__ void conv_tile(const double* ah, int la8, const double* bp, int pad, int k0,
                                          double (&s)[T])
{
    static_assert(T == 4 || T == 8, "window sizes with a cheap modulus");
    const double* q = bp + pad + k0;
    double Rw[T];
#pragma unroll
    for (int i = 0; i < T; ++i) { Rw[i] = q[i]; s[i] = 0.0; }
    for (int j = 0; j < la8; j += T) {          // la8 is a multiple of 8
#pragma unroll
        for (int u = 0; u < T; ++u) {
            const double a = ah[j + u];
#pragma unroll
            for (int i = 0; i < T; ++i) s[i] = fma(a, Rw[(i - u) & (T - 1)], s[i]);
            Rw[(-(u + 1)) & (T - 1)] = q[-(j + u + 1)];      // window element 0 of step t+1
        }
    }
}
__device__ __forceinline__ void conv_tile8(const double* ah, int la8, const double* bp, int pad, int k0, double (&s)[8])
{
    conv_tile<8>(ah, la8, bp, pad, k0, s);
}

// Balanced schedule of a product  c = a * b  with len(a) = len(b) = L data entries (zero beyond).  Output tile t
// (T consecutive coefficients) needs j in [lo(t), hi(t)) -- a triangle over t, longest in the middle -- and
// tiles (t, H-1-t), (H+t, Tn-1-t) have complementary lengths (Tn tiles, H = Tn / 2), so a lane that walks such
// a pair in ONE loop does about len(a) + T steps whatever t is: half the lanes of "one tile per lane" for the
// same time, and no multiplications by padding zeros.  Lanes 0..H-1 of a half-wave take the H pairs; the two
// half-waves run two products side by side.  Every tile is summed by one lane in ascending j, as conv_tile does.
// b is addressed as bq[k - j] and must be readable (zeros) for T entries before its start.
__host__ __device__ inline bool conv_pairs_ok(int n_out, int T)
{
    const int Tn = (n_out + T - 1) / T;
    return Tn % 4 == 0 && Tn / 2 <= 32;
}

template <int T>
__device__ __forceinline__ void conv_pairs(const double* ah, const double* bq, int L, int La8, int n_out, int slot,
                                           int& ta, int& tb, double (&first)[T], double (&second)[T])
{
    const int Tn = (n_out + T - 1) / T, H = Tn / 2;
    const bool active = slot < H;
    const bool rising = slot < H / 2;
    const int i0 = rising ? slot : slot - H / 2;
    ta = rising ? i0 : Tn - 1 - i0;                    // the short tile first
    tb = rising ? H - 1 - i0 : H + i0;
    auto lo_of = [&](int t) { const int v = T * t - (L - 1); return v > 0 ? (v & ~(T - 1)) : 0; };
    auto hi_of = [&](int t) { return min(La8, T * t + T); };
    const int ja0 = lo_of(ta), la = active ? hi_of(ta) - ja0 : 0;
    const int jb0 = lo_of(tb), lb = active ? hi_of(tb) - jb0 : 0;
    int total = la + lb;
#pragma unroll
    for (int msk = 32; msk >= 1; msk >>= 1) total = max(total, __shfl_xor(total, msk));   // wave-uniform trip count
    double Rw[T];
    const double* q = bq + T * ta;
    int j = ja0, left = la;                            // current segment: tile pointer q, next j, steps left
    bool on_second = false;
#pragma unroll
    for (int i = 0; i < T; ++i) { Rw[i] = q[i - j]; second[i] = 0.0; first[i] = 0.0; }
    for (int v = 0; v < total; v += T) {
        if (left == 0 && !on_second) {                 // switch to the long tile
#pragma unroll
            for (int i = 0; i < T; ++i) { first[i] = second[i]; second[i] = 0.0; }
            q = bq + T * tb; j = jb0; left = lb; on_second = true;
#pragma unroll
            for (int i = 0; i < T; ++i) Rw[i] = q[i - j];
        }
        if (left > 0) {
#pragma unroll
            for (int u = 0; u < T; ++u) {
                const double a = ah[j + u];
#pragma unroll
                for (int i = 0; i < T; ++i) second[i] = fma(a, Rw[(i - u) & (T - 1)], second[i]);
                Rw[(-(u + 1)) & (T - 1)] = q[-(j + u + 1)];
            }
            j += T; left -= T;
        }
    }
    if (!on_second) {
#pragma unroll
        for (int i = 0; i < T; ++i) { first[i] = second[i]; second[i] = 0.0; }
    }
}

// the products of the generic angular rate that take the balanced schedule
__host__ __device__ inline bool angrate_balanced(int m) { return conv_pairs_ok(4 * m + 1, 8); }
__host__ __device__ inline bool angrate_balanced2(int m) { return conv_pairs_ok(2 * m + 1, 4) && 2 * m + 1 <= 4 * 64; }

// generic angular rate: one wave per (row, vehicle); m = n + R
__global__ __launch_bounds__(kWave) void k_generic_angrate(const GenParams p)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const int n = p.n, nc = n + 1, m = n + p.R, mc = m + 1, L2 = 2 * m + 1, L4 = 4 * m + 1;
    const int mc8 = (mc + 7) & ~7, L28 = (L2 + 7) & ~7;
    // leading zeros of the padded b-operands: a whole operand length for the plain sliding-window products; the
    // balanced schedule of the squares never reaches further back than one tile
    const bool balanced = angrate_balanced(m), balanced2 = angrate_balanced2(m);
    const int padm = balanced2 ? 8 : mc8, padf = balanced ? 8 : L28;
    // b-operand buffers: the plain sliding window reads zeros up to the product's length, the balanced schedule
    // only within one tile of the data
    const int szm = padm + (balanced2 ? mc : L2) + 16, szf = padf + (balanced ? L2 : L4) + 16;
    const long gi = blockIdx.x;
    const int b = (int)(gi / p.n_veh), veh = (int)(gi - (long)b * p.n_veh);
    // LDS (the occupancy of this kernel is set by it: 18 KB per item at m = 110, two waves per SIMD):
    //   h2 | xp | yp | np_ | dp_ ; the elevation / derivative scratch lives inside np_/dp_, which are not
    //   needed before the degree-2m products have been read out of h2 / xp / yp.
    double* h2 = lds;              // [2][mc8] C(m,.) * d2, zero padded   (a-operands xDD, yDD)
    double* xp = h2 + 2 * mc8;     // [szm]  zero-padded C(m,.) * xD      (b-operand; xp + padm is the a-operand xD)
    double* yp = xp + szm;         // [szm]  zero-padded C(m,.) * yD
    double* np_ = yp + szm;        // [szf]  zero-padded C(2m,.) * num1   (np_ + padf is the a-operand of the square)
    double* dp_ = np_ + szf;       // [szf]  zero-padded C(2m,.) * den1
    double* pe = np_;              // [2][mc] elevated position           } scratch of the first stage,
    double* d1 = pe + 2 * mc;      // [2][mc] first derivative (plain)    } 8 mc <= 2 szf doubles
    double* d2 = d1 + 2 * mc;      // [2][mc] second derivative (plain)
    double* tm = d2 + 2 * mc;      // [2][mc] scratch
    const double* bn = p.bin + p.o_n;
    const double* bR = p.bin + p.o_R;
    const double* bm = p.bin + p.o_m;
    const double* b2m = p.bin + p.o_2m;
    const double* b4m = p.bin + p.o_4m;
    const double* v = p.Y + ((size_t)b * p.n_veh + veh) * 2 * nc;
    const double val = (double)m / p.tf[b];

    // zero everything that is read as padding (np_/dp_ are cleared after the first stage has used them)
    for (int e = lane; e < 2 * mc8 + 2 * szm; e += kWave) h2[e] = 0.0;
    // pos.elev(R)  (optimization.py:453)
    if (p.R == 0) {
        for (int e = lane; e < 2 * nc; e += kWave) pe[e] = v[e];
    } else {
        for (int e = lane; e < 2 * nc; e += kWave) tm[e] = v[e] * bn[e % nc];
        __syncthreads();
        for (int e = lane; e < 2 * mc; e += kWave) {
            const int q = e / mc, k = e - q * mc;
            pe[e] = conv_at(tm + q * nc, nc, bR, p.R + 1, k) / bm[k];
        }
    }
    __syncthreads();
    // two diff() passes, each derivative + elev(1)
    const double* srcs[2] = { pe, d1 };
    double* dsts[2] = { d1, d2 };
    for (int pass = 0; pass < 2; ++pass) {
        const double* s = srcs[pass];
        double* d = dsts[pass];
        for (int e = lane; e < 2 * mc; e += kWave) {
            const int q = e / mc, c = e - q * mc;
            tm[e] = (c < m) ? s[q * mc + c] * (-val) + s[q * mc + c + 1] * val : 0.0;
        }
        __syncthreads();
        for (int e = lane; e < 2 * mc; e += kWave) {
            const int q = e / mc, c = e - q * mc;
            double r;
            if (c == 0) r = tm[q * mc];
            else if (c == m) r = tm[q * mc + m - 1];
            else r = tm[q * mc + c - 1] * ((double)c / (double)m) + tm[q * mc + c] * ((double)(m - c) / (double)m);
            d[e] = r;
        }
        __syncthreads();
    }
    for (int e = lane; e < 2 * mc; e += kWave) {
        const int q = e / mc, c = e - q * mc;
        h2[q * mc8 + c] = d2[e] * bm[c];
        (q == 0 ? xp : yp)[padm + c] = d1[e] * bm[c];
    }
    __syncthreads();
    for (int e = lane; e < 2 * szf; e += kWave) np_[e] = 0.0;      // the scratch is dead: now the padded squares' operands
    __syncthreads();
    const double *xD = xp + padm, *yD = yp + padm, *xDD = h2, *yDD = h2 + mc8;
    const double *nu = np_ + padf, *de = dp_ + padf;
    // num1 = yDD*xD - xDD*yD, den1 = xD*xD + yD*yD  (degree 2m), 8 coefficients per lane
    // (4 per lane while that keeps more lanes busy: 2m+1 outputs are 28 lanes' worth of 8 at m = 110)
    auto stage2 = [&](auto tile_tag) {
        constexpr int T = decltype(tile_tag)::value;
        for (int k0 = T * lane; k0 < L2; k0 += T * kWave) {
            double t1[T], t2[T], e1[T], e2[T];
            conv_tile<T>(yDD, mc8, xp, padm, k0, t1);
            conv_tile<T>(xDD, mc8, yp, padm, k0, t2);
            conv_tile<T>(xD, mc8, xp, padm, k0, e1);
            conv_tile<T>(yD, mc8, yp, padm, k0, e2);
#pragma unroll
            for (int i = 0; i < T; ++i) {
                const int k = k0 + i;
                if (k < L2) {
                    const double c = b2m[k];
                    const double nk = (t1[i] / c - t2[i] / c) * c, dk = (e1[i] / c + e2[i] / c) * c;
                    np_[padf + k] = nk; dp_[padf + k] = dk;
                }
            }
        }
    };
    if (balanced2) {
        // two products side by side per pass: (yDD * xD | xDD * yD) -> num1, then (xD * xD | yD * yD) -> den1
        const int side = lane >> 5, slot = lane & 31;
        const int Hn = ((L2 + 3) / 4) / 2;
        for (int pass = 0; pass < 2; ++pass) {
            const double* ah = pass == 0 ? (side ? xDD : yDD) : (side ? yD : xD);
            const double* bq = (pass == 0 ? (side ? yp : xp) : (side ? yp : xp)) + padm;
            double fa[4], fb[4];
            int ta, tb;
            conv_pairs<4>(ah, bq, mc, mc8, L2, slot, ta, tb, fa, fb);
            double* dst = pass == 0 ? np_ : dp_;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const double oa = __shfl(fa[i], slot + 32), ob = __shfl(fb[i], slot + 32);
                if (side == 0 && slot < Hn) {
                    const int ka = 4 * ta + i, kb = 4 * tb + i;
                    // the sums are C(2m,k) times the Bernstein coefficients, which is the scaling the next
                    // product wants: no division by C(2m,k) and multiplication back
                    if (ka < L2) dst[padf + ka] = pass == 0 ? fa[i] - oa : fa[i] + oa;
                    if (kb < L2) dst[padf + kb] = pass == 0 ? fb[i] - ob : fb[i] + ob;
                }
            }
        }
    } else if (L2 <= 4 * kWave) stage2(std::integral_constant<int, 4>{});
    else stage2(std::integral_constant<int, 8>{});
    __syncthreads();
    double* o = p.out + ((size_t)b * p.n_veh + veh) * L4;
    if (balanced) {
        const int side = lane >> 5, slot = lane & 31;
        const int H = ((L4 + 7) / 8) / 2;
        double first[8], s8[8];
        int ta, tb;
        conv_pairs<8>(side ? de : nu, (side ? dp_ : np_) + padf, L2, L28, L4, slot, ta, tb, first, s8);
        // quotients: the num lane of a slot takes den's sums from lane + 32
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double da = __shfl(first[i], slot + 32), db = __shfl(s8[i], slot + 32);
            if (side == 0 && slot < H) {
                const int ka = 8 * ta + i, kb = 8 * tb + i;
                // num.cpts / den.cpts element-wise (optimization.py:608): the common factor C(4m,k) cancels
                if (ka < L4) o[ka] = p.offset - first[i] / da;
                if (kb < L4) o[kb] = p.offset - s8[i] / db;
            }
        }
        return;
    }
    for (int k0 = 8 * lane; k0 < L4; k0 += 8 * kWave) {
        double sn[8], sd[8];
        conv_tile8(nu, L28, np_, padf, k0, sn);
        conv_tile8(de, L28, dp_, padf, k0, sd);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = k0 + i;
            if (k < L4) o[k] = p.offset - (sn[i] / b4m[k]) / (sd[i] / b4m[k]);
        }
    }
}

// =====================================================================================
//  single-curve Bernstein algebra (Bezier object methods), one wave per row
// =====================================================================================
struct BernParams {
    const double* __restrict__ a;
    const double* __restrict__ b;
    const double* __restrict__ bin;
    double* __restrict__ out;
    int rows, m, n, R, d;
    int o_a, o_b, o_c;   // binomial rows
    double T;
};

__global__ __launch_bounds__(kWave) void k_bern_elev(const BernParams p)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x, r = blockIdx.x, nc = p.n + 1, LR = nc + p.R;
    const double* bn = p.bin + p.o_a;
    const double* bR = p.bin + p.o_b;
    const double* bo = p.bin + p.o_c;
    if (p.R == 0) {
        for (int k = lane; k < nc; k += kWave) p.out[(size_t)r * nc + k] = p.a[(size_t)r * nc + k];
        return;
    }
    for (int e = lane; e < nc; e += kWave) lds[e] = p.a[(size_t)r * nc + e] * bn[e];
    __syncthreads();
    for (int k = lane; k < LR; k += kWave) p.out[(size_t)r * LR + k] = conv_at(lds, nc, bR, p.R + 1, k) / bo[k];
}

__global__ __launch_bounds__(kWave) void k_bern_diff(const BernParams p)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x, r = blockIdx.x, n = p.n, nc = n + 1;
    const double val = (double)n / p.T;
    const double* s = p.a + (size_t)r * nc;
    for (int c = lane; c < n; c += kWave) lds[c] = s[c] * (-val) + s[c + 1] * val;
    __syncthreads();
    for (int c = lane; c < nc; c += kWave) {
        double d;
        if (c == 0) d = lds[0];
        else if (c == n) d = lds[n - 1];
        else d = lds[c - 1] * ((double)c / (double)n) + lds[c] * ((double)(n - c) / (double)n);
        p.out[(size_t)r * nc + c] = d;
    }
}

__global__ __launch_bounds__(kWave) void k_bern_mul(const BernParams p)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x, r = blockIdx.x, mc = p.m + 1, nc = p.n + 1, L = p.m + p.n + 1;
    double* ah = lds;
    double* bh = lds + mc;
    const double* bm = p.bin + p.o_a;
    const double* bn = p.bin + p.o_b;
    const double* bo = p.bin + p.o_c;
    for (int e = lane; e < mc; e += kWave) ah[e] = p.a[(size_t)r * mc + e] * bm[e];
    for (int e = lane; e < nc; e += kWave) bh[e] = p.b[(size_t)r * nc + e] * bn[e];
    __syncthreads();
    for (int k = lane; k < L; k += kWave) p.out[(size_t)r * L + k] = conv_at(ah, mc, bh, nc, k) / bo[k];
}

__global__ __launch_bounds__(kWave) void k_bern_normsq(const BernParams p)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x, nc = p.n + 1, L = 2 * p.n + 1;
    const double* bn = p.bin + p.o_a;
    const double* bo = p.bin + p.o_c;
    for (int e = lane; e < p.d * nc; e += kWave) lds[e] = p.a[e] * bn[e % nc];
    __syncthreads();
    for (int k = lane; k < L; k += kWave) {
        double s = 0.0;
        for (int q = 0; q < p.d; ++q) s += conv_at(lds + q * nc, nc, lds + q * nc, nc, k);
        p.out[k] = (0.5 * p.d) * s / bo[k];
    }
}

// Bezier.split (bezier.py:533-572) -> deCasteljauSplit (bezier.py:985-1027): the de Casteljau triangle at
// z = (tDiv - t0) / (tf - t0), one wave per row, the row in LDS.  Level by level every lane forms
// (1-z) c_i + z c_{i+1} from the previous level; left[k] is the first element of level k, right[k] (already in
// the curve's own orientation, i.e. the reference's `right[::-1]`) the last element of level n-k.
__global__ __launch_bounds__(kWave) void k_bern_split(const BernParams p, double* __restrict__ right)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x, r = blockIdx.x, n = p.n, nc = n + 1;
    double* cur = lds;
    double* nxt = lds + nc;
    const double z = p.T, w = 1.0 - z;
    for (int e = lane; e < nc; e += kWave) cur[e] = p.a[(size_t)r * nc + e];
    __syncthreads();
    double* L = p.out + (size_t)r * nc;
    double* Rt = right + (size_t)r * nc;
    for (int lev = 0; lev < n; ++lev) {
        const int len = nc - lev;
        if (lane == 0) { L[lev] = cur[0]; Rt[n - lev] = cur[len - 1]; }
        for (int i = lane; i < len - 1; i += kWave) nxt[i] = w * cur[i] + z * cur[i + 1];
        __syncthreads();
        double* t = cur; cur = nxt; nxt = t;
    }
    if (lane == 0) { L[n] = cur[0]; Rt[0] = cur[0]; }
}

// Bezier.__call__ / Bezier.curve (bezier.py:184-199, 233-258) -> deCasteljauCurve (bezier.py:945-982): every row of control
// points at every value of tau.  T = (tau - t0) / (tf - t0), then the triangle `(1 - t) c_i + t c_{i+1}` per sample: one
// lane per (row, sample), the row's control points in LDS (all lanes read the same address: a broadcast), the lane's
// working copy in LDS as well (pitch nc | 1: conflict-free).
__global__ __launch_bounds__(kWave) void k_bern_eval(const double* __restrict__ cpts, const double* __restrict__ tau, int n_tau,
                                                     int nc, double t0, double tf, double* __restrict__ out)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x, r = blockIdx.y, k = blockIdx.x * kWave + lane, pitch = nc | 1;
    double* row = lds;                         // [nc]
    double* w = lds + nc + lane * pitch;       // [nc] per lane
    for (int e = lane; e < nc; e += kWave) row[e] = cpts[(size_t)r * nc + e];
    __syncthreads();
    if (k >= n_tau) return;
    const double t = (tau[k] - t0) / (tf - t0), u = 1.0 - t;
    for (int i = 0; i < nc; ++i) w[i] = row[i];
    for (int len = nc; len > 1; --len)
        for (int i = 0; i < len - 1; ++i) w[i] = u * w[i] + t * w[i + 1];
    out[(size_t)r * n_tau + k] = w[0];
}

// =====================================================================================
//  finite-difference batch, objectives
// =====================================================================================
__global__ void k_fd_batch(const double* __restrict__ Y0, double* __restrict__ Y, int rows, int nc,
                           int fixed, double h, int B, int row0)
{
    const size_t ysz = (size_t)rows * nc;
    const size_t total = ysz * B;
    const int free_cols = nc - 2 * fixed;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (size_t)gridDim.x * blockDim.x) {
        const int bl = (int)(e / ysz), b = bl + row0;          // local row, row of the batch
        const int r = (int)(e - (size_t)bl * ysz);
        double v = Y0[r];
        if (b > 0) {
            const int k = b - 1, pr = k / free_cols, pc = fixed + (k - pr * free_cols);
            if (r == pr * nc + pc) v += h;
        }
        Y[e] = v;
    }
}

// optimization.py:462-489: sum over vehicles and segments of |P_{i+1} - P_i| (unused third
// slot of the reference's np.empty(3) scratch taken as 0 for dim == 2)
__global__ __launch_bounds__(kWave) void k_euclid(const double* __restrict__ Y, double* __restrict__ out,
                                                  int n_veh, int dim, int nc)
{
    const int lane = threadIdx.x, b = blockIdx.x, n = nc - 1;
    const double* Yr = Y + (size_t)b * n_veh * dim * nc;
    double s = 0.0;
    for (int e = lane; e < n_veh * n; e += kWave) {
        const int v = e / n, i = e - v * n;
        double q = 0.0;
        for (int j = 0; j < dim; ++j) {
            const double t = Yr[(size_t)(v * dim + j) * nc + i + 1] - Yr[(size_t)(v * dim + j) * nc + i];
            q = fma(t, t, q);
        }
        s += sqrt(q);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if (lane == 0) out[b] = s;
}

__global__ __launch_bounds__(kWave) void k_rowsum(const double* __restrict__ in, double* __restrict__ out, int len)
{
    const int lane = threadIdx.x, b = blockIdx.x;
    double s = 0.0;
    for (int e = lane; e < len; e += kWave) s += in[(size_t)b * len + e];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if (lane == 0) out[b] = s;
}

// =====================================================================================
//  launchers
// =====================================================================================
template <int NC, int DIM, int MODE, bool MINONLY>
static int launch_ns_t(obtg_ctx* c, NsParams p, int B, int kernel_id)
{
    using S = NsShape<NC, DIM>;
    // the R > 0 path always transposes 64 rows x kTileK columns
    const size_t budget = kNsLdsBudget;
    const size_t stage = (size_t)p.stage_slots * S::VP * sizeof(double);
    size_t lds = 0;
    if (MINONLY) { p.tile_rows = 0; lds = stage; }
    else if (p.R > 0) { p.tile_rows = kWave; lds = ((stage + 15) & ~(size_t)15) + (size_t)p.waves * elev_mfma_wave_doubles(S::L, p.R) * sizeof(double); }
    else {
        for (int tr = kWave; tr >= 16; tr >>= 1) {
            p.tile_rows = tr;
            lds = stage + (size_t)p.waves * tr * S::TPF * sizeof(double);
            if (lds <= budget) break;
        }
    }
    void (*kern)(const NsParams) = p.R > 0 ? k_normsq_elev<NC, DIM, MODE, MINONLY, true>
                                           : k_normsq_elev<NC, DIM, MODE, MINONLY, false>;
    if (MODE == 0 && !MINONLY) {
        if (const size_t lc = sep_elev_coop_lds<NC, DIM>(p)) { kern = k_sep_elev_coop<NC, DIM>; lds = lc; p.waves = 4; }
    }
    if (lds > 160 * 1024) return OBTG_ERR_UNSUPPORTED;
    if (lds > 48 * 1024)
        OBTG_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((unsigned)((size_t)B * p.wgs_per_row));
    ScopedKernelTimer t(c, kernel_id);
    hipLaunchKernelGGL(kern, grid, dim3(kWave * p.waves), lds, c->stream, p);
    OBTG_HIP(c, hipGetLastError());
    return OBTG_OK;
}

template <int MODE, bool MINONLY>
static int dispatch_ns(obtg_ctx* c, const NsParams& p, int B, int kernel_id)
{
    const int nc = c->deg + 1;
#define OBTG_CASE(NC_, D_) \
    if (nc == NC_ && c->dim == D_) return launch_ns_t<NC_, D_, MODE, MINONLY>(c, p, B, kernel_id);
#define OBTG_CASE_D(NC_) OBTG_CASE(NC_, 2) OBTG_CASE(NC_, 3)
    OBTG_NC_SEP(OBTG_CASE_D)
#undef OBTG_CASE_D
#undef OBTG_CASE
    return OBTG_ERR_UNSUPPORTED;
}

static bool fast_shape(const obtg_ctx* c)
{
    const int nc = c->deg + 1;
    return nc_in_sep(nc) && (c->dim == 2 || c->dim == 3) && c->R <= 512;
}

static int gen_common(obtg_ctx* c, GenParams& g)
{
    // make every binomial row resident BEFORE taking the base pointer
    const int n = c->deg, R = c->R, m = n + R;
    int need[] = { n, 2 * n, R, 2 * n + R, m, 2 * m, 4 * m };
    for (int v : need) { int o = binrow_offset(c, v); if (o < 0) return o; }
    g.o_n = binrow_offset(c, n); g.o_2n = binrow_offset(c, 2 * n); g.o_R = binrow_offset(c, R);
    g.o_2nR = binrow_offset(c, 2 * n + R);
    g.o_m = binrow_offset(c, m); g.o_2m = binrow_offset(c, 2 * m); g.o_4m = binrow_offset(c, 4 * m);
    g.o_1m = 0;
    g.bin = c->d_binrows.as<double>();
    g.n_veh = c->n_veh; g.n_obj = c->n_obj; g.dim = c->dim; g.n = n; g.R = R;
    g.obs = c->d_obs.as<double>();
    g.pairs = c->d_pairs.as<int2>();
    return OBTG_OK;
}

// Parameters of the fast-path temporal-separation launch (everything but the output transform);
// OBTG_ERR_UNSUPPORTED when the shape has no specialised kernel.
int plan_temporal_sep(obtg_ctx* c, const double* dY, int B, int pair_begin, int pair_count, double* d_out, NsParams& p)
{
    if (!fast_shape(c)) return OBTG_ERR_UNSUPPORTED;
    p = NsParams{};
    p.Y = dY; p.obs = c->d_obs.as<double>(); p.tf = nullptr;
    p.pairs = c->d_pairs.as<int2>(); p.W2 = c->d_w2.as<double>(); p.Tt = c->d_Tt.as<double>(); p.Td = c->d_Td.as<double>();
    p.Tf = c->d_Tf.as<double>();
    p.out = d_out; p.n_veh = c->n_veh; p.n_obj = c->n_obj; p.R = c->R;
    p.item_begin = pair_begin; p.item_count = pair_count;
    // workgroup = 4 waves sharing one staging of the row's objects; each wave walks
    // groups of 64 pairs.  Keep >= ~4k workgroups so that 256 CUs x 4 resident stay fed.
    const int vlen = c->dim * (c->deg + 1), vp = (vlen % 2 == 0) ? vlen + 1 : vlen;
    const size_t row_bytes = sizeof(double) * (size_t)c->n_obj * vp;
    int groups_total = (pair_count + kWave - 1) / kWave;
    if (row_bytes <= 24 * 1024 || c->n_obj <= 2 * kWave) {
        p.waves = groups_total >= 4 ? 4 : groups_total;
        int gpw = 16;   // groups per workgroup
        if (const char* e = getenv("OBTG_SEP_GPW")) gpw = std::max(4, std::min(64, atoi(e)));      // (experiments)
        while (gpw > p.waves && (long)B * ((groups_total + gpw - 1) / gpw) < 4096) gpw >>= 1;
        if (gpw < p.waves) gpw = p.waves;
        p.groups_per_wg = gpw;
        p.wgs_per_row = (groups_total + gpw - 1) / gpw;
        // LDS slots: replay the kernel's staging rule over this launch's chunks
        const int chunk = kWave * gpw;
        int slots = 0;
        for (int it0 = pair_begin; it0 < pair_begin + pair_count; it0 += chunk) {
            const int last = std::min(pair_begin + pair_count, it0 + chunk) - 1;
            const int fx = c->h_pairs[2 * it0], fy = c->h_pairs[2 * it0 + 1];
            const int lx = c->h_pairs[2 * last], ly = c->h_pairs[2 * last + 1];
            const int nI = lx - fx + 1;
            const int nA = (nI == 1) ? (ly - fy + 1) : (c->n_obj - fy);
            const int nB = (nI == 1) ? 0 : std::max(0, ((nI >= 3) ? c->n_obj - 1 : ly) - (fx + 2) + 1);
            slots = std::max(slots, nI + nA + nB);
        }
        p.stage_all = c->n_obj <= slots ? 1 : 0;
        p.stage_slots = p.stage_all ? c->n_obj : slots;
        p.tiling = 0; p.tiles = nullptr;
    } else {
        // large swarm: row-window tiles (kTileRows rows x 64 columns of the pair triangle; a wave takes
        // rows wave, wave + 4, ...: twice the rows per staging of the 64-column window)
        constexpr int kTileRows = 8;       // 4 rows: 15.3 ms at C4, 8 rows: 14.4 ms, 16 rows: 15.3 ms
        p.waves = 4; p.groups_per_wg = kTileRows; p.stage_all = 0; p.tiling = 1;
        p.stage_slots = kTileRows + kWave;
        if (c->tiles_begin != pair_begin || c->tiles_count != pair_count || c->h_tiles.empty()) {
            c->h_tiles.clear();
            const long nobj = c->n_obj, pend = (long)pair_begin + pair_count;
            for (int i0 = 0; i0 < c->n_obj - 1; i0 += kTileRows) {
                const int i1 = std::min(c->n_obj - 2, i0 + kTileRows - 1);
                // pair-index span of rows i0..i1
                const long lo = (long)i0 * nobj - (long)i0 * (i0 + 1) / 2;
                const long hi = (long)i1 * nobj - (long)i1 * (i1 + 1) / 2 + (nobj - 1 - i1 - 1);
                if (hi < pair_begin || lo >= pend) continue;
                for (int j0 = ((i0 + 1) / kWave) * kWave; j0 < c->n_obj; j0 += kWave) {
                    c->h_tiles.push_back(i0);
                    c->h_tiles.push_back(j0);
                }
            }
            int rc2 = c->d_tiles.reserve(sizeof(int) * c->h_tiles.size());
            if (rc2) return rc2;
            OBTG_HIP(c, hipMemcpyAsync(c->d_tiles.p, c->h_tiles.data(), sizeof(int) * c->h_tiles.size(),
                                       hipMemcpyHostToDevice, c->stream));
            OBTG_HIP(c, hipStreamSynchronize(c->stream));
            c->tiles_begin = pair_begin; c->tiles_count = pair_count;
        }
        p.tiles = c->d_tiles.as<int2>();
        p.wgs_per_row = (int)(c->h_tiles.size() / 2);
    }
    return OBTG_OK;
}

// LDS bytes of a full-output R == 0 launch planned above: the largest transposition tile (64, 32 or 16
// rows per wave, sets p.tile_rows) that fits `budget`
size_t temporal_sep_lds_bytes(const obtg_ctx* c, NsParams& p, size_t budget)
{
    const int vlen = c->dim * (c->deg + 1), vp = (vlen % 2 == 0) ? vlen + 1 : vlen;
    const int L = 2 * c->deg + 1, tpf = (L % 2 == 0) ? L + 1 : L;
    const size_t stage = (size_t)p.stage_slots * vp * sizeof(double);
    for (int tr = kWave; tr >= 16; tr >>= 1) {
        p.tile_rows = tr;
        const size_t lds = stage + (size_t)p.waves * tr * tpf * sizeof(double);
        if (lds <= budget || tr == 16) return lds;
    }
    return 0;
}

// items x L values -> per item the k smallest, ascending, with their positions (the any-degree shapes' route to
// obtg_temporal_sep_active: their kernels write whole rows; the specialised kernels select in their own epilogue)
__global__ __launch_bounds__(256) void k_select_smallest(const double* __restrict__ rows, long items, int L, int k,
                                                         double* __restrict__ out, int* __restrict__ idx)
{
    const long it = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (it >= items) return;
    const double* r = rows + it * L;
    Smallest4 sm;
    for (int j = 0; j < L; ++j) sm.put(r[j], j);
    sm.store(out, idx, (size_t)it * k, k);
}

int launch_temporal_sep(obtg_ctx* c, const double* dY, int B, double max_sep, int pair_begin,
                        int pair_count, bool min_only, double* d_out, int sel_k, int* d_sel_idx)
{
    if (B <= 0 || pair_count <= 0) return OBTG_OK;
    if (sel_k < 0 || sel_k > 4) return OBTG_ERR_ARG;
    if (sel_k > 0) min_only = true;           // the selection lives in the reduced kernels' epilogue
    int rc = ensure_tables(c);
    if (rc) return rc;
    NsParams p{};
    rc = plan_temporal_sep(c, dY, B, pair_begin, pair_count, d_out, p);
    if (rc != OBTG_OK && rc != OBTG_ERR_UNSUPPORTED) return rc;
    if (rc == OBTG_OK) {
        p.sign = 1.0; p.offset = 0.0 - square_as_python(max_sep);
        p.sel_k = sel_k; p.sel_idx = d_sel_idx;
        if (c->fd.Y0) { p.Y = c->fd.Y0; p.fd = 1 + c->fd.row0; p.fd_fixed = c->fd.fixed; p.fd_h = c->fd.h; }
        rc = min_only ? dispatch_ns<0, true>(c, p, B, OBTG_K_TEMPORAL_SEP)
                      : dispatch_ns<0, false>(c, p, B, OBTG_K_TEMPORAL_SEP);
        if (rc != OBTG_ERR_UNSUPPORTED) return rc;
    }
    if (c->fd.Y0) return kNeedBatch;          // the generic kernel reads its rows from memory
    if (sel_k > 0) {
        // any-degree shapes: whole rows into a workspace, then the selection as a launch of its own
        const int L = 2 * c->deg + c->R + 1;
        const long items = (long)B * pair_count;
        DevBuf& ws = c->ws_misc[7];
        if ((rc = ws.reserve(sizeof(double) * (size_t)items * L))) return rc;
        if ((rc = launch_temporal_sep(c, dY, B, max_sep, pair_begin, pair_count, false, ws.as<double>(), 0, nullptr))) return rc;
        hipLaunchKernelGGL(k_select_smallest, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, c->stream,
                           ws.as<double>(), items, L, sel_k, d_out, d_sel_idx);
        OBTG_HIP(c, hipGetLastError());
        return OBTG_OK;
    }
    GenParams g{};
    rc = gen_common(c, g);
    if (rc) return rc;
    if (2 * c->deg + c->R + 1 > kMaxGenericLen) return OBTG_ERR_UNSUPPORTED;
    g.Y = dY; g.out = d_out; g.item_begin = pair_begin; g.item_count = pair_count; g.B = B;
    g.sign = 1.0; g.offset = 0.0 - square_as_python(max_sep); g.min_only = min_only ? 1 : 0;
    const int nc = c->deg + 1;
    size_t lds = sizeof(double) * ((size_t)2 * c->dim * nc + 2 * c->deg + 1);
    ScopedKernelTimer t(c, OBTG_K_TEMPORAL_SEP);
    hipLaunchKernelGGL(k_generic_normsq_elev<0>, dim3((unsigned)((size_t)B * pair_count)), dim3(kWave), lds,
                       c->stream, g);
    OBTG_HIP(c, hipGetLastError());
    return OBTG_OK;
}

int launch_speed(obtg_ctx* c, const double* dY, const double* d_tf, int B, double bound, int is_max,
                 double* d_out)
{
    if (B <= 0) return OBTG_OK;
    int rc = ensure_tables(c);
    if (rc) return rc;
    const double b2 = square_as_python(bound);
    if (fast_shape(c)) {
        NsParams p{};
        p.Y = dY; p.obs = nullptr; p.tf = d_tf; p.pairs = nullptr;
        p.W2 = c->d_w2.as<double>(); p.Tt = c->d_Tt.as<double>(); p.Td = c->d_Td.as<double>(); p.out = d_out;
        p.Tf = c->d_Tf.as<double>();
        p.n_veh = c->n_veh; p.n_obj = c->n_veh; p.R = c->R;
        p.item_begin = 0; p.item_count = c->n_veh;
        p.groups_per_wg = 1;
        p.waves = 1;
        p.wgs_per_row = (c->n_veh + kWave - 1) / kWave;
        p.stage_slots = std::min(c->n_veh, kWave);
        p.stage_all = 0; p.tiling = 0; p.tiles = nullptr;
        p.sign = is_max ? -1.0 : 1.0; p.offset = is_max ? b2 : -b2;
        if (c->fd.Y0) { p.Y = c->fd.Y0; p.fd = 1 + c->fd.row0; p.fd_fixed = c->fd.fixed; p.fd_h = c->fd.h; }
        rc = dispatch_ns<1, false>(c, p, B, OBTG_K_SPEED);
        if (rc != OBTG_ERR_UNSUPPORTED) return rc;
    }
    if (c->fd.Y0) return kNeedBatch;
    GenParams g{};
    rc = gen_common(c, g);
    if (rc) return rc;
    if (2 * c->deg + c->R + 1 > kMaxGenericLen) return OBTG_ERR_UNSUPPORTED;
    g.Y = dY; g.tf = d_tf; g.out = d_out; g.item_begin = 0; g.item_count = c->n_veh; g.B = B;
    g.sign = is_max ? -1.0 : 1.0; g.offset = is_max ? b2 : -b2; g.min_only = 0;
    const int nc = c->deg + 1;
    size_t lds = sizeof(double) * ((size_t)2 * c->dim * nc + 2 * c->deg + 1);
    ScopedKernelTimer t(c, OBTG_K_SPEED);
    hipLaunchKernelGGL(k_generic_normsq_elev<1>, dim3((unsigned)((size_t)B * c->n_veh)), dim3(kWave), lds,
                       c->stream, g);
    OBTG_HIP(c, hipGetLastError());
    return OBTG_OK;
}

template <int NC>
static int launch_dyn_t(obtg_ctx* c, const AngParams& p, int kernel_id)
{
    constexpr int L4 = 4 * (NC - 1) + 1, L2 = 2 * (NC - 1) + 1;
    size_t lds = sizeof(double) * kWave * L4;
    const unsigned groups = (unsigned)((p.total + kWave - 1) / kWave);
    ScopedKernelTimer t(c, kernel_id);
    if (p.out) {   // angular rate (with or without the speed rows): two waves per group
        hipLaunchKernelGGL(k_dynamics2<NC>, dim3(groups), dim3(2 * kWave), lds + sizeof(double) * kWave * L2, c->stream, p);
        OBTG_HIP(c, hipGetLastError());
        return OBTG_OK;
    }
    hipLaunchKernelGGL(k_dynamics<NC>, dim3(groups * kDynParts), dim3(kWave), lds, c->stream, p);
    OBTG_HIP(c, hipGetLastError());
    return OBTG_OK;
}

// obtg_ctx_set_ang_rate_order(2): the list the dynamics groups fill (zeroed here) ...
static int ang_exact_prepare(obtg_ctx* c, int total_items, int** d_flags)
{
    *d_flags = nullptr;
    if (!c->ang_exact) return OBTG_OK;
    if (c->ang_dd_R != c->R || c->d_ang_dd.p == nullptr) {
        std::vector<double> t;
        c->ang_dd_off = angrate_dd_tables(c->deg, c->R, t);
        int rc = c->d_ang_dd.reserve(t.size() * sizeof(double));
        if (rc) return rc;
        OBTG_HIP(c, hipMemcpyAsync(c->d_ang_dd.p, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
        OBTG_HIP(c, hipStreamSynchronize(c->stream));
        c->ang_dd_R = c->R;
    }
    int rc = c->d_ang_flags.reserve(sizeof(int) * ((size_t)total_items + 1));
    if (rc) return rc;
    OBTG_HIP(c, hipMemsetAsync(c->d_ang_flags.p, 0, sizeof(int), c->stream));
    *d_flags = c->d_ang_flags.as<int>();
    return OBTG_OK;
}

// ... and the double-double pass over it, behind the launch that filled it
template <int NC>
static int ang_exact_finish(obtg_ctx* c, const AngParams& a, const int* d_flags, int total_items)
{
    if (!d_flags) return OBTG_OK;
    constexpr int L2 = 2 * (NC - 1) + 1, L4 = 4 * (NC - 1) + 1;
    AngDdParams q{};
    q.a = a; q.flags = d_flags; q.tab = c->d_ang_dd.as<double>(); q.off = c->ang_dd_off; q.R = c->R; q.cap = total_items;
    const size_t lds = sizeof(double) * 2 * (7 * NC + 2 * L2 + 2 * L4);
    ScopedKernelTimer t(c, OBTG_K_ANG_RATE);
    hipLaunchKernelGGL(k_angrate_dd<NC>, dim3((unsigned)std::max(1, std::min(total_items, 512))), dim3(kWave), lds, c->stream, q);
    OBTG_HIP(c, hipGetLastError());
    return OBTG_OK;
}

template <int NC>
static int launch_dyn_elev_t(obtg_ctx* c, AngElevParams q, int kernel_id)
{
    const size_t lds = sizeof(double) * dyn_elev_lds_doubles(NC - 1, q.R);
    const unsigned groups = (unsigned)((q.a.total + kWave - 1) / kWave);
    if (lds > 48 * 1024)
        OBTG_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_dynamics_elev<NC>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int rc = ang_exact_prepare(c, q.a.total, &q.flags);
    if (rc) return rc;
    {
        ScopedKernelTimer t(c, kernel_id);
        hipLaunchKernelGGL(k_dynamics_elev<NC>, dim3(groups), dim3(4 * kWave), lds, c->stream, q);
        OBTG_HIP(c, hipGetLastError());
    }
    return ang_exact_finish<NC>(c, q.a, q.flags, q.a.total);
}

// DEG_ELEV > 0, angular rate wanted: products at degree 4n, then elevation by 4R (k_dynamics_elev).
// obtg_ctx_set_ang_rate_order(ctx, 1) keeps the reference's order of operations (generic kernel) instead.
static bool dyn_fast_elev(const obtg_ctx* c)
{
    const int nc = c->deg + 1;
    return c->dim == 2 && c->R > 0 && !c->ang_elevate_first && c->d_ang_T4.p != nullptr && c->d_ang_cv2.p != nullptr && nc_in_dyn(nc);
}

static bool dyn_fast(const obtg_ctx* c)
{
    const int nc = c->deg + 1;
    return c->dim == 2 && c->R == 0 && nc_in_dyn(nc);
}

// Which order of operations the angular rate of this context's shape REALLY runs in (obtg_ctx_ang_rate_order_in_effect):
// the request of obtg_ctx_set_ang_rate_order holds only where the kernels it names exist.
int ang_rate_order_in_effect(obtg_ctx* c)
{
    if (c->dim != 2) return OBTG_ERR_ARG;
    if (c->R == 0) return 0;                       // one order of operations: nothing to choose
    if (c->ang_elevate_first) return 1;
    int rc = ensure_tables(c);
    if (rc) return rc;
    if (!dyn_fast_elev(c)) return 1;               // no products-then-elevation kernel for this (deg, R): the any-degree kernel elevates first
    return c->ang_exact ? 2 : 0;
}

// shapes whose dynamics kernels form a virtual finite-difference batch on the fly (see obtg_ctx::fd)
bool dynamics_fd_on_the_fly(const obtg_ctx* c, bool want_ang) { return dyn_fast(c) || (want_ang && dyn_fast_elev(c)); }
bool bernstein_fd_on_the_fly(const obtg_ctx* c) { return fast_shape(c); }

// the other speed bound's rows of the same pass (obtg_ctx_set_second_speed_bound)
static void second_speed_rows(const obtg_ctx* c, AngParams& p)
{
    if (!c->speed2.d_out || !p.out_speed) return;
    const double b2 = square_as_python(c->speed2.bound);
    p.out_speed2 = c->speed2.d_out;
    p.sp2_sign = c->speed2.is_max ? -1.0 : 1.0; p.sp2_offset = c->speed2.is_max ? b2 : -b2;
}

// speed and/or angular rate in one launch (either output may be null)
int launch_temporal_sep_fd(obtg_ctx* c, const double* dY0, int n_pert, const int* d_prow, const int* d_pcol,
                           const double* d_pval, double max_sep, double* d_out, int min_only, int fd_row0, int fd_fixed, double fd_h)
{
    if (n_pert <= 0 || c->n_obj < 2) return OBTG_OK;
    int rc = ensure_tables(c);
    if (rc) return rc;
    if (!fast_shape(c)) return OBTG_ERR_UNSUPPORTED;
    TsepFdParams p{};
    p.Y0 = dY0; p.obs = c->d_obs.as<double>(); p.W2 = c->d_w2.as<double>(); p.Td = c->d_Td.as<double>();
    p.prow = d_prow; p.pcol = d_pcol; p.pval = d_pval; p.out = d_out;
    p.min_only = min_only; p.fd_row0 = fd_row0; p.fd_fixed = fd_fixed; p.fd_h = fd_h;
    p.n_veh = c->n_veh; p.n_obj = c->n_obj; p.R = c->R; p.n_pert = n_pert;
    p.sign = 1.0; p.offset = 0.0 - square_as_python(max_sep);
    const long items = (long)n_pert * (c->n_obj - 1);
    const dim3 grid((unsigned)((items + kWave - 1) / kWave));
    const int nc = c->deg + 1;
    void (*kern)(const TsepFdParams) = nullptr;
#define OBTG_CASE(NC_, D_) if (nc == NC_ && c->dim == D_) kern = k_tsep_fd<NC_, D_>;
#define OBTG_CASE_D(NC_) OBTG_CASE(NC_, 2) OBTG_CASE(NC_, 3)
    OBTG_NC_SEP(OBTG_CASE_D)
#undef OBTG_CASE_D
#undef OBTG_CASE
    if (!kern) return OBTG_ERR_UNSUPPORTED;
    ScopedKernelTimer t(c, OBTG_K_TEMPORAL_SEP);
    hipLaunchKernelGGL(kern, grid, dim3(kWave), 0, c->stream, p);
    OBTG_HIP(c, hipGetLastError());
    return OBTG_OK;
}

int launch_one_vs_many_min(obtg_ctx* c, const double* d_one, int B, const double* d_many, int K, double max_sep, double* d_out)
{
    if (B <= 0 || K <= 0) return OBTG_OK;
    int rc = ensure_tables(c);
    if (rc) return rc;
    if (!fast_shape(c)) return OBTG_ERR_UNSUPPORTED;
    OneManyParams p{};
    p.one = d_one; p.many = d_many; p.W2 = c->d_w2.as<double>(); p.Td = c->d_Td.as<double>(); p.out = d_out;
    p.B = B; p.K = K; p.R = c->R;
    p.sign = 1.0; p.offset = 0.0 - square_as_python(max_sep);
    const long items = (long)B * K;
    const int nc = c->deg + 1;
    void (*kern)(const OneManyParams) = nullptr;
#define OBTG_CASE(NC_, D_) if (nc == NC_ && c->dim == D_) kern = k_one_vs_many<NC_, D_>;
#define OBTG_CASE_D(NC_) OBTG_CASE(NC_, 2) OBTG_CASE(NC_, 3)
    OBTG_NC_SEP(OBTG_CASE_D)
#undef OBTG_CASE_D
#undef OBTG_CASE
    if (!kern) return OBTG_ERR_UNSUPPORTED;
    // small problems: 64-lane workgroups spread a few hundred items over more CUs
    const int threads = items >= 16384 ? 256 : 64;
    ScopedKernelTimer t(c, OBTG_K_TEMPORAL_SEP);
    hipLaunchKernelGGL(kern, dim3((unsigned)((items + threads - 1) / threads)), dim3(threads), 0, c->stream, p);
    OBTG_HIP(c, hipGetLastError());
    return OBTG_OK;
}

template <int NC>
static int launch_sep_dyn_elev_t(obtg_ctx* c, SepDynElevParams& sp, int B)
{
    sp.ts.tile_rows = kWave;
    const size_t lds_t = sep_elev_coop_lds<NC, 2>(sp.ts);      // the separation workgroups are sep_elev_coop_body's
    if (!lds_t) return OBTG_ERR_UNSUPPORTED;                   // (rows that do not stage whole, 2n + R + 1 > 128: separate launches)
    const size_t lds = std::max(lds_t, sizeof(double) * dyn_elev_lds_doubles(NC - 1, sp.dyn.R));
    if (lds > 80 * 1024 - 512) return OBTG_ERR_UNSUPPORTED;            // at least two workgroups per CU (three at C5: 53 KB)
    sp.n_kind[0] = B * sp.ts.wgs_per_row;
    sp.n_kind[1] = (sp.dyn.a.total + kWave - 1) / kWave;
    // shares of every 16 block ids by count: both kinds run out at about the same block id
    const double tot = (double)sp.n_kind[0] + sp.n_kind[1];
    sp.per16[1] = std::min(15, std::max(1, (int)(16.0 * sp.n_kind[1] / tot + 0.5)));
    if (const char* e = getenv("OBTG_SEP_DYN_SHARE")) sp.per16[1] = std::min(15, std::max(1, atoi(e)));      // (experiments)
    sp.per16[0] = 16 - sp.per16[1];
    int seen[2] = { 0, 0 };
    for (int i = 0; i < 16; ++i) {         // kind 1 in the slots where its running share crosses an integer
        const int k = ((i + 1) * sp.per16[1]) / 16 > (i * sp.per16[1]) / 16 ? 1 : 0;
        sp.pat[i] = (unsigned char)k; sp.rank[i] = (unsigned char)seen[k]++;
    }
    const int groups = std::max((sp.n_kind[0] + sp.per16[0] - 1) / sp.per16[0], (sp.n_kind[1] + sp.per16[1] - 1) / sp.per16[1]);
    if (lds > 48 * 1024)
        OBTG_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_sep_dynamics_elev<NC>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int rc = ang_exact_prepare(c, sp.dyn.a.total, &sp.dyn.flags);
    if (rc) return rc;
    {
        ScopedKernelTimer t(c, OBTG_K_TEMPORAL_SEP);
        hipLaunchKernelGGL(k_sep_dynamics_elev<NC>, dim3((unsigned)groups * 16u), dim3(4 * kWave), lds, c->stream, sp);
        OBTG_HIP(c, hipGetLastError());
    }
    return ang_exact_finish<NC>(c, sp.dyn.a, sp.dyn.flags, sp.dyn.a.total);
}

// DEG_ELEV > 0, planar: the separation rows and the speed / angular-rate rows of a batch in one launch
// (k_sep_dynamics_elev).  OBTG_ERR_UNSUPPORTED: shape outside it, nothing launched.
int launch_sep_dynamics_elev(obtg_ctx* c, const double* dY, int B, double max_sep, double* d_out_sep, const SweepFold& f)
{
    static const bool on = !(getenv("OBTG_SEP_DYN_ELEV") && getenv("OBTG_SEP_DYN_ELEV")[0] == '0');
    const int nc = c->deg + 1;
    if (!on || B <= 0 || c->R <= 0 || c->dim != 2 || !f.d_out_ang || !f.d_tf || !d_out_sep || c->n_pairs <= 0 ||
        !nc_in_elev(nc)) return OBTG_ERR_UNSUPPORTED;
    int rc = ensure_tables(c);
    if (rc) return rc;
    if (!dyn_fast_elev(c) || !fast_shape(c)) return OBTG_ERR_UNSUPPORTED;
    SepDynElevParams sp{};
    rc = plan_temporal_sep(c, dY, B, 0, c->n_pairs, d_out_sep, sp.ts);
    if (rc) return rc;
    if (sp.ts.waves != 4) return OBTG_ERR_UNSUPPORTED;                 // (fewer than four 64-pair groups per row)
    sp.ts.sign = 1.0; sp.ts.offset = 0.0 - square_as_python(max_sep);
    AngParams& p = sp.dyn.a;
    p.Y = dY; p.tf = f.d_tf; p.out = f.d_out_ang; p.out_speed = f.d_out_speed;
    p.n_veh = c->n_veh; p.total = B * c->n_veh;
    p.w2 = square_as_python(f.max_rate);
    const double b2 = square_as_python(f.speed_bound);
    p.sp_sign = f.speed_is_max ? -1.0 : 1.0; p.sp_offset = f.speed_is_max ? b2 : -b2;
    p.W2n = c->d_ang_w2n.as<double>(); p.W22n = c->d_ang_w22n.as<double>(); p.Wn = c->d_ang_wn.as<double>();
    second_speed_rows(c, p);
    if (c->fd.Y0) {
        sp.ts.Y = c->fd.Y0; sp.ts.fd = 1 + c->fd.row0; sp.ts.fd_fixed = c->fd.fixed; sp.ts.fd_h = c->fd.h;
        p.Y = c->fd.Y0; p.fd = 1 + c->fd.row0; p.fd_fixed = c->fd.fixed; p.fd_h = c->fd.h;
    }
    sp.dyn.cv4 = c->d_ang_T4.as<double>(); sp.dyn.cv2 = c->d_ang_cv2.as<double>(); sp.dyn.R = c->R;
    switch (nc) {
#define OBTG_CASE(NC_) case NC_: return launch_sep_dyn_elev_t<NC_>(c, sp, B);
        OBTG_NC_ELEV(OBTG_CASE)
#undef OBTG_CASE
    }
    return OBTG_ERR_UNSUPPORTED;
}

int launch_dynamics(obtg_ctx* c, const double* dY, const double* d_tf, int B, double bound, int is_max,
                    double max_rate, double* d_out_speed, double* d_out_ang)
{
    if (B <= 0) return OBTG_OK;
    int rc = ensure_tables(c);
    if (rc) return rc;
    if (d_out_ang && c->dim != 2) return OBTG_ERR_ARG;   // optimization.py:590-593 raises for dim != 2
    if (dyn_fast(c)) {
        AngParams p{};
        p.Y = dY; p.tf = d_tf; p.out = d_out_ang; p.out_speed = d_out_speed;
        p.n_veh = c->n_veh; p.total = B * c->n_veh;
        p.w2 = square_as_python(max_rate);
        const double b2 = square_as_python(bound);
        p.sp_sign = is_max ? -1.0 : 1.0; p.sp_offset = is_max ? b2 : -b2;
        p.W2n = c->d_ang_w2n.as<double>();
        p.W22n = c->d_ang_w22n.as<double>();
        p.Wn = c->d_ang_wn.as<double>();
        second_speed_rows(c, p);
        if (c->fd.Y0) { p.Y = c->fd.Y0; p.fd = 1 + c->fd.row0; p.fd_fixed = c->fd.fixed; p.fd_h = c->fd.h; }
        const int kid = d_out_ang ? OBTG_K_ANG_RATE : OBTG_K_SPEED;
        switch (c->deg + 1) {
#define OBTG_CASE(NC_) case NC_: return launch_dyn_t<NC_>(c, p, kid);
            OBTG_NC_DYN(OBTG_CASE)
#undef OBTG_CASE
        }
    }
    if (d_out_ang && dyn_fast_elev(c)) {
        AngElevParams q{};
        AngParams& p = q.a;
        p.Y = dY; p.tf = d_tf; p.out = d_out_ang; p.out_speed = d_out_speed;
        p.n_veh = c->n_veh; p.total = B * c->n_veh;
        p.w2 = square_as_python(max_rate);
        const double b2 = square_as_python(bound);
        p.sp_sign = is_max ? -1.0 : 1.0; p.sp_offset = is_max ? b2 : -b2;
        p.W2n = c->d_ang_w2n.as<double>();
        p.W22n = c->d_ang_w22n.as<double>();
        p.Wn = c->d_ang_wn.as<double>();
        second_speed_rows(c, p);
        if (c->fd.Y0) { p.Y = c->fd.Y0; p.fd = 1 + c->fd.row0; p.fd_fixed = c->fd.fixed; p.fd_h = c->fd.h; }
        q.cv4 = c->d_ang_T4.as<double>(); q.cv2 = c->d_ang_cv2.as<double>(); q.R = c->R;
        switch (c->deg + 1) {
#define OBTG_CASE(NC_) case NC_: return launch_dyn_elev_t<NC_>(c, q, OBTG_K_ANG_RATE);
            OBTG_NC_DYN(OBTG_CASE)
#undef OBTG_CASE
        }
    }
    if (d_out_speed && (rc = launch_speed(c, dY, d_tf, B, bound, is_max, d_out_speed))) return rc;
    if (d_out_speed && c->speed2.d_out &&
        (rc = launch_speed(c, dY, d_tf, B, c->speed2.bound, c->speed2.is_max, c->speed2.d_out))) return rc;
    if (d_out_ang && (rc = launch_ang_rate(c, dY, d_tf, B, max_rate, d_out_ang))) return rc;
    return OBTG_OK;
}

int launch_ang_rate(obtg_ctx* c, const double* dY, const double* d_tf, int B, double max_rate,
                    double* d_out)
{
    if (c->dim != 2) return OBTG_ERR_ARG;   // optimization.py:590-593 raises for dim != 2
    if (B <= 0) return OBTG_OK;
    int rc = ensure_tables(c);
    if (rc) return rc;
    if (dyn_fast(c) || dyn_fast_elev(c)) return launch_dynamics(c, dY, d_tf, B, 0.0, 1, max_rate, nullptr, d_out);
    if (c->fd.Y0) return kNeedBatch;
    GenParams g{};
    rc = gen_common(c, g);
    if (rc) return rc;
    const int m = c->deg + c->R, mc = m + 1, L2 = 2 * m + 1;
    if (m > 250) return OBTG_ERR_UNSUPPORTED;   // C(4m,2m) must stay finite in binary64
    g.Y = dY; g.tf = d_tf; g.out = d_out; g.B = B; g.offset = square_as_python(max_rate);
    const int mc8 = (mc + 7) & ~7, L28 = (L2 + 7) & ~7;
    const bool bal = angrate_balanced(m), bal2 = angrate_balanced2(m);
    const int padf = bal ? 8 : L28, padm = bal2 ? 8 : mc8;
    const size_t szm = padm + (bal2 ? mc : L2) + 16, szf = padf + (bal ? L2 : 4 * m + 1) + 16;
    // the first stage's scratch (8 mc doubles) starts where the squares' operands (2 szf) will live
    size_t lds = sizeof(double) * ((size_t)2 * mc8 + 2 * szm + std::max((size_t)2 * szf, (size_t)8 * mc));
    if (lds > 160 * 1024) return OBTG_ERR_UNSUPPORTED;
    if (lds > 48 * 1024)
        OBTG_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_generic_angrate),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ScopedKernelTimer t(c, OBTG_K_ANG_RATE);
    hipLaunchKernelGGL(k_generic_angrate, dim3((unsigned)((size_t)B * c->n_veh)), dim3(kWave), lds, c->stream, g);
    OBTG_HIP(c, hipGetLastError());
    return OBTG_OK;
}

int launch_fd_batch(obtg_ctx* c, const double* dY0, int n_fixed_cols, double h, int B, double* dY, int row0)
{
    const int rows = c->n_veh * c->dim, nc = c->deg + 1;
    if (nc - 2 * n_fixed_cols <= 0 || row0 < 0) return OBTG_ERR_ARG;
    if (row0 + B > rows * (nc - 2 * n_fixed_cols) + 1) return OBTG_ERR_ARG;
    const size_t total = (size_t)rows * nc * B;
    unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 256 * 8);
    ScopedKernelTimer t(c, OBTG_K_FD_BATCH);
    hipLaunchKernelGGL(k_fd_batch, dim3(blocks), dim3(256), 0, c->stream, dY0, dY, rows, nc, n_fixed_cols, h, B, row0);
    OBTG_HIP(c, hipGetLastError());
    return OBTG_OK;
}

static int bern_rows(obtg_ctx* c, BernParams& p, int ra, int rb, int rc_)
{
    int need[] = { ra, rb, rc_ };
    for (int v : need) if (v >= 0) { int o = binrow_offset(c, v); if (o < 0) return o; }
    p.o_a = ra >= 0 ? binrow_offset(c, ra) : 0;
    p.o_b = rb >= 0 ? binrow_offset(c, rb) : 0;
    p.o_c = rc_ >= 0 ? binrow_offset(c, rc_) : 0;
    p.bin = c->d_binrows.as<double>();
    return OBTG_OK;
}

int launch_bern_elev(obtg_ctx* c, const double* d_in, int rows, int n, int R, double* d_out)
{
    if (rows <= 0) return OBTG_OK;
    if (n < 0 || R < 0 || n + R + 1 > kMaxGenericLen) return OBTG_ERR_UNSUPPORTED;
    BernParams p{};
    p.a = d_in; p.out = d_out; p.rows = rows; p.n = n; p.R = R;
    int rc = bern_rows(c, p, n, R, n + R);
    if (rc) return rc;
    ScopedKernelTimer t(c, OBTG_K_BERN);
    hipLaunchKernelGGL(k_bern_elev, dim3(rows), dim3(kWave), sizeof(double) * (n + 1), c->stream, p);
    OBTG_HIP(c, hipGetLastError());
    return OBTG_OK;
}

int launch_bern_diff(obtg_ctx* c, const double* d_in, int rows, int n, double T, double* d_out)
{
    if (rows <= 0) return OBTG_OK;
    if (n < 1 || n + 1 > kMaxGenericLen) return OBTG_ERR_UNSUPPORTED;
    BernParams p{};
    p.a = d_in; p.out = d_out; p.rows = rows; p.n = n; p.T = T;
    ScopedKernelTimer t(c, OBTG_K_BERN);
    hipLaunchKernelGGL(k_bern_diff, dim3(rows), dim3(kWave), sizeof(double) * (n + 1), c->stream, p);
    OBTG_HIP(c, hipGetLastError());
    return OBTG_OK;
}

int launch_bern_split(obtg_ctx* c, const double* d_in, int rows, int n, double z, double* d_left, double* d_right)
{
    if (rows <= 0) return OBTG_OK;
    if (n < 0 || n + 1 > kMaxGenericLen) return OBTG_ERR_UNSUPPORTED;
    BernParams p{};
    p.a = d_in; p.out = d_left; p.rows = rows; p.n = n; p.T = z;
    ScopedKernelTimer t(c, OBTG_K_BERN);
    hipLaunchKernelGGL(k_bern_split, dim3(rows), dim3(kWave), sizeof(double) * 2 * (n + 1), c->stream, p, d_right);
    OBTG_HIP(c, hipGetLastError());
    return OBTG_OK;
}

int launch_bern_eval(obtg_ctx* c, const double* d_cpts, int rows, int n, const double* d_tau, int n_tau, double t0, double tf,
                     double* d_out)
{
    if (rows <= 0 || n_tau <= 0) return OBTG_OK;
    if (n < 0 || n + 1 > kMaxGenericLen) return OBTG_ERR_UNSUPPORTED;
    const int nc = n + 1;
    const size_t lds = sizeof(double) * ((size_t)nc + (size_t)kWave * (nc | 1));
    if (lds > 64 * 1024) return OBTG_ERR_UNSUPPORTED;
    ScopedKernelTimer t(c, OBTG_K_BERN);
    hipLaunchKernelGGL(k_bern_eval, dim3((unsigned)((n_tau + kWave - 1) / kWave), (unsigned)rows), dim3(kWave), lds, c->stream,
                       d_cpts, d_tau, n_tau, nc, t0, tf, d_out);
    OBTG_HIP(c, hipGetLastError());
    return OBTG_OK;
}

int launch_bern_mul(obtg_ctx* c, const double* d_a, const double* d_b, int rows, int m, int n,
                    double* d_out)
{
    if (rows <= 0) return OBTG_OK;
    if (m < 0 || n < 0 || m + n + 1 > kMaxGenericLen) return OBTG_ERR_UNSUPPORTED;
    BernParams p{};
    p.a = d_a; p.b = d_b; p.out = d_out; p.rows = rows; p.m = m; p.n = n;
    int rc = bern_rows(c, p, m, n, m + n);
    if (rc) return rc;
    ScopedKernelTimer t(c, OBTG_K_BERN);
    hipLaunchKernelGGL(k_bern_mul, dim3(rows), dim3(kWave), sizeof(double) * (m + n + 2), c->stream, p);
    OBTG_HIP(c, hipGetLastError());
    return OBTG_OK;
}

int launch_bern_normsq(obtg_ctx* c, const double* d_x, int d, int n, double* d_out)
{
    if (d <= 0 || n < 0 || 2 * n + 1 > kMaxGenericLen) return OBTG_ERR_UNSUPPORTED;
    BernParams p{};
    p.a = d_x; p.out = d_out; p.d = d; p.n = n;
    int rc = bern_rows(c, p, n, -1, 2 * n);
    if (rc) return rc;
    ScopedKernelTimer t(c, OBTG_K_BERN);
    hipLaunchKernelGGL(k_bern_normsq, dim3(1), dim3(kWave), sizeof(double) * d * (n + 1), c->stream, p);
    OBTG_HIP(c, hipGetLastError());
    return OBTG_OK;
}

int launch_euclidean_obj(obtg_ctx* c, const double* dY, int B, double* d_out)
{
    if (B <= 0) return OBTG_OK;
    ScopedKernelTimer t(c, OBTG_K_BERN);
    hipLaunchKernelGGL(k_euclid, dim3(B), dim3(kWave), 0, c->stream, dY, d_out, c->n_veh, c->dim, c->deg + 1);
    OBTG_HIP(c, hipGetLastError());
    return OBTG_OK;
}

// optimization.py:503-539: per vehicle (d/dt)^order pos -> normSquare().elev(R), control points
// summed.  order-1 plain derivative passes, then the speed-style sweep (one more derivative,
// product, elevation) with sign +1 and offset 0.
int launch_deriv_energy_obj(obtg_ctx* c, const double* dY, const double* d_tf, double tf0, int B, int order, double* d_out)
{
    if (B <= 0) return OBTG_OK;
    if (order < 1 || order > 4) return OBTG_ERR_ARG;
    const int rows = B * c->n_veh * c->dim, nc = c->deg + 1;
    int rc = c->ws_misc[3].reserve(sizeof(double) * (size_t)rows * nc);
    if (rc) return rc;
    if ((rc = c->ws_misc[6].reserve(sizeof(double) * (size_t)rows * nc))) return rc;
    if ((rc = c->ws_misc[4].reserve(sizeof(double) * (size_t)B * c->n_veh * (2 * c->deg + c->R + 1)))) return rc;
    // the reference passes ONE model['tf'] (optimization.py:294-308; tf differs per row only in time-optimal
    // problems, whose objective is x[-1]): tf0 is that value, d_tf[B] holds it B times (checked by the caller)
    const double* src = dY;
    double* bufs[2] = { c->ws_misc[3].as<double>(), c->ws_misc[6].as<double>() };
    for (int k = 0; k < order - 1; ++k) {
        if ((rc = launch_bern_diff(c, src, rows, c->deg, tf0, bufs[k & 1]))) return rc;
        src = bufs[k & 1];
    }
    if ((rc = launch_speed(c, src, d_tf, B, 0.0, 0, c->ws_misc[4].as<double>()))) return rc;
    ScopedKernelTimer t(c, OBTG_K_BERN);
    hipLaunchKernelGGL(k_rowsum, dim3(B), dim3(kWave), 0, c->stream, c->ws_misc[4].as<double>(), d_out,
                       c->n_veh * (2 * c->deg + c->R + 1));
    OBTG_HIP(c, hipGetLastError());
    return OBTG_OK;
}

}  // namespace obtg
