// Device side of the Bernstein sweeps (fast path): shared by bern_kernels.hip (stand-alone
// kernels) and gjk_kernels.hip (the one-launch pair sweep).  Floating-point contraction is the
// including translation unit's business: gjk_kernels.hip brackets this header with
// `#pragma clang fp contract(fast)` so that both units generate the same arithmetic.
#pragma once
#include <hip/hip_runtime.h>

#include "obtg_internal.h"

namespace obtg {

// Read-only coefficient tables are addressed through the CONSTANT address space: their loads
// have wave-uniform addresses and must become scalar loads (s_load_dwordxN into SGPRs, usable
// directly as v_fma_f64 operands).  Through a plain global pointer the compiler cannot prove
// that the kernel's own stores do not clobber the table and falls back to one vector load +
// s_waitcnt vmcnt(0) per product row -- a serialised L2 round trip per row (measured with
// s_memtime stamps: 8.5 k cycles for the 200 FMAs of one group, 1.7 k after this change).
typedef const double __attribute__((address_space(4))) * ctab_t;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
__device__ __forceinline__ ctab_t as_ctab(const double* p) { return (ctab_t)p; }
#pragma clang diagnostic pop

// The constraint vectors are written once and never re-read by these kernels: non-temporal
// stores keep the 390 MB output stream from evicting the control points from L2 / Infinity
// Cache (measured: -10 % kernel time on the temporal sweep).
typedef double d2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_nt(double* p, double v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void store_nt2(double* p, double v0, double v1)
{
    d2_t v; v.x = v0; v.y = v1;
    __builtin_nontemporal_store(v, reinterpret_cast<d2_t*>(p));
}

// =====================================================================================
//  fast path: register-resident product, one item per lane
// =====================================================================================
struct NsParams {
    const double* __restrict__ Y;     // [B][n_veh*DIM][NC]
    const double* __restrict__ obs;   // [n_obs][DIM]   (pair mode)
    const double* __restrict__ tf;    // [B]            (vehicle mode)
    const int2* __restrict__ pairs;   // [P]            (pair mode)
    const double* __restrict__ W2;    // folded weights [L][NC]
    const double* __restrict__ Tt;    // elevation tables (R > 0), three rows back to back:
                                      //   scale[L]      = C(2n, j)
                                      //   binp[R+2L-1+8] = C(R, m) for m = -(L-1) .. R+L-1+8 (0 outside 0..R)
                                      //   inv[L+R+8]    = 1 / C(2n+R, k) (0 past the end)
    const double* __restrict__ Td;    // dense elevation table, transposed: Td[k][j] = T[j][k], [L+R][L] (R > 0)
    double* __restrict__ out;
    int n_veh, n_obj, R;
    int item_begin, item_count;       // items of this launch (pairs or vehicles)
    int groups_per_wg, wgs_per_row;
    int stage_slots;                  // LDS slots reserved for staged objects
    int stage_all;                    // 1: every object of the row is staged, slot == object id
    int tile_rows;                    // rows of the per-wave transposition tile (64, 32 or 16)
    int waves;                        // waves per workgroup
    int tiling;                       // 1: row-window tiles (large swarms), see k_normsq_elev
    const int2* __restrict__ tiles;   // [wgs_per_row] (first row, first column) of each tile
    double sign, offset;              // out = sign * value + offset
    int fd, fd_fixed;                 // fd != 0: Y is ONE row; evaluation row b >= 1 = Y with its (b-1)-th free control point
    double fd_h;                      //          advanced by fd_h (obtg_fd_batch_dev's rows), formed while staging
};

// element of an evaluation row [n_rows][nc] that row b of a virtual finite-difference batch advances (-1: none)
__device__ __forceinline__ int fd_element(int fd, int fixed, int nc, int b)
{
    if (!fd || b <= 0) return -1;
    const int free_cols = nc - 2 * fixed, kq = b - 1, pr = kq / free_cols;
    return pr * nc + fixed + (kq - pr * free_cols);
}

constexpr int kTileK = 32;            // k-chunk of the transposition tile when R > 0

template <int NC, int DIM>
struct NsShape {
    static constexpr int N = NC - 1;
    static constexpr int L = 2 * N + 1;
    static constexpr int VLEN = DIM * NC;
    static constexpr int VP = (VLEN % 2 == 0) ? VLEN + 1 : VLEN;   // odd pitch (doubles)
    static constexpr int TPF = (L % 2 == 0) ? L + 1 : L;           // full-tile pitch (R == 0)
    static constexpr int TPC = kTileK + 1;                          // chunked-tile pitch (R > 0)
};

// stage objects [lo, lo+cnt) of one evaluation row into LDS slots [slot0, slot0+cnt)
template <int NC, int DIM>
__device__ __forceinline__ void stage_objects(double* __restrict__ vl, const double* __restrict__ Yrow,
                                              const double* __restrict__ obs, int n_veh, int lo,
                                              int cnt, int slot0, int tid, int nthreads, int fd_e = -1, double fd_h = 0.0)
{
    using S = NsShape<NC, DIM>;
    const int total = cnt * S::VLEN;
    for (int e = tid; e < total; e += nthreads) {
        const int v = e / S::VLEN, r = e - v * S::VLEN;
        const int obj = lo + v;
        double val;
        if (obj < n_veh) { const int ee = obj * S::VLEN + r; val = Yrow[ee]; if (ee == fd_e) val += fd_h; }
        else val = obs[(obj - n_veh) * DIM + r / NC];   // constant curve (optimization.py:86-98)
        vl[(slot0 + v) * S::VP + r] = val;
    }
}

// LDS traffic between the lanes of ONE wave only needs wave-level ordering (a wave's DS
// operations complete in issue order); a workgroup barrier here would also be unsafe because
// the waves of a workgroup run different numbers of groups.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// (d/2) * sum_q a_q^2 as Bernstein coefficients c[0..L)
template <int NC, int DIM>
__device__ __forceinline__ void normsq_coeffs(const double (&a)[DIM][NC], ctab_t W2,
                                              double (&c)[2 * NC - 1])
{
    constexpr int N = NC - 1, L = 2 * N + 1;
#pragma unroll
    for (int k = 0; k < L; ++k) {
        double s = 0.0;
#pragma unroll
        for (int j = (k - N > 0 ? k - N : 0); 2 * j <= k; ++j) {
            double xa = a[0][j] * a[0][k - j];
#pragma unroll
            for (int q = 1; q < DIM; ++q) xa = fma(a[q][j], a[q][k - j], xa);
            s = fma(W2[k * NC + j], xa, s);
        }
        c[k] = s;
    }
}

// write a full [n_valid][LR] tile (pitch TP) as one contiguous run of n_valid*LR doubles
template <int LR, int TP>
__device__ __forceinline__ void flush_full(const double* __restrict__ tile, double* __restrict__ gout,
                                           size_t gbase /* element index in out */, int n_valid, int lane)
{
    const int total = n_valid * LR;
    const int shift = (int)(gbase & 1);   // make the 16-byte stores 16-byte aligned
    const int npairs = (total + shift + 1) >> 1;
    if (LR == TP) {                       // the tile is the output run itself: no row / column arithmetic
        if (shift == 0 && (total & 1) == 0 && (reinterpret_cast<size_t>(tile) & 15) == 0) {
            const d2_t* t2 = reinterpret_cast<const d2_t*>(tile);
            d2_t* g2 = reinterpret_cast<d2_t*>(gout + gbase);
            for (int m = lane; m < (total >> 1); m += kWave) __builtin_nontemporal_store(t2[m], g2 + m);
            return;
        }
        for (int m = lane; m < npairs; m += kWave) {
            const int e0 = 2 * m - shift, e1 = e0 + 1;
            if (e0 >= 0 && e1 < total) store_nt2(gout + gbase + e0, tile[e0], tile[e1]);
            else if (e0 >= 0) store_nt(gout + gbase + e0, tile[e0]);
            else if (e1 < total) store_nt(gout + gbase + e1, tile[e1]);
        }
        return;
    }
    for (int m = lane; m < npairs; m += kWave) {
        const int e0 = 2 * m - shift, e1 = e0 + 1;
        double v0 = 0.0, v1 = 0.0;
        if (e0 >= 0) { const int pr = e0 / LR, q = e0 - pr * LR; v0 = tile[pr * TP + q]; }
        if (e1 < total) { const int pr = e1 / LR, q = e1 - pr * LR; v1 = tile[pr * TP + q]; }
        if (e0 >= 0 && e1 < total) store_nt2(gout + gbase + e0, v0, v1);
        else if (e0 >= 0) store_nt(gout + gbase + e0, v0);
        else if (e1 < total) store_nt(gout + gbase + e1, v1);
    }
}

// write chunk columns [k0, k0+kc) of n_valid rows; row r lives at gout[grow + r*LR + ...]
template <int TP>
__device__ __forceinline__ void flush_chunk(const double* __restrict__ tile, double* __restrict__ gout,
                                            size_t grow, int LR, int k0, int kc, int n_valid, int lane)
{
    const int total = n_valid * kc;
    if (kc == kTileK) {           // full chunk: power-of-two row length, no integer division
        static_assert((kTileK & (kTileK - 1)) == 0, "kTileK must be a power of two");
        for (int e = lane; e < total; e += kWave) {
            const int pr = e / kTileK, q = e & (kTileK - 1);
            gout[grow + (size_t)pr * LR + k0 + q] = tile[pr * TP + q];
        }
        return;
    }
    for (int e = lane; e < total; e += kWave) {
        const int pr = e / kc, q = e - pr * kc;
        gout[grow + (size_t)pr * LR + k0 + q] = tile[pr * TP + q];
    }
}

// ---- degree elevation as a binomially scaled convolution, lane = item -----------------------------------------
//     elev(a, Q)_k = (1 / C(P+Q, k)) sum_j [C(P, j) a_j] C(Q, k-j)
// All output columns share ONE weight row C(Q, .): a block of kElevBlock columns walks a window of that row with
// one wave-uniform scalar per step (kElevBlock independent FMA chains) instead of fetching P+1 weights per column.
// Outputs leave through a per-wave LDS tile of kElevChunk columns so that stores are 256-byte runs.
constexpr int kConvPad = 8;           // zero entries behind the padded binomial row and the 1/C(P+Q, .) row
constexpr int kElevChunk = 32;
constexpr int kElevBlock = 8;

template <int LIN>
__device__ __forceinline__ void elev_store_chunk(const double* __restrict__ tile, double* __restrict__ gout, size_t grow,
                                                 int LR, int k0, int kc, int n_valid, int lane)
{
    constexpr int TP = kElevChunk + 1;
    if (kc == kElevChunk && n_valid == kWave) {
        // full tile: 32 store instructions of 2 x 256-byte runs; the LDS reads go out eight at a time (one read,
        // one wait, one store per trip left the loop bound by the LDS latency: 32 x ~100 cycles per chunk)
        const int q = lane & (kElevChunk - 1), half = lane >> 5;
        const double* t = tile + half * TP + q;
        double* g = gout + grow + (size_t)half * LR + k0 + q;
#pragma unroll
        for (int it0 = 0; it0 < kWave / 2; it0 += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = t[(it0 + u) * 2 * TP];
#pragma unroll
            for (int u = 0; u < 8; ++u) store_nt(g + (size_t)(it0 + u) * 2 * LR, v[u]);
        }
        return;
    }
    if (kc == kElevChunk) {
        for (int e = lane; e < n_valid * kElevChunk; e += kWave) {
            const int pr = e / kElevChunk, q = e & (kElevChunk - 1);
            store_nt(gout + grow + (size_t)pr * LR + k0 + q, tile[pr * TP + q]);
        }
        return;
    }
    for (int e = lane; e < n_valid * kc; e += kWave) {
        const int pr = e / kc, q = e - pr * kc;
        store_nt(gout + grow + (size_t)pr * LR + k0 + q, tile[pr * TP + q]);
    }
}

// sa[i] = sum_j a[j] c[(k0+i) - j], i < kElevBlock, with cp = (padded row) + k0: cp[m] = c[k0 - (LIN-1) + m]
template <int LIN>
__device__ __forceinline__ void conv_block2(const ctab_t cp, const double (&a)[LIN], const double (&b)[LIN],
                                            double (&sa)[kElevBlock], double (&sb)[kElevBlock])
{
#pragma unroll
    for (int i = 0; i < kElevBlock; ++i) sa[i] = sb[i] = 0.0;
#pragma unroll
    for (int m = 0; m < LIN - 1 + kElevBlock; ++m) {
        const double c = cp[m];
#pragma unroll
        for (int i = 0; i < kElevBlock; ++i) {
            const int j = i + LIN - 1 - m;
            if (j >= 0 && j < LIN) { sa[i] = fma(c, a[j], sa[i]); sb[i] = fma(c, b[j], sb[i]); }
        }
    }
}

// the same walk with the window in registers (per-lane column block): w[m] = c[k0 - (LIN-1) + m]
template <int LIN>
__device__ __forceinline__ void conv_block1_reg(const double (&w)[LIN - 1 + kElevBlock], const double (&a)[LIN],
                                                double (&sa)[kElevBlock])
{
#pragma unroll
    for (int i = 0; i < kElevBlock; ++i) sa[i] = 0.0;
#pragma unroll
    for (int m = 0; m < LIN - 1 + kElevBlock; ++m) {
#pragma unroll
        for (int i = 0; i < kElevBlock; ++i) {
            const int j = i + LIN - 1 - m;
            if (j >= 0 && j < LIN) sa[i] = fma(w[m], a[j], sa[i]);
        }
    }
}

template <int LIN>
__device__ __forceinline__ void conv_block1(const ctab_t cp, const double (&a)[LIN], double (&sa)[kElevBlock])
{
#pragma unroll
    for (int i = 0; i < kElevBlock; ++i) sa[i] = 0.0;
#pragma unroll
    for (int m = 0; m < LIN - 1 + kElevBlock; ++m) {
        const double c = cp[m];
#pragma unroll
        for (int i = 0; i < kElevBlock; ++i) {
            const int j = i + LIN - 1 - m;
            if (j >= 0 && j < LIN) sa[i] = fma(c, a[j], sa[i]);
        }
    }
}

// ELEV = false: DEG_ELEV == 0 (the product IS the output); ELEV = true: R > 0.  Separate
// instantiations so that the R > 0 code (two weight columns in registers) does not cost the
// R == 0 kernel its occupancy.
// Temporal separation (optimization.py:311-346, R == 0, d == 2, no point obstacles) of one evaluation
// row whose vehicles are ALREADY staged point-major in LDS -- (x_c, y_c) pairs with object pitch
// `vpq`, the layout of the planar GJK sweep -- so that the GJK workgroups of a row can write the
// row's separation block themselves (gjk_kernels.hip, pair sweep): same differences, weights and
// output transform as k_normsq_elev<NC, 2, 0, false, false>, hence the same bits.
// The calling workgroup takes the 64-pair groups g = g_first + (wave + it*n_waves)*g_step, it = 0, 1, ...
struct TsepXYParams {
    const int2* __restrict__ pairs;   // [n_pairs] lexicographic (i, j), i < j < n_veh
    const double* __restrict__ W2;    // folded product weights of (deg, dim = 2)
    double* __restrict__ out;         // [B][n_pairs][2n+1]; nullptr = no temporal work
    int n_pairs;
    double sign, offset;
    const double* __restrict__ Tt;    // DEG_ELEV > 0 (structured step): elevation as a scaled convolution (NsParams::Tt)
    int R;
    int n_veh, obs_shift;             // pairs may name point obstacles (object ids >= n_veh, optimization.py:86-98): those are
                                      // staged obs_shift slots further on (behind the polygons of the hull sweep); 0, 0: none
};
__device__ __forceinline__ int tsep_slot(const TsepXYParams& t, int obj) { return obj < t.n_veh ? obj : obj + t.obs_shift; }

template <int NC>
__device__ __forceinline__ void tsep_groups_from_xy(const TsepXYParams& t, const double2* xy, const int vpq,
                                                    const int b, const int g_first, const int g_step,
                                                    double* tile_base, const int TR, const int it_lo, const int it_hi)
{
    using S = NsShape<NC, 2>;
    constexpr int L = S::L;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    double* tile = tile_base + wave * (TR * S::TPF);
    const int n_groups = (t.n_pairs + kWave - 1) / kWave;
    // a wave's it-th group is g_first + (wave + it * n_waves) * g_step; this call does it_lo <= it < it_hi
    for (int it = it_lo; it < it_hi; ++it) {
        const int g = g_first + (wave + it * n_waves) * g_step;
        if (g >= n_groups) break;
        const int itg = g * kWave;
        const int n_valid = min(kWave, t.n_pairs - itg);
        const int item = min(itg + lane, t.n_pairs - 1);      // idle lanes recompute the last item
        const int2 ij = t.pairs[item];
        const double2* vi = xy + (t.obs_shift ? tsep_slot(t, ij.x) : ij.x) * vpq;
        const double2* vj = xy + (t.obs_shift ? tsep_slot(t, ij.y) : ij.y) * vpq;
        double a[2][NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const double2 pi = vi[c], pj = vj[c];
            a[0][c] = pi.x - pj.x;
            a[1][c] = pi.y - pj.y;
        }
        double cf[L];
        normsq_coeffs<NC, 2>(a, as_ctab(t.W2), cf);
        const size_t row = (size_t)b * t.n_pairs + (size_t)itg;
        for (int r0 = 0; r0 < n_valid; r0 += TR) {
            if (lane >= r0 && lane < r0 + TR && lane < n_valid) {
#pragma unroll
                for (int k = 0; k < L; ++k) tile[(lane - r0) * S::TPF + k] = t.sign * cf[k] + t.offset;
            }
            wave_sync();
            flush_full<L, S::TPF>(tile, t.out, (row + r0) * L, min(TR, n_valid - r0), lane);
            wave_sync();
        }
    }
}

// Structured finite-difference step (gjk_kernels.hip k_step_fd_structured), row 0's part: ONE wave evaluates the 64-pair
// group g of the staged row into `tile`, laid out as the output run itself ([pair][2n+1], no padding); returns the
// number of pairs in the group.  Arithmetic as above.
template <int NC>
__device__ __forceinline__ int tsep_group_to_tile(const TsepXYParams& t, const double2* xy, const int vpq, const int g, double* tile)
{
    using S = NsShape<NC, 2>;
    constexpr int L = S::L;
    const int lane = threadIdx.x & (kWave - 1);
    const int itg = g * kWave;
    const int n_valid = min(kWave, t.n_pairs - itg);
    const int item = min(itg + lane, t.n_pairs - 1);
    const int2 ij = t.pairs[item];
    const double2* vi = xy + ij.x * vpq;
    const double2* vj = xy + ij.y * vpq;
    double a[2][NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const double2 pi = vi[c], pj = vj[c];
        a[0][c] = pi.x - pj.x;
        a[1][c] = pi.y - pj.y;
    }
    double cf[L];
    normsq_coeffs<NC, 2>(a, as_ctab(t.W2), cf);
    if (lane < n_valid) {
#pragma unroll
        for (int k = 0; k < L; ++k) tile[lane * L + k] = t.sign * cf[k] + t.offset;
    }
    return n_valid;
}

// ... and a perturbed row's part: the separation rows of every pair that contains vehicle v of row b, one pair per
// lane, each lane writing its own 8 (2n+1)-byte run (n_veh - 1 pairs per row: a hundredth of the block).
template <int NC>
__device__ __forceinline__ void tsep_rows_of_vehicle(const TsepXYParams& t, const double2* xy, const int vpq, const int b,
                                                     const int n_veh, const int v)
{
    using S = NsShape<NC, 2>;
    constexpr int L = S::L;
    for (int u0 = threadIdx.x; u0 < n_veh - 1; u0 += blockDim.x) {
        const int u = u0 < v ? u0 : u0 + 1;
        const int i = min(u, v), j = max(u, v);
        const int q = i * (2 * n_veh - i - 1) / 2 + (j - i - 1);        // position of (i, j) in the lexicographic pair list
        const double2* vi = xy + i * vpq;
        const double2* vj = xy + j * vpq;
        double a[2][NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const double2 pi = vi[c], pj = vj[c];
            a[0][c] = pi.x - pj.x;
            a[1][c] = pi.y - pj.y;
        }
        double cf[L];
        normsq_coeffs<NC, 2>(a, as_ctab(t.W2), cf);
        double* o = t.out + ((size_t)b * t.n_pairs + q) * L;
#pragma unroll
        for (int k = 0; k < L; ++k) o[k] = t.sign * cf[k] + t.offset;
    }
}

// Structured finite-difference step with DEG_ELEV = R > 0 (gjk_kernels.hip k_step_fd_structured<NC, true>), row 0's part:
// the workgroup (four waves) evaluates the 64-pair group g of the staged row -- wave 0 the products, scaled, into chT;
// then every wave its share of the passes of rpp rows x all 2n+R+1 columns, exactly as normsq_elev_body's full-row form
// does (lane = (row of the pass, block of 8 columns), window of the binomial row in registers, conv_block1_reg) -- and
// streams each pass from its output tile into every batch row b0 <= b < b1, leaving out the rows of pairs that contain
// batch row b's own vehicle (fd_element).  chT: [64][2n+1] doubles, otile_base: 4 x rpp x nb2 x 9 doubles.
template <int NC>
__device__ __forceinline__ void tsep_elev_group_stream(const TsepXYParams& t, const double2* xy, const int vpq, const int g,
                                                       double* chT, double* otile_base, const int b0, const int b1,
                                                       const int fd, const int fd_fixed)
{
    using S = NsShape<NC, 2>;
    constexpr int L = S::L;
    const int LR = L + t.R;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const int itg = g * kWave;
    const int n_valid = min(kWave, t.n_pairs - itg);
    const ctab_t escale = as_ctab(t.Tt);
    if (wave == 0) {
        const int item = min(itg + lane, t.n_pairs - 1);
        const int2 ij = t.pairs[item];
        const double2* vi = xy + ij.x * vpq;
        const double2* vj = xy + ij.y * vpq;
        double a[2][NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const double2 pi = vi[c], pj = vj[c];
            a[0][c] = pi.x - pj.x;
            a[1][c] = pi.y - pj.y;
        }
        double cf[L];
        normsq_coeffs<NC, 2>(a, as_ctab(t.W2), cf);
        if (lane < n_valid) {
#pragma unroll
            for (int j = 0; j < L; ++j) chT[lane * L + j] = cf[j] * escale[j];
        }
    }
    __syncthreads();
    const double* ebin_g = t.Tt + L;
    const double* einv_g = ebin_g + (t.R + 2 * L - 1 + kConvPad);
    const int nblk = (LR + kElevBlock - 1) / kElevBlock;        // <= 64 (the launcher checks 2n + R + 1 <= 512)
    int nb2 = 1;
    while (nb2 < nblk) nb2 <<= 1;
    const int rpp = kWave / nb2;                                 // rows per pass
    const int cb = lane & (nb2 - 1), rs = lane / nb2;
    constexpr int BP = kElevBlock + 1;
    const int opitch = nb2 * BP;
    double* otile = otile_base + wave * (rpp * opitch);
    const bool colv = cb < nblk;
    double w[L - 1 + kElevBlock], inv8[kElevBlock];
#pragma unroll
    for (int m = 0; m < L - 1 + kElevBlock; ++m) w[m] = colv ? ebin_g[cb * kElevBlock + m] : 0.0;
#pragma unroll
    for (int i = 0; i < kElevBlock; ++i) inv8[i] = colv ? einv_g[cb * kElevBlock + i] : 0.0;
    for (int r0 = wave * rpp; r0 < n_valid; r0 += n_waves * rpp) {
        const int rr = r0 + rs;
        if (rr < n_valid && colv) {
            double ch[L], sa[kElevBlock];
#pragma unroll
            for (int j = 0; j < L; ++j) ch[j] = chT[rr * L + j];
            conv_block1_reg<L>(w, ch, sa);
            double* o = otile + rs * opitch + cb * BP;
#pragma unroll
            for (int i = 0; i < kElevBlock; ++i) o[i] = t.sign * (sa[i] * inv8[i]) + t.offset;
        }
        wave_sync();
        const int rows = min(rpp, n_valid - r0);
        // the pass's rows are one contiguous run of rows x (2n+R+1) doubles in every batch row: written back to back per
        // batch row (the L2 merges the lines neighbouring rows share), from registers (each lane keeps its columns)
        // (the pass as ONE run in 16-byte pieces with a per-element pair test, the R = 0 stream's form: bit-identical, 0.90
        // instead of 0.84 ms at C5 on the same box -- the per-piece predicates cost more than the wider stores save; with
        // non-temporal stores 0.92, and the 8-byte form below with non-temporal stores 0.89)
        constexpr int kMaxRows = 8, kMaxCols = 4;                                   // rpp <= 8 (LR > 56), ceil(LR / 64) <= 4 (LR <= 256); else the LDS form below
        if (rows <= kMaxRows && LR <= kMaxCols * kWave) {
            double v[kMaxRows][kMaxCols];
            int pi[kMaxRows], pj[kMaxRows];
#pragma unroll
            for (int q = 0; q < kMaxRows; ++q) {
                const int2 ij = q < rows ? t.pairs[itg + r0 + q] : make_int2(-2, -2);    // wave-uniform
                pi[q] = ij.x; pj[q] = ij.y;
#pragma unroll
                for (int cc = 0; cc < kMaxCols; ++cc) {
                    const int kc = lane + cc * kWave;
                    v[q][cc] = (q < rows && kc < LR) ? otile[q * opitch + (kc >> 3) * BP + (kc & 7)] : 0.0;
                }
            }
            for (int b = b0; b < b1; ++b) {
                const int fd_e = fd_element(fd, fd_fixed, NC, b);
                const int vb = fd_e >= 0 ? fd_e / (2 * NC) : -1;
                double* gp = t.out + ((size_t)b * t.n_pairs + (size_t)(itg + r0)) * LR + lane;
#pragma unroll
                for (int q = 0; q < kMaxRows; ++q) {
                    if (q < rows && pi[q] != vb && pj[q] != vb) {                    // (a pair of row b's own vehicle: its workgroup writes it)
#pragma unroll
                        for (int cc = 0; cc < kMaxCols; ++cc)
                            if (lane + cc * kWave < LR) gp[(size_t)q * LR + cc * kWave] = v[q][cc];
                    }
                }
            }
        } else {
            for (int b = b0; b < b1; ++b) {
                const int fd_e = fd_element(fd, fd_fixed, NC, b);
                const int vb = fd_e >= 0 ? fd_e / (2 * NC) : -1;
                for (int q = 0; q < rows; ++q) {
                    const int2 ij = t.pairs[itg + r0 + q];
                    if (ij.x == vb || ij.y == vb) continue;
                    double* gp = t.out + ((size_t)b * t.n_pairs + (size_t)(itg + r0 + q)) * LR;
                    for (int kc = lane; kc < LR; kc += kWave) gp[kc] = otile[q * opitch + (kc >> 3) * BP + (kc & 7)];
                }
            }
        }
        wave_sync();
    }
}

// ... and a perturbed row's part with DEG_ELEV > 0: the elevated separation rows of every pair that contains vehicle v of
// row b, one pair per lane, each lane its own run (the arithmetic of k_tsep_fd's elevated form, which equals the batch's).
template <int NC>
__device__ __forceinline__ void tsep_elev_rows_of_vehicle(const TsepXYParams& t, const double2* xy, const int vpq, const int b,
                                                          const int n_veh, const int v)
{
    using S = NsShape<NC, 2>;
    constexpr int L = S::L;
    const int LR = L + t.R;
    const ctab_t escale = as_ctab(t.Tt), ebin = escale + L, einv = ebin + (t.R + 2 * L - 1 + kConvPad);
    for (int u0 = threadIdx.x; u0 < n_veh - 1; u0 += blockDim.x) {
        const int u = u0 < v ? u0 : u0 + 1;
        const int i = min(u, v), j = max(u, v);
        const int q = i * (2 * n_veh - i - 1) / 2 + (j - i - 1);
        const double2* vi = xy + i * vpq;
        const double2* vj = xy + j * vpq;
        double a[2][NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const double2 pi = vi[c], pj = vj[c];
            a[0][c] = pi.x - pj.x;
            a[1][c] = pi.y - pj.y;
        }
        double cf[L];
        normsq_coeffs<NC, 2>(a, as_ctab(t.W2), cf);
        double ch[L];
#pragma unroll
        for (int jj = 0; jj < L; ++jj) ch[jj] = cf[jj] * escale[jj];
        double* o = t.out + ((size_t)b * t.n_pairs + q) * LR;
        for (int k = 0; k < LR; k += kElevBlock) {
            double sa[kElevBlock];
            conv_block1<L>(ebin + k, ch, sa);
#pragma unroll
            for (int ii = 0; ii < kElevBlock; ++ii)
                if (k + ii < LR) o[k + ii] = t.sign * (sa[ii] * einv[k + ii]) + t.offset;
        }
    }
}

// The same for the TILED sweep (large rows): the workgroup has staged the vehicles of one TA x 64 tile of the pair
// matrix -- rows ti0 .. ti0+ta-1 at LDS slots rowslot[.], columns tj0 .. tj0+63 at colslot[.] -- and writes the
// separation rows of that tile's pairs: row i's pairs (i, j), j in the window, are one contiguous run of the output
// (normsq_elev_body's `tiling` arrangement on the point-major layout).  n_veh == n_obj (no point obstacles).
// Wave `wave` takes the tile rows g = wave + it * n_waves, it_lo <= it < it_hi.
template <int NC>
__device__ __forceinline__ void tsep_tile_from_xy(const TsepXYParams& t, const double2* xy, const int vpq, const int b,
                                                  const int n_veh, const int ti0, const int tj0, const int ta,
                                                  const unsigned short* rowslot, const unsigned short* colslot,
                                                  double* tile_base, const int TR, const int it_lo, const int it_hi)
{
    using S = NsShape<NC, 2>;
    constexpr int L = S::L;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    double* tile = tile_base + wave * (TR * S::TPF);
    for (int it = it_lo; it < it_hi; ++it) {
        const int g = wave + it * n_waves;
        if (g >= ta) break;
        const int i = ti0 + g;
        if (i >= n_veh - 1) break;
        const int j = tj0 + lane;
        const bool valid = j > i && j < n_veh;
        const unsigned long long m = __ballot(valid);
        if (m == 0ull) continue;
        const int lane0 = __ffsll((long long)m) - 1, n_valid = __popcll(m);   // valid lanes are [lane0, lane0 + n_valid)
        const int jj = min(max(j, i + 1), n_veh - 1);                        // idle lanes recompute a valid pair
        const double2* vi = xy + (int)rowslot[g] * vpq;
        const double2* vj = xy + (int)colslot[jj - tj0] * vpq;
        double a[2][NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const double2 pi = vi[c], pj = vj[c];
            a[0][c] = pi.x - pj.x;
            a[1][c] = pi.y - pj.y;
        }
        double cf[L];
        normsq_coeffs<NC, 2>(a, as_ctab(t.W2), cf);
        const long tri = (long)i * n_veh - (long)i * (i + 1) / 2 - i - 1;    // pair index of (i, j) = tri + j
        const size_t row = (size_t)b * t.n_pairs + (size_t)(tri + tj0 + lane0);
        const int r = lane - lane0;
        for (int r0 = 0; r0 < n_valid; r0 += TR) {
            if (valid && r >= r0 && r < r0 + TR) {
#pragma unroll
                for (int k = 0; k < L; ++k) tile[(r - r0) * S::TPF + k] = t.sign * cf[k] + t.offset;
            }
            wave_sync();
            flush_full<L, S::TPF>(tile, t.out, (row + r0) * L, min(TR, n_valid - r0), lane);
            wave_sync();
        }
    }
}

// per-wave LDS of the elevated full-row form: the scaled products of HALF a 64-item group (the lanes keep theirs in
// registers and hand them over 32 rows at a time) and the 64 x 9 output tile of a pass.  32 instead of 64 rows: 10 instead of
// 15 KB per wave, three workgroups of four waves per CU instead of two at C5 (measured: the same 0.52 ms -- the kernel is
// bound by its 968-byte rows' write rate, not by occupancy; kept for the LDS it leaves to others)
constexpr int kElevHalfRows = 32;
__host__ __device__ constexpr int ns_elev_tile_doubles(int L) { return kElevHalfRows * L + kWave * (kElevBlock + 1); }

// (b, w) = evaluation row and workgroup index inside the row; lds = the workgroup's dynamic LDS.
// A device function so that the pair sweep can run it next to the GJK workgroups in ONE launch
// (gjk_kernels.hip k_pair_sweep); k_normsq_elev below is the stand-alone kernel.
template <int NC, int DIM, int MODE /*0 = pairs, 1 = vehicles*/, bool MINONLY, bool ELEV>
__device__ __forceinline__ void normsq_elev_body(const NsParams& p, const int b, const int w, double* lds)
{
    using S = NsShape<NC, DIM>;
    constexpr int N = S::N, L = S::L;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const int chunk = kWave * p.groups_per_wg;
    const int it0 = p.item_begin + w * chunk;
    const int it_end = min(p.item_begin + p.item_count, it0 + chunk);
    if (!p.tiling && it0 >= it_end) return;

    // LDS: [staged objects: stage_slots * VP][per-wave transposition tiles]
    double* vl = lds;
    double* tile = lds + p.stage_slots * S::VP + wave * (ELEV ? ns_elev_tile_doubles(S::L) : p.tile_rows * S::TPF);

    // ---- stage the objects this workgroup touches
    const double* Yrow = p.fd ? p.Y : p.Y + (size_t)b * p.n_veh * S::VLEN;
    const int fd_e = fd_element(p.fd, p.fd_fixed, NC, b);
    // three staging schemes:
    //   stage_all  (small swarms): every object of the row, slot == object id;
    //   tiling     (large swarms): the workgroup owns rows ti0..ti0+groups_per_wg-1 of the pair
    //              triangle restricted to the 64-wide column window [tj0, tj0+64): it stages
    //              n_waves + 64 objects, each wave handles one row segment (contiguous pairs);
    //   otherwise: a chunk of lexicographic pairs touches rows i0..i0+nI-1 (segment I), the
    //              j-range of its first row (segment A) and of the later rows (segment B).
    int i0 = 0, nI = 0, a_lo = 0, nA = 0, b_lo = 0, ti0 = 0, tj0 = 0;
    if (p.stage_all) {
        stage_objects<NC, DIM>(vl, Yrow, p.obs, p.n_veh, 0, p.n_obj, 0, threadIdx.x, blockDim.x, fd_e, p.fd_h);
    } else if (MODE == 0 && p.tiling) {
        const int2 t = p.tiles[w];
        ti0 = t.x; tj0 = t.y;
        stage_objects<NC, DIM>(vl, Yrow, p.obs, p.n_veh, ti0, min(p.groups_per_wg, p.n_obj - ti0), 0, threadIdx.x, blockDim.x, fd_e, p.fd_h);
        stage_objects<NC, DIM>(vl, Yrow, p.obs, p.n_veh, tj0, min(kWave, p.n_obj - tj0), p.groups_per_wg, threadIdx.x, blockDim.x, fd_e, p.fd_h);
    } else if (MODE == 0) {
        const int2 f = p.pairs[it0], l = p.pairs[it_end - 1];
        i0 = f.x; nI = l.x - f.x + 1;
        a_lo = f.y;
        nA = (nI == 1) ? (l.y - f.y + 1) : (p.n_obj - f.y);
        b_lo = i0 + 2;
        const int nB = (nI == 1) ? 0 : max(0, ((nI >= 3) ? p.n_obj - 1 : l.y) - b_lo + 1);
        stage_objects<NC, DIM>(vl, Yrow, p.obs, p.n_veh, i0, nI, 0, threadIdx.x, blockDim.x, fd_e, p.fd_h);
        stage_objects<NC, DIM>(vl, Yrow, p.obs, p.n_veh, a_lo, nA, nI, threadIdx.x, blockDim.x, fd_e, p.fd_h);
        stage_objects<NC, DIM>(vl, Yrow, p.obs, p.n_veh, b_lo, nB, nI + nA, threadIdx.x, blockDim.x, fd_e, p.fd_h);
    } else {
        i0 = it0; nI = it_end - it0;
        stage_objects<NC, DIM>(vl, Yrow, p.obs, p.n_veh, i0, nI, 0, threadIdx.x, blockDim.x, fd_e, p.fd_h);
    }
    __syncthreads();

    const int LR = L + p.R;
    const int TR = p.tile_rows;
    for (int g = wave; g < p.groups_per_wg; g += n_waves) {
        // this wave's group: valid lanes are [lane0, lane0 + n_valid), lane r0 + lane0 owns output
        // row `row + r0`; si/sj are the LDS slots of the lane's two curves
        int n_valid, lane0 = 0, si = 0, sj = 0, item = 0;
        size_t row;
        if (MODE == 0 && p.tiling) {
            const int i = ti0 + g;                        // group g of the tile = its row g
            if (i >= p.n_obj - 1) break;
            const int j = tj0 + lane;
            const long tri = (long)i * p.n_obj - (long)i * (i + 1) / 2 - i - 1;   // p(i,j) = tri + j
            const long pidx = tri + j;
            const bool valid = j > i && j < p.n_obj && pidx >= p.item_begin &&
                               pidx < (long)p.item_begin + p.item_count;
            const unsigned long long m = __ballot(valid);
            if (m == 0ull) continue;                      // (a later row of the tile may still be inside a partition's pair range)
            lane0 = __ffsll((long long)m) - 1;
            n_valid = __popcll(m);
            si = g;
            sj = p.groups_per_wg + (min(j, p.n_obj - 1) - tj0);
            row = (size_t)b * p.item_count + (size_t)(tri + tj0 + lane0 - p.item_begin);
        } else {
            const int itg = it0 + g * kWave;
            if (itg >= it_end) break;
            n_valid = min(kWave, it_end - itg);
            item = min(itg + lane, it_end - 1);   // idle lanes recompute the last item
            row = (size_t)b * p.item_count + (size_t)(itg - p.item_begin);
            if (MODE == 0) {
                const int2 ij = p.pairs[item];
                if (p.stage_all) { si = ij.x; sj = ij.y; }
                else { si = ij.x - i0; sj = (ij.x == i0) ? nI + (ij.y - a_lo) : nI + nA + (ij.y - b_lo); }
            } else si = p.stage_all ? item : item - i0;
        }
        const int r = lane - lane0;                     // this lane's row inside the group
        const bool mine = r >= 0 && r < n_valid;

        // ---- source curve a[q][c]
        double a[DIM][NC];
        if (MODE == 0) {
            const double* vi = vl + si * S::VP;
            const double* vj = vl + sj * S::VP;
#pragma unroll
            for (int q = 0; q < DIM; ++q)
#pragma unroll
                for (int c = 0; c < NC; ++c) a[q][c] = vi[q * NC + c] - vj[q * NC + c];
        } else {
            // Bezier.diff(): (n/T)(P_{i+1}-P_i), then elev(1) back to degree n (bezier.py:497-519)
            const double* v = vl + si * S::VP;
            const double val = (double)N / p.tf[b];
#pragma unroll
            for (int q = 0; q < DIM; ++q) {
                double t[NC];
#pragma unroll
                for (int c = 0; c < N; ++c) t[c] = v[q * NC + c] * (-val) + v[q * NC + c + 1] * val;
                a[q][0] = t[0];
                a[q][N] = t[N - 1];
#pragma unroll
                for (int c = 1; c < N; ++c)
                    a[q][c] = t[c - 1] * ((double)c / (double)N) + t[c] * ((double)(N - c) / (double)N);
            }
        }

        double cf[L];
        normsq_coeffs<NC, DIM>(a, as_ctab(p.W2), cf);

        if (!ELEV) {
            // elevMatrix(2n, 0) is the identity (bezier.py:1141-1147): the product IS the output
            if (MINONLY) {
                double m = cf[0];
#pragma unroll
                for (int k = 1; k < L; ++k) m = fmin(m, cf[k]);
                if (mine) p.out[row + r] = p.sign * m + p.offset;
            } else {
                // transpose TR rows at a time through the wave's tile
                for (int r0 = 0; r0 < n_valid; r0 += TR) {
                    if (mine && r >= r0 && r < r0 + TR) {
#pragma unroll
                        for (int k = 0; k < L; ++k) tile[(r - r0) * S::TPF + k] = p.sign * cf[k] + p.offset;
                    }
                    wave_sync();
                    flush_full<L, S::TPF>(tile, p.out, (row + r0) * L, min(TR, n_valid - r0), lane);
                    wave_sync();
                }
            }
        } else {
            if (MINONLY) {
                // elev(R) as a binomially scaled convolution (bezier.py:1127-1147 written out):
                //   out_k = (1/C(2n+R,k)) * sum_j [C(2n,j) c_j] * C(R, k-j)
                // only the minimum over k leaves the lane
                const ctab_t escale = as_ctab(p.Tt), ebin = escale + L, einv = ebin + (p.R + 2 * L - 1 + kConvPad);
                double ch[L];
#pragma unroll
                for (int j = 0; j < L; ++j) ch[j] = cf[j] * escale[j];
                double m = INFINITY;
                for (int k = 0; k < LR; ++k) {
                    const ctab_t win = ebin + k;              // win[L-1-j] = C(R, k-j)
                    double s = 0.0;
#pragma unroll
                    for (int j = 0; j < L; ++j) s = fma(ch[j], win[L - 1 - j], s);
                    m = fmin(m, s * einv[k]);
                }
                if (mine) p.out[row + r] = p.sign * m + p.offset;
            } else {
                // Full elevated rows.  The wave's pre-scaled product coefficients go to LDS (64 x L); then the lane <->
                // data mapping changes to lane = (row of a pass, block of 8 output columns): a pass covers `rpp` = 64 /
                // nb2 rows x all columns, the lane keeps the window of the binomial row its 8 columns need in registers
                // for the whole kernel and reads its row's L coefficients as LDS broadcasts -- 8 FMAs per LDS operand --
                // and a pass's rows, contiguous in the output, leave through a small tile as ONE linear run of 16-byte
                // stores.  (Earlier forms: lane = output column with 2 columns per lane was bound by the LDS pipe, 0.65 ms
                // at C5; lane = item with 32-column chunks wrote 256-byte pieces of 968-byte rows: 0.87-1.24 ms.)
                const ctab_t escale = as_ctab(p.Tt);
                const double* ebin_g = p.Tt + L;
                const double* einv_g = ebin_g + (p.R + 2 * L - 1 + kConvPad);
                const int nblk = (LR + kElevBlock - 1) / kElevBlock;
                int nb2 = 1;
                while (nb2 < nblk && nb2 < kWave) nb2 <<= 1;
                const int rpp = kWave / nb2;                       // rows per pass
                const int cb = lane & (nb2 - 1), rs = lane / nb2;
                double* chT = tile;                                // [kElevHalfRows][L]
                // output tile of a pass: row q, column block c, column i of the block at q * opitch + c * 9 + i -- the
                // pitch of 9 doubles per 8-column block keeps the 16 lanes of a row on 16 different bank pairs (at
                // pitch 8 they fall on 4: PMC showed 60 % of the LDS cycles of this loop as bank conflicts).
                // Ordinary write-back stores: a row is 8 LR bytes, not a multiple of the 128-byte line, so neighbouring
                // store instructions share lines and the L2 has to merge them -- with non-temporal stores the same loop
                // ran 0.65 instead of 0.55 ms at C5 (2.26 GB of output per launch).
                constexpr int BP = kElevBlock + 1;
                double* otile = tile + kElevHalfRows * L;          // [rpp][nb2 * 9]
                const int opitch = nb2 * BP;
                for (int cg = 0; cg < nblk; cg += nb2) {            // column groups (one unless LR > 512)
                    const int blk = cg + cb;
                    const bool colv = blk < nblk;
                    double w[L - 1 + kElevBlock], inv8[kElevBlock];
#pragma unroll
                    for (int m = 0; m < L - 1 + kElevBlock; ++m) w[m] = colv ? ebin_g[blk * kElevBlock + m] : 0.0;
#pragma unroll
                    for (int i = 0; i < kElevBlock; ++i) inv8[i] = colv ? einv_g[blk * kElevBlock + i] : 0.0;
                    const int c_lo = cg * kElevBlock, c_n = min(LR - c_lo, nb2 * kElevBlock);   // columns of this group
                    for (int hb = 0; hb < n_valid; hb += kElevHalfRows) {   // the group's rows, half a wave at a time
                    const int nh = min(kElevHalfRows, n_valid - hb);
                    if (mine && r >= hb && r < hb + kElevHalfRows) {
#pragma unroll
                        for (int j = 0; j < L; ++j) chT[(r - hb) * L + j] = cf[j] * escale[j];
                    }
                    wave_sync();
                    for (int r0 = 0; r0 < nh; r0 += rpp) {
                        const int rr = r0 + rs;
                        if (rr < nh && colv) {
                            double ch[L], sa[kElevBlock];
#pragma unroll
                            for (int j = 0; j < L; ++j) ch[j] = chT[rr * L + j];
                            conv_block1_reg<L>(w, ch, sa);
                            double* o = otile + rs * opitch + cb * BP;
#pragma unroll
                            for (int i = 0; i < kElevBlock; ++i) o[i] = p.sign * (sa[i] * inv8[i]) + p.offset;
                        }
                        wave_sync();
                        const int rows = min(rpp, nh - r0);
                        // (as 16-byte pieces -- rows of an odd length start on odd elements every other time -- 0.65 instead of 0.54 ms at C5)
                        for (int q = 0; q < rows; ++q) {            // each row: a run of c_n doubles, 512 bytes per store instruction
                            double* g = p.out + (row + hb + r0 + q) * LR + c_lo;
                            for (int kc = lane; kc < c_n; kc += kWave) g[kc] = otile[q * opitch + (kc >> 3) * BP + (kc & 7)];
                        }
                        wave_sync();
                    }
                    }
                }
            }
        }
    }
}

// =====================================================================================
//  angular rate + speed, fast path (R == 0, d == 2): shared by bern_kernels.hip and the pair sweep
// =====================================================================================
struct AngParams {
    const double* __restrict__ Y;    // [B][n_veh*2][NC]
    const double* __restrict__ tf;   // [B]
    const double* __restrict__ W2n;  // folded weights degree n,   dim factor 1   [2n+1][n+1]
    const double* __restrict__ W22n; // folded weights degree 2n,  dim factor 1   [4n+1][2n+1]
    const double* __restrict__ Wn;   // plain weights  degree n                   [2n+1][n+1]
    double* __restrict__ out;        // [B][n_veh][4n+1]           (nullable)
    double* __restrict__ out_speed;  // [B][n_veh][2n+1]           (nullable)
    int n_veh, total;                // total = B * n_veh
    double w2;                       // max_rate^2
    double sp_sign, sp_offset;       // speed output = sp_sign * |v|^2 + sp_offset
    double* __restrict__ out_speed2; // the OTHER speed bound's rows from the same curve (obtg_ctx_set_second_speed_bound;
    double sp2_sign, sp2_offset;     // optimization.py:135-169 exposes min AND max speed); nullable, needs out_speed
    int fd, fd_fixed;                // fd != 0: Y is ONE row [n_veh*2][NC]; row b >= 1 = Y with its (b-1)-th free
    double fd_h;                     //          control point advanced by fd_h (the rows obtg_fd_batch_dev writes)
};

// control points of item (b, veh) of a 2-D batch: from the materialised batch, or formed on the fly (p.fd)
template <int NC>
__device__ __forceinline__ void load_item_xy(const AngParams& p, int item, int b, double (&x)[NC], double (&y)[NC])
{
    if (!p.fd) {
        const double* src = p.Y + (size_t)item * 2 * NC;
#pragma unroll
        for (int c = 0; c < NC; ++c) { x[c] = src[c]; y[c] = src[NC + c]; }
        return;
    }
    const int veh = item - b * p.n_veh;
    const double* src = p.Y + (size_t)veh * 2 * NC;
    int pl = -1;                                           // perturbed element inside this vehicle's 2 NC values
    if (b > 0) {
        const int free_cols = NC - 2 * p.fd_fixed, kq = b - 1, pr = kq / free_cols, pc = p.fd_fixed + (kq - pr * free_cols);
        pl = pr * NC + pc - veh * 2 * NC;
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const double vx = src[c], vy = src[NC + c];
        x[c] = (c == pl) ? vx + p.fd_h : vx;
        y[c] = (NC + c == pl) ? vy + p.fd_h : vy;
    }
}

template <int NC>
__device__ __forceinline__ void diff_elev1(const double (&p)[NC], double val, double (&d)[NC])
{
    constexpr int N = NC - 1;
    double t[NC];
#pragma unroll
    for (int c = 0; c < N; ++c) t[c] = p[c] * (-val) + p[c + 1] * val;
    d[0] = t[0];
    d[N] = t[N - 1];
#pragma unroll
    for (int c = 1; c < N; ++c) d[c] = t[c - 1] * ((double)c / (double)N) + t[c] * ((double)(N - c) / (double)N);
}

// Which items a dynamics group evaluates and where its rows go (the structured finite-difference step, gjk_kernels.hip
// k_step_fd_structured: all but one vehicle of row b >= 1 repeat row 0's rows when tf[b] is tf[0]).
//   mode 0: items [item_begin + 64 group, item_end), rows written where the items are (the plain launch);
//   mode 1: item t of the group is (row t + 1, the vehicle that row advances), t < item_end; its rows go to that item's
//           place -- unless tf[t + 1] differs from tf[0] (the whole row is then a mode-0 group of its own);
//   mode 2: the group of ROW 0's vehicles, its rows copied into every row of [b0, b1) whose tf is tf[0], leaving out the
//           vehicle that row advances.  b1 - b0 <= 64.
// s_map: 64 ints of LDS, filled by dyn_emit_prepare.
struct DynEmit {
    int mode, item_begin, item_end, b0, b1;
    int* s_map;
};

__device__ __forceinline__ bool same_bits(double a, double b) { return __double_as_longlong(a) == __double_as_longlong(b); }

// item of this lane, first item and valid count of the group (mode 1: "first item" is unused)
template <int NC>
__device__ __forceinline__ int dyn_item_of_lane(const AngParams& p, const DynEmit* em, int group, int lane, int& it0, int& n_valid)
{
    if (!em) {
        it0 = group * kWave;
        n_valid = min(kWave, p.total - it0);
        return min(it0 + lane, p.total - 1);
    }
    if (em->mode == 1) {
        it0 = 0;
        n_valid = min(kWave, em->item_end - group * kWave);
        const int b = min(group * kWave + lane, em->item_end - 1) + 1;
        return b * p.n_veh + max(fd_element(p.fd, p.fd_fixed, NC, b), 0) / (2 * NC);
    }
    it0 = em->item_begin + group * kWave;
    n_valid = min(kWave, em->item_end - it0);
    return min(it0 + lane, em->item_end - 1);
}

// the first wave fills s_map; a barrier (or, for the same wave, wave_sync) must follow before dyn_emit_* reads it
template <int NC>
__device__ __forceinline__ void dyn_emit_prepare(const AngParams& p, const DynEmit& em, int item, int it0, int n_valid, int lane)
{
    if (em.mode == 1) {
        const int b = item / p.n_veh;
        em.s_map[lane] = (lane < n_valid && same_bits(p.tf[b], p.tf[0])) ? item : -1;
    } else if (em.mode == 2) {
        const int b = em.b0 + lane;
        if (b < em.b1) {
            int m = -2;
            if (same_bits(p.tf[b], p.tf[0])) {
                const int fd_e = fd_element(p.fd, p.fd_fixed, NC, b);
                const int local = fd_e >= 0 ? fd_e / (2 * NC) - it0 : -1;
                m = (local >= 0 && local < n_valid) ? local : -1;
            }
            em.s_map[lane] = m;
        }
    }
}

// rows tile[r * pitch + q], r < n_valid, q < ncol  ->  columns k0 .. k0+ncol of the rows of length LROW that em names
__device__ __forceinline__ void dyn_emit_rows(const double* __restrict__ tile, int pitch, int ncol, double* __restrict__ out,
                                              int LROW, int k0, const AngParams& p, const DynEmit& em, int it0, int n_valid,
                                              int tid, int nthr)
{
    const int total = n_valid * ncol;
    if (em.mode == 1) {
        for (int e = tid; e < total; e += nthr) {
            const int r = e / ncol, q = e - r * ncol, it = em.s_map[r];
            if (it >= 0) store_nt(out + (size_t)it * LROW + k0 + q, tile[r * pitch + q]);
        }
        return;
    }
    // the map of the range lives in one register per lane (entry j in lane j): the loop over rows reads it with
    // v_readlane, no memory access between the stores
    const int nb = em.b1 - em.b0;
    const int lane = tid & (kWave - 1);
    const int mreg = lane < nb ? em.s_map[lane] : -2;
    const size_t row_stride = (size_t)p.n_veh * LROW;
    for (int e = tid; e < total; e += nthr) {
        const int r = e / ncol, q = e - r * ncol;
        const double v = tile[r * pitch + q];
        double* o = out + ((size_t)em.b0 * p.n_veh + it0 + r) * LROW + k0 + q;
        for (int j = 0; j < nb; ++j, o += row_stride) {
            const int m = __builtin_amdgcn_readlane(mreg, j);
            if (m == -2) continue;
            if (m != r) store_nt(o, v);
        }
    }
}

// The second half of dynamics2_group for a wave that holds the coefficients K0 <= k < K1 of its side's degree-4n square:
// square, hand the other role its share through the exchange tile (row `trow` of the item), divide, leave the quotients
// in the tile.  Role 0 (denominator side) divides k < KS, role 1 (numerator side, the dearer products) the rest.  Three
// workgroup barriers, the same number for every (K0, K1): waves of one workgroup may run different instances.
template <int NC, int K0, int K1>
__device__ __forceinline__ void dyn2_tail(const int role, const double w2, const ctab_t W22n, const double (&q1)[2 * (NC - 1) + 1],
                                          double* trow)
{
    constexpr int N = NC - 1, L2 = 2 * N + 1, KS = K0 + ((K1 - K0) * 5) / 8;
    double sq[K1 - K0];
#pragma unroll
    for (int k = K0; k < K1; ++k) {
        double s = 0.0;
#pragma unroll
        for (int j = (k - 2 * N > 0 ? k - 2 * N : 0); 2 * j <= k; ++j) s = fma(W22n[k * L2 + j], q1[j] * q1[k - j], s);
        sq[k - K0] = s;
    }
    if (role == 0) {
#pragma unroll
        for (int k = KS; k < K1; ++k) trow[k] = sq[k - K0];
    } else {
#pragma unroll
        for (int k = K0; k < KS; ++k) trow[k] = sq[k - K0];
    }
    __syncthreads();
    // constraint = w^2 - num.cpts / den.cpts (optimization.py:608), each wave its share of k
    double q[K1 - K0];
    if (role == 0) {
#pragma unroll
        for (int k = K0; k < KS; ++k) q[k - K0] = w2 - trow[k] / sq[k - K0];
    } else {
#pragma unroll
        for (int k = KS; k < K1; ++k) q[k - K0] = w2 - sq[k - K0] / trow[k];
    }
    __syncthreads();
    if (role == 0) {
#pragma unroll
        for (int k = K0; k < KS; ++k) trow[k] = q[k - K0];
    } else {
#pragma unroll
        for (int k = KS; k < K1; ++k) trow[k] = q[k - K0];
    }
    __syncthreads();
}

// Two waves (threads 0..127 of the workgroup) on the 64 items of `group`; the other waves of a larger workgroup must
// have returned before the call (the barriers below count the surviving waves).  k_dynamics2 is this on its own grid.
// W4: FOUR waves on the group (the pair sweeps' grids, whose workgroups have four): wave = (role, half) -- the two roles
// as before, each wave squaring only its half of the degree-4n coefficients (k < KH or k >= KH) and dividing its share of
// that half; every element by the same operations as the two-wave form, so the bits are the same.
// HALF_SP: the speed rows leave through a 32-row tile in two halves (5.4 KB less LDS: the form the pair sweep's grid runs)
template <int NC, bool HALF_SP = false, bool W4 = false>
__device__ __forceinline__ void dynamics2_group(const AngParams& p, double* lds, const int group, const DynEmit* em = nullptr)
{
    constexpr int N = NC - 1, L2 = 2 * N + 1, L4 = 4 * N + 1;
    constexpr int KH = W4 ? (L4 + 1) / 2 : L4;                  // first coefficient of the second half
    double* tile = lds;                        // [kWave][L4]: exchange, then the output rows
    double* tile_sp = lds + kWave * L4;        // [kWave][L2]: speed rows (wave 0)
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;         // wave-uniform
    const int role = wave & 1, half = W4 ? wave >> 1 : 0;
    int it0, n_valid;
    const int item = dyn_item_of_lane<NC>(p, em, group, lane, it0, n_valid);
    const int b = item / p.n_veh;
    const bool mapped = em && em->mode != 0;   // (the 64-row speed tile only: the structured step's form)
    if (mapped && wave == 0) dyn_emit_prepare<NC>(p, *em, item, it0, n_valid, lane);
    double x[NC], y[NC];
    load_item_xy<NC>(p, item, b, x, y);
    const double val = (double)N / p.tf[b];
    double xD[NC], yD[NC];
    diff_elev1<NC>(x, val, xD);
    diff_elev1<NC>(y, val, yD);
    const ctab_t W22n = as_ctab(p.W22n);
    double q1[L2];                             // den1 (role 0) or num1 (role 1), degree 2n
    if (role == 0) {
        const ctab_t W2n = as_ctab(p.W2n);
#pragma unroll
        for (int k = 0; k < L2; ++k) {
            double sd = 0.0;
#pragma unroll
            for (int j = (k - N > 0 ? k - N : 0); 2 * j <= k; ++j)
                sd = fma(W2n[k * NC + j], fma(xD[j], xD[k - j], yD[j] * yD[k - j]), sd);
            q1[k] = sd;
        }
        for (int which = 0; which < ((p.out_speed && wave == 0) ? (p.out_speed2 ? 2 : 1) : 0); ++which) {
            double* const dst = which ? p.out_speed2 : p.out_speed;
            const double sgn = which ? p.sp2_sign : p.sp_sign, off = which ? p.sp2_offset : p.sp_offset;
            if (which) wave_sync();
            if (!HALF_SP) {
#pragma unroll
                for (int k = 0; k < L2; ++k) tile_sp[lane * L2 + k] = sgn * q1[k] + off;
                wave_sync();
                if (mapped) dyn_emit_rows(tile_sp, L2, L2, dst, L2, 0, p, *em, it0, n_valid, lane, kWave);
                else flush_full<L2, L2>(tile_sp, dst, (size_t)it0 * L2, n_valid, lane);
            } else {
                for (int h0 = 0; h0 < n_valid; h0 += kWave / 2) {
                    if (lane >= h0 && lane < h0 + kWave / 2) {
#pragma unroll
                        for (int k = 0; k < L2; ++k) tile_sp[(lane - h0) * L2 + k] = sgn * q1[k] + off;
                    }
                    wave_sync();
                    flush_full<L2, L2>(tile_sp, dst, ((size_t)it0 + h0) * L2, min(kWave / 2, n_valid - h0), lane);
                    wave_sync();
                }
            }
        }
    } else {
        const ctab_t Wn = as_ctab(p.Wn);
        double xDD[NC], yDD[NC];
        diff_elev1<NC>(xD, val, xDD);
        diff_elev1<NC>(yD, val, yDD);
#pragma unroll
        for (int k = 0; k < L2; ++k) {
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int j = (k - N > 0 ? k - N : 0); j <= (N < k ? N : k); ++j) {
                const double wkj = Wn[k * NC + j];
                s1 = fma(wkj, yDD[j] * xD[k - j], s1);
                s2 = fma(wkj, xDD[j] * yD[k - j], s2);
            }
            q1[k] = s1 - s2;
        }
    }
    // the square (degree 4n) of either side, this wave's coefficients; then the quotient through the exchange tile
    // (the coefficient range is a template argument: register arrays want constant indices)
    if (!W4) dyn2_tail<NC, 0, L4>(role, p.w2, W22n, q1, tile + lane * L4);
    else if (half == 0) dyn2_tail<NC, 0, KH>(role, p.w2, W22n, q1, tile + lane * L4);
    else dyn2_tail<NC, (W4 ? KH : 0), L4>(role, p.w2, W22n, q1, tile + lane * L4);
    constexpr int NT = (W4 ? 4 : 2) * kWave;
    if (mapped) {
        dyn_emit_rows(tile, L4, L4, p.out, L4, 0, p, *em, it0, n_valid, (int)threadIdx.x, NT);
        return;
    }
    // rows of consecutive items are contiguous in the output: one linear copy by all waves
    const size_t grow = (size_t)it0 * L4;
    const int total = n_valid * L4;
    const int shift = (int)(grow & 1);
    const int npairs = (total + shift + 1) >> 1;
    for (int m = threadIdx.x; m < npairs; m += NT) {
        const int e0 = 2 * m - shift, e1 = e0 + 1;
        if (e0 >= 0 && e1 < total) store_nt2(p.out + grow + e0, tile[e0], tile[e1]);
        else if (e0 >= 0) store_nt(p.out + grow + e0, tile[e0]);
        else if (e1 < total) store_nt(p.out + grow + e1, tile[e1]);
    }
}

#ifndef OBTG_DYN_ELEV_WAVES
#define OBTG_DYN_ELEV_WAVES 2
#endif
struct AngElevParams {
    AngParams a;
    const double* __restrict__ cv4;  // scale[4n+1] = C(4n, j) | padded row C(4R, m) 2^-e, m = -(4n) .. 4R+4n+8
    const double* __restrict__ cv2;  // scale[2n+1] = C(2n, j) | padded row C(R, m), m = -(2n) .. R+2n+8 | 1/C(2n+R, k) (+8)
    int R;
};

template <int NC>
__device__ __forceinline__ void dynamics_elev_group(const AngElevParams& q, double* lds, const int group, const DynEmit* em = nullptr)
{
    constexpr int N = NC - 1, L2 = 2 * N + 1, L4 = 4 * N + 1, TP = kElevChunk + 1;
    const AngParams& p = q.a;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: table reads become scalar loads
    double* tile = lds + wave * (kWave * TP);
    int it0, n_valid;
    const int item = dyn_item_of_lane<NC>(p, em, group, lane, it0, n_valid);
    const int b = item / p.n_veh;
    const bool mapped = em && em->mode != 0;
    if (mapped) {
        if (wave == 0) dyn_emit_prepare<NC>(p, *em, item, it0, n_valid, lane);
        __syncthreads();
    }
    double num[L4], den[L4];
    {   // ---- phase A: degree-4n numerator and denominator from the original control points (as k_dynamics2)
        double x[NC], y[NC];
        load_item_xy<NC>(p, item, b, x, y);
        const double val = (double)N / p.tf[b];
        double xD[NC], yD[NC], xDD[NC], yDD[NC];
        diff_elev1<NC>(x, val, xD);
        diff_elev1<NC>(y, val, yD);
        diff_elev1<NC>(xD, val, xDD);
        diff_elev1<NC>(yD, val, yDD);
        const ctab_t Wn = as_ctab(p.Wn), W2n = as_ctab(p.W2n), W22n = as_ctab(p.W22n);
        double num1[L2], den1[L2];
#pragma unroll
        for (int k = 0; k < L2; ++k) {
            double s1 = 0.0, s2 = 0.0, sd = 0.0;
#pragma unroll
            for (int j = (k - N > 0 ? k - N : 0); j <= (N < k ? N : k); ++j) {
                const double wkj = Wn[k * NC + j];
                s1 = fma(wkj, yDD[j] * xD[k - j], s1);
                s2 = fma(wkj, xDD[j] * yD[k - j], s2);
            }
#pragma unroll
            for (int j = (k - N > 0 ? k - N : 0); 2 * j <= k; ++j)
                sd = fma(W2n[k * NC + j], fma(xD[j], xD[k - j], yD[j] * yD[k - j]), sd);
            num1[k] = s1 - s2;
            den1[k] = sd;
        }
        // speed rows = elev(den1, R): this wave's share of their 32-column chunks, while den1 is still live
        if (p.out_speed) {
            const int L2R = L2 + q.R;
            const ctab_t sc2 = as_ctab(q.cv2), row2 = sc2 + L2, inv2 = row2 + (q.R + 1 + 2 * (L2 - 1) + kElevBlock);
            double dh[L2];
#pragma unroll
            for (int j = 0; j < L2; ++j) dh[j] = sc2[j] * den1[j];
            for (int k0 = wave * kElevChunk; k0 < L2R; k0 += 4 * kElevChunk) {
                const int kc = min(kElevChunk, L2R - k0);
                // one pass per requested bound (the second one, obtg_ctx_set_second_speed_bound, repeats the chunk's
                // convolution: the degree-2n curve is what the two share)
                for (int which = 0; which < (p.out_speed2 ? 2 : 1); ++which) {
                    const double sgn = which ? p.sp2_sign : p.sp_sign, off = which ? p.sp2_offset : p.sp_offset;
                    for (int kb = 0; kb < kc; kb += kElevBlock) {
                        double sa[kElevBlock];
                        conv_block1<L2>(row2 + k0 + kb, dh, sa);
#pragma unroll
                        for (int i = 0; i < kElevBlock; ++i)
                            tile[lane * TP + kb + i] = sgn * (sa[i] * inv2[k0 + kb + i]) + off;
                    }
                    wave_sync();
                    if (mapped) dyn_emit_rows(tile, TP, kc, which ? p.out_speed2 : p.out_speed, L2R, k0, p, *em, it0, n_valid, lane, kWave);
                    else elev_store_chunk<L2>(tile, which ? p.out_speed2 : p.out_speed, (size_t)it0 * L2R, L2R, k0, kc, n_valid, lane);
                    wave_sync();
                }
            }
        }
        const ctab_t sc4 = as_ctab(q.cv4);
#pragma unroll
        for (int k = 0; k < L4; ++k) {
            double sn = 0.0, sd = 0.0;
#pragma unroll
            for (int j = (k - 2 * N > 0 ? k - 2 * N : 0); 2 * j <= k; ++j) {
                const double wkj = W22n[k * L2 + j];
                sn = fma(wkj, num1[j] * num1[k - j], sn);
                sd = fma(wkj, den1[j] * den1[k - j], sd);
            }
            const double sck = sc4[k];          // C(4n, k): the convolution form's pre-scaling
            num[k] = sck * sn;
            den[k] = sck * sd;
        }
    }
    // ---- phase B: elevate both by 4R and divide, 32 output columns at a time; chunk t belongs to wave t mod 4
    const int L4R = L4 + 4 * q.R;
    const ctab_t row4 = as_ctab(q.cv4) + L4;
    for (int k0 = wave * kElevChunk; k0 < L4R; k0 += 4 * kElevChunk) {
        const int kc = min(kElevChunk, L4R - k0);
        for (int kb = 0; kb < kc; kb += kElevBlock) {
            double sn[kElevBlock], sd[kElevBlock];
            conv_block2<L4>(row4 + k0 + kb, num, den, sn, sd);
#pragma unroll
            for (int i = 0; i < kElevBlock; ++i) tile[lane * TP + kb + i] = p.w2 - sn[i] / sd[i];
        }
        wave_sync();
        if (mapped) dyn_emit_rows(tile, TP, kc, p.out, L4R, k0, p, *em, it0, n_valid, lane, kWave);
        else elev_store_chunk<L4>(tile, p.out, (size_t)it0 * L4R, L4R, k0, kc, n_valid, lane);
        wave_sync();
    }
}

}  // namespace obtg
