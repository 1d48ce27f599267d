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
    const double* __restrict__ Tf;    // the same matrix as v_mfma_f64_16x16x4 B fragments (tables.cpp elev_table_frag)
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
    int fd, fd_fixed;                 // fd != 0: Y is ONE row; batch row g >= 1 = Y with its (g-1)-th free control point advanced by
    double fd_h;                      //          fd_h (obtg_fd_batch_dev's rows), formed while staging; local row b is batch row
                                      //          b + fd - 1 (fd = 1 + first row of the range: obtg_fd_view_begin_rows)
    int sel_k;                        // MINONLY kernels: 0 = the item's minimum; 1..4 = its sel_k SMALLEST values, ascending
    int* __restrict__ sel_idx;        //   (obtg_temporal_sep_active): out[item][sel_k], sel_idx[item][sel_k] = their columns (nullable)
};

// The four smallest of a stream of values with their positions, ascending in value while they are collected; ties keep
// the earlier position first (the order numpy's stable argsort gives).  Branch-free insertion: v[] is sorted, so x < v[j] implies x < v[j + 1].
struct Smallest4 {
    double v[4];
    int i[4];
    __device__ __forceinline__ Smallest4()
    {
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = INFINITY; i[j] = -1; }
    }
    __device__ __forceinline__ void put(double x, int k)
    {
        const bool l0 = x < v[0], l1 = x < v[1], l2 = x < v[2], l3 = x < v[3];
        v[3] = l2 ? v[2] : (l3 ? x : v[3]); i[3] = l2 ? i[2] : (l3 ? k : i[3]);
        v[2] = l1 ? v[1] : (l2 ? x : v[2]); i[2] = l1 ? i[1] : (l2 ? k : i[2]);
        v[1] = l0 ? v[0] : (l1 ? x : v[1]); i[1] = l0 ? i[0] : (l1 ? k : i[1]);
        v[0] = l0 ? x : v[0];               i[0] = l0 ? k : i[0];
    }
    // the n smallest to out[base .. base + n), positions to idx (nullable) -- in ascending POSITION, not ascending value:
    // a row then follows ONE control point for as long as the set's membership stands, and two members whose values cross
    // do not swap rows (in value order every such crossing is a kink in both rows; SLSQP's quasi-Newton update sees each)
    __device__ __forceinline__ void store(double* out, int* idx, size_t base, int n) const
    {
        double w[4];
        int q[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { w[j] = v[j]; q[j] = j < n ? i[j] : 0x7fffffff; }   // the others sort behind the n
        auto cswap = [&](int a, int b) {
            const bool s = q[b] < q[a];
            const int qa = s ? q[b] : q[a], qb = s ? q[a] : q[b];
            const double wa = s ? w[b] : w[a], wb = s ? w[a] : w[b];
            q[a] = qa; q[b] = qb; w[a] = wa; w[b] = wb;
        };
        cswap(0, 1); cswap(2, 3); cswap(0, 2); cswap(1, 3); cswap(1, 2);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < n) { out[base + j] = w[j]; if (idx) idx[base + j] = q[j]; }
    }
};

// element of an evaluation row [n_rows][nc] that LOCAL row b of a virtual finite-difference batch advances (-1: none);
// fd - 1 is the batch row of local row 0
__device__ __forceinline__ int fd_element(int fd, int fixed, int nc, int b)
{
    if (!fd || b + fd - 1 <= 0) return -1;
    const int free_cols = nc - 2 * fixed, kq = b + fd - 2, pr = kq / free_cols;
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

// (d/2) * sum_q a_q^2 as Bernstein coefficients c[0..L) (bezier.py:884, 1183-1208).  The product weights
// C(n,j) C(n,k-j) / C(2n,k) are separable: with u_j = C(n,j) a_j (the binomials are exact in binary64),
//   c_k = S_k (2 sum_{j < k-j} u_j u_{k-j} + [k even] u_{k/2}^2),   S_k = (d/2) / C(2n,k)   (odd k: the 2 is folded into S_k)
// -- two instructions per term and dimension instead of three (186 instead of 220 per pair at n = 10, d = 2).  C(n, .) and
// S sit behind the folded weights in the same table (tables.cpp folded_product_weights).
template <int NC, int DIM>
__device__ __forceinline__ void normsq_coeffs(const double (&a)[DIM][NC], ctab_t W2,
                                              double (&c)[2 * NC - 1])
{
    constexpr int N = NC - 1, L = 2 * N + 1;
#ifdef OBTG_NORMSQ_FOLDED_WEIGHTS        // the form before round 4 (A/B builds, tools/build_variant.sh): one weight per folded term
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
    return;
#endif
    const ctab_t Cn = W2 + L * NC, Sk = Cn + NC;
    double u[DIM][NC];
#pragma unroll
    for (int q = 0; q < DIM; ++q)
#pragma unroll
        for (int j = 0; j < NC; ++j) u[q][j] = Cn[j] * a[q][j];
#pragma unroll
    for (int k = 0; k < L; ++k) {
        const int jlo = k - N > 0 ? k - N : 0;
        double so = 0.0;
#pragma unroll
        for (int j = jlo; 2 * j < k; ++j)
#pragma unroll
            for (int q = 0; q < DIM; ++q) so = (j == jlo && q == 0) ? u[q][j] * u[q][k - j] : fma(u[q][j], u[q][k - j], so);
        double s = so;
        if ((k & 1) == 0) {
            const int h = k >> 1;
            double dg = u[0][h] * u[0][h];
#pragma unroll
            for (int q = 1; q < DIM; ++q) dg = fma(u[q][h], u[q][h], dg);
            s = (2 * jlo < k) ? fma(2.0, so, dg) : dg;
        }
        c[k] = Sk[k] * s;
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

// Keep a value that came from memory in its register, waited for HERE: a load left pending across a loop's back edge makes
// the first use in every iteration an s_waitcnt vmcnt(0) -- which also waits for every store the wave has issued.
__device__ __forceinline__ void pin_reg(double& x) { asm volatile("" : "+v"(x)); }

constexpr int kElevBlock = 8;          // zero entries behind the padded binomial rows of tables.cpp elev_conv_padded

// ---- degree elevation as a matrix product on v_mfma_f64_16x16x4_f64 ------------------------------------------------
//     out[row][k] = offset + sum_j a[row][j] T[j][k],   T = elevMatrix(LIN-1, R) (bezier.py:1127-1147), a = sign * coefficients
// At R = 100 the matrix is 83 % dense (2121 of 2541 entries) and the batch kernels multiply [64 rows x LIN] blocks by it
// on the matrix instruction: one instruction replaces 16 v_fma_f64 and the LDS broadcasts that fed them, and keeps the
// FP64 units busy from ONE wave per SIMD (a wave alone issues a v_fma_f64 every 9 clocks, two waves one every 4.75;
// tools/mfma_f64_probe.hip -- which also shows that the instruction takes 64 clocks, the time of its 16 v_fma_f64, and
// that f64 vector work of another wave on the SIMD stands still meanwhile: it runs ON the FP64 vector units, so what it
// buys is issue slots and operand traffic, not a second pipe).  The instruction is, bit for bit, a chain of fused
// multiply-adds in ascending k from its C operand (the same probe: 12 800 of 12 800 results), so the lane-per-item forms
// (elev_at below: k_tsep_fd, the minima, the structured step's fix-up rows) reproduce the batch exactly by running that
// chain over a row of the dense table; zero entries of the band and of the padding add nothing (fma(x, 0, s) = s).
typedef double v4d_t __attribute__((ext_vector_type(4)));

template <int LIN>
struct ElevMfma {
    static constexpr int KS = (LIN + 3) / 4;          // k-steps of one output tile
    static constexpr int PA = 4 * KS + 2;             // row pitch of the coefficient image: PA / 2 odd => the 32 lanes of a
                                                      // ds_read_b64 group (16 rows x 2 k) fall on 32 different bank pairs
    static constexpr int NTG = KS <= 6 ? 8 : 4;       // output tiles (16 columns each) per column group: B fragments and
    static constexpr int CW = 16 * NTG;               // accumulators of a group live in registers (NTG * KS, NTG * 4 doubles)
};

// per-wave LDS of elev_rows_mfma: the coefficient image [64][PA], then (same place) the output tile [16][cw]
__host__ __device__ constexpr int elev_mfma_wave_doubles(int LIN, int R)
{
    const int KS = (LIN + 3) / 4, PA = 4 * KS + 2, CW = 16 * (KS <= 6 ? 8 : 4);
    const int cw = LIN + R < CW ? LIN + R : CW;
    return kWave * PA > 16 * cw ? kWave * PA : 16 * cw;
}

// one row in one lane: out_k = offset + sum_j a_j T[j][k] as the matrix instruction forms it; Tdk = row k of the dense
// transposed table (wave-uniform k: scalar loads)
template <int LIN>
__device__ __forceinline__ double elev_at(const double (&a)[LIN], const ctab_t Tdk, const double offset)
{
    double s = offset;
#pragma unroll
    for (int j = 0; j < LIN; ++j) s = fma(a[j], Tdk[j], s);
    return s;
}

// B fragments of the column group that starts at output tile t0.  Past the last tile the last one is read again: those
// accumulators are never stored.  One wave-uniform base, the lane's offset, constant displacements: the loads need one
// address register between them, not one pair each.
template <int LIN>
__device__ __forceinline__ void elev_load_bfrag(const double* __restrict__ Tf, const int t0, const int NT, const int lane,
                                                double (&bfr)[ElevMfma<LIN>::NTG][ElevMfma<LIN>::KS])
{
    using E = ElevMfma<LIN>;
#pragma unroll
    for (int t = 0; t < E::NTG; ++t) {
        const double* ft = Tf + (size_t)__builtin_amdgcn_readfirstlane(min(t0 + t, NT - 1)) * (E::KS * kWave);
#pragma unroll
        for (int s = 0; s < E::KS; ++s) bfr[t][s] = ft[s * kWave + lane];
    }
}

// one output tile: acc = offset + A-tile x B-fragments of n-tile t; lane l holds rows (l >> 4) + 4 r, column 16 t + (l & 15).
// (a chain of dependent instructions issues at the rate of independent ones -- 64 clocks either way, mfma_f64_probe --
// so one accumulator per tile is enough and a 16-row tile needs 8 registers of them, not 8 per column tile)
template <int LIN>
__device__ __forceinline__ v4d_t elev_mfma_tile(const double (&afr)[ElevMfma<LIN>::KS], const double (&bfr)[ElevMfma<LIN>::KS],
                                                const double offset)
{
    using E = ElevMfma<LIN>;
    v4d_t acc;
    acc[0] = offset; acc[1] = offset; acc[2] = offset; acc[3] = offset;
#ifdef OBTG_EXP_MFMA_STEPS     // TIMING builds only (tools/build_variant.sh): a truncated chain, WRONG results -- what the launch costs with fewer matrix instructions
#pragma unroll
    for (int s = 0; s < (E::KS < OBTG_EXP_MFMA_STEPS ? E::KS : OBTG_EXP_MFMA_STEPS); ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[s], bfr[s], acc, 0, 0, 0);
#else
#pragma unroll
    for (int s = 0; s < E::KS; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[s], bfr[s], acc, 0, 0, 0);
#endif
    return acc;
}

// A fragments of rows 16 m .. 16 m + 15 of a coefficient image [64][PA] in LDS
template <int LIN>
__device__ __forceinline__ void elev_load_afrag(const double* img, const int m, const int lane, double (&afr)[ElevMfma<LIN>::KS])
{
    using E = ElevMfma<LIN>;
    const double* rowp = img + (16 * m + (lane & 15)) * E::PA + (lane >> 4);
#pragma unroll
    for (int s = 0; s < E::KS; ++s) afr[s] = rowp[4 * s];
}

// this lane's row of the coefficient image (zero padded to 4 KS entries)
template <int LIN>
__device__ __forceinline__ void elev_store_image_row(double* img, const int r, const double (&a)[LIN])
{
    using E = ElevMfma<LIN>;
#pragma unroll
    for (int j = 0; j < 4 * E::KS; ++j) img[r * E::PA + j] = j < LIN ? a[j] : 0.0;
}

// the accumulator of n-tile t of a 16-row tile -> LDS tile [16][cw] (the output run itself when cw is the whole row)
__device__ __forceinline__ void elev_acc_to_tile(const v4d_t acc, const int t, double* tile, const int cw, const int lane)
{
    const int c = 16 * t + (lane & 15), r0 = lane >> 4;
    if (c < cw) {
#pragma unroll
        for (int r = 0; r < 4; ++r) tile[(r0 + 4 * r) * cw + c] = acc[r];
    }
}

// `total` doubles of an LDS run -> gout[e0 ..): 16-byte stores, 16-byte aligned in memory; the LDS reads go out four at
// a time (one read, one wait, one store per trip leaves the loop bound by the LDS latency)
template <bool NT>
__device__ __forceinline__ void store_run(const double* __restrict__ tile, double* __restrict__ gout, const size_t e0, const int total,
                                          const int lane)
{
    const int shift = (int)(e0 & 1);
    if (shift == 0 && (total & 1) == 0 && (reinterpret_cast<size_t>(tile) & 15) == 0) {
        const d2_t* t2 = reinterpret_cast<const d2_t*>(tile);
        d2_t* g2 = reinterpret_cast<d2_t*>(gout + e0);
        const int np = total >> 1;
        for (int m0 = lane; m0 < np; m0 += 4 * kWave) {
            d2_t v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) if (m0 + u * kWave < np) v[u] = t2[m0 + u * kWave];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (m0 + u * kWave < np) { if (NT) __builtin_nontemporal_store(v[u], g2 + m0 + u * kWave); else g2[m0 + u * kWave] = v[u]; }
        }
        return;
    }
    const int npairs = (total + shift + 1) >> 1;
    for (int m = lane; m < npairs; m += kWave) {
        const int a0 = 2 * m - shift, a1 = a0 + 1;
        if (a0 >= 0 && a1 < total) {
            if (NT) store_nt2(gout + e0 + a0, tile[a0], tile[a1]);
            else { d2_t v; v.x = tile[a0]; v.y = tile[a1]; *reinterpret_cast<d2_t*>(gout + e0 + a0) = v; }
        } else if (a0 >= 0) gout[e0 + a0] = tile[a0];
        else if (a1 < total) gout[e0 + a1] = tile[a1];
    }
}

// The elevated rows of ONE wave's group: lane r < n_valid holds row r's coefficients a[] (sign applied); rows are
// contiguous in the output from element `e_row0` on, LR = LIN + R doubles each.  wl: elev_mfma_wave_doubles of LDS.
// Four 16-row tiles; per tile NT x KS matrix instructions, the accumulators through the LDS tile -- the tile of a
// whole-row group (LR <= CW) IS the output run of its 16 rows, 128-byte aligned whenever the output is -- and out as
// linear 16-byte-per-lane stores.
template <int LIN>
__device__ __forceinline__ void elev_rows_mfma(const double (&a)[LIN], const bool mine, const int r, const double* __restrict__ Tf,
                                               const int LR, const double offset, double* wl, double* __restrict__ gout,
                                               const size_t e_row0, const int n_valid, const int lane)
{
    using E = ElevMfma<LIN>;
    if (mine) elev_store_image_row<LIN>(wl, r, a);
    wave_sync();
    double afr[4][E::KS];
#pragma unroll
    for (int m = 0; m < 4; ++m) elev_load_afrag<LIN>(wl, m, lane, afr[m]);
    wave_sync();                                       // the image is dead: its place becomes the output tile
    const int NT = (LR + 15) >> 4;
    for (int t0 = 0; t0 < NT; t0 += E::NTG) {
        const int c0 = 16 * t0, cw = min(LR - c0, E::CW);
        double bfr[E::NTG][E::KS];
        elev_load_bfrag<LIN>(Tf, t0, NT, lane, bfr);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int rows = min(16, n_valid - 16 * m);        // wave-uniform
            if (rows > 0) {
#pragma unroll
                for (int t = 0; t < E::NTG; ++t) elev_acc_to_tile(elev_mfma_tile<LIN>(afr[m], bfr[t], offset), t, wl, cw, lane);
                wave_sync();
                if (cw == LR) store_run<false>(wl, gout, e_row0 + (size_t)(16 * m) * LR, rows * LR, lane);
                else {
                    for (int q = 0; q < rows; ++q)             // a run of cw doubles per row
                        for (int kc = lane; kc < cw; kc += kWave) gout[e_row0 + (size_t)(16 * m + q) * LR + c0 + kc] = wl[q * cw + kc];
                }
                wave_sync();
            }
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
    const double* __restrict__ Td;    // DEG_ELEV > 0 (structured step): the elevation matrix, dense rows and matrix-instruction
    const double* __restrict__ Tf;    // fragments (NsParams::Td, ::Tf)
    int R;
    int n_veh, obs_shift;             // pairs may name point obstacles (object ids >= n_veh, optimization.py:86-98): those are
                                      // staged obs_shift slots further on (behind the polygons of the hull sweep); 0, 0: none
    int n_tobj;                       // objects of the separation pair table (vehicles + point obstacles); 0: n_veh of the caller
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
    const int n_to = t.n_tobj > 0 ? t.n_tobj : n_veh;                   // partners: the other vehicles and the point obstacles
    for (int u0 = threadIdx.x; u0 < n_to - 1; u0 += blockDim.x) {
        const int u = u0 < v ? u0 : u0 + 1;
        const int i = min(u, v), j = max(u, v);
        const int q = i * (2 * n_to - i - 1) / 2 + (j - i - 1);         // position of (i, j) in the lexicographic pair list
        const double2* vi = xy + tsep_slot(t, i) * vpq;
        const double2* vj = xy + tsep_slot(t, j) * vpq;
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

// ---- Structured finite-difference step with DEG_ELEV = R > 0 (gjk_kernels.hip k_step_fd_structured<NC, true>) -----------------
// Both kinds that write separation rows (S: row 0's groups streamed into row ranges; F: a perturbed row's own pairs) form
// their elevated rows as sep_elev_coop_body does: the workgroup's four waves share each 16-row tile, wave w owning column
// tiles w and w + 4 of the 128-column group with its B fragments stationary in 2 KS registers, accumulators -> an LDS tile
// [16][cw].  (Round 3/4's form -- one wave per 16-row tile with all eight column tiles' fragments, 32 elements per lane of
// the tile kept in registers with row and pair of each -- needed 250 VGPRs and spilled: two workgroups per CU for the whole
// launch, and 8-byte stores.  This one stays below the dynamics groups' 168.)
template <int L>
__device__ __forceinline__ void coop_load_bfrag(const double* __restrict__ Tf, const int NT, const int cg, const int wave, const int lane,
                                                double (&bfr)[2][ElevMfma<L>::KS])
{
    using E = ElevMfma<L>;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const double* f = Tf + (size_t)min(8 * cg + wave + 4 * i, NT - 1) * (E::KS * kWave);
#pragma unroll
        for (int sx = 0; sx < E::KS; ++sx) bfr[i][sx] = f[sx * kWave + lane];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int sx = 0; sx < E::KS; ++sx) pin_reg(bfr[i][sx]);
}

// rows 16 m .. 16 m + 15 of the coefficient image x column group cg -> tile [16][cw] (this wave's two column tiles of it)
template <int L>
__device__ __forceinline__ void coop_tile(const double* img, const int m, const double (&bfr)[2][ElevMfma<L>::KS], const double offset,
                                          double* tile, const int cw, const int cg, const int NT, const int wave, const int lane)
{
    using E = ElevMfma<L>;
    double afr[E::KS];
    elev_load_afrag<L>(img, m, lane, afr);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int tl = wave + 4 * i;
        if (8 * cg + tl < NT) elev_acc_to_tile(elev_mfma_tile<L>(afr, bfr[i], offset), tl, tile, cw, lane);
    }
}

// product coefficients (sign applied) of the pair of objects at LDS slots si, sj -> row r of the coefficient image
template <int NC>
__device__ __forceinline__ void tsep_image_row(const TsepXYParams& t, const double2* xy, const int vpq, const int si, const int sj,
                                               double* img, const int r)
{
    constexpr int L = 2 * NC - 1;
    const double2* vi = xy + si * vpq;
    const double2* vj = xy + sj * vpq;
    double a[2][NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const double2 pi = vi[c], pj = vj[c];
        a[0][c] = pi.x - pj.x;
        a[1][c] = pi.y - pj.y;
    }
    double cf[L];
    normsq_coeffs<NC, 2>(a, as_ctab(t.W2), cf);
#pragma unroll
    for (int j = 0; j < L; ++j) cf[j] *= t.sign;
    elev_store_image_row<L>(img, r, cf);
}

// S: the 64-pair group g of the staged row, elevated tile by tile and streamed from registers (16-byte pieces, 256 threads:
// four pieces each per tile) into every batch row b0 <= b < b1, leaving out the rows of pairs that contain batch row b's own
// vehicle (fd_element) -- those are the F workgroups'.  img: [64][PA], tile: [16][min(LR, 128)] doubles of LDS, disjoint.
template <int NC>
__device__ __forceinline__ void tsep_elev_group_stream(const TsepXYParams& t, const double2* xy, const int vpq, const int g,
                                                       double* img, double* tile, const int b0, const int b1,
                                                       const int fd, const int fd_fixed)
{
    constexpr int L = 2 * NC - 1;
    using E = ElevMfma<L>;
    const int LR = L + t.R, NT = (LR + 15) >> 4;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int itg = g * kWave;
    const int n_valid = min(kWave, t.n_pairs - itg);
    __shared__ unsigned s_rowpair[kWave];                              // the group's pairs, i | j << 16 (0xffffffff: no such row)
    if (wave == 0) {
        const int2 ij = t.pairs[min(itg + lane, t.n_pairs - 1)];      // (lanes beyond the last pair: a valid pair, never streamed)
        tsep_image_row<NC>(t, xy, vpq, ij.x, ij.y, img, lane);
        s_rowpair[lane] = lane < n_valid ? ((unsigned)ij.x | ((unsigned)ij.y << 16)) : 0xffffffffu;
    }
    __syncthreads();
    const int n_tiles = (n_valid + 15) >> 4;
    constexpr int kSlots = 4;                                          // 16 x 128 doubles = 1024 pieces over 256 threads
#ifndef OBTG_STRUCT_S_TILE_MAJOR
    if (NT <= 8) {
        // Rows of up to 128 columns (one column group): all four tiles of the group wait in registers and the stream goes batch
        // row by batch row -- the workgroup writes its 64 pairs' rows (up to 62 KB, contiguous) of batch row b, while the
        // workgroups of the neighbouring groups write theirs: together one dense run per batch row, which is what the memory
        // system rewards (tools/store_pattern_probe2.hip: a dense advancing write front).  Tile after tile across the batch
        // rows, as below (and with -DOBTG_STRUCT_S_TILE_MAJOR, the A/B build), every workgroup leaves a quarter of each row for later:
        // C5 0.545 against 0.528 ms over four interleaved pairs of runs on one box.
        double bfr[2][E::KS];
        coop_load_bfrag<L>(t.Tf, NT, 0, wave, lane, bfr);
        double v0[4][kSlots], v1[4][kSlots];
        int qrow[4][kSlots];                                           // row (inside its tile) of the piece's first element | 16 if the second one is the next row's
        unsigned myp[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int rows = m < n_tiles ? min(16, n_valid - 16 * m) : 0;      // (workgroup-uniform)
            if (m < n_tiles) coop_tile<L>(img, m, bfr, t.offset, tile, LR, 0, NT, wave, lane);
            __syncthreads();
            const int n_el = rows * LR;
#pragma unroll
            for (int sx = 0; sx < kSlots; ++sx) {
                const int e0 = 2 * (tid + 256 * sx), e1 = e0 + 1;
                v0[m][sx] = e0 < n_el ? tile[e0] : 0.0;
                v1[m][sx] = e1 < n_el ? tile[e1] : 0.0;
                const int q0 = e0 / LR;
                qrow[m][sx] = q0 | ((e1 - q0 * LR >= LR) ? 16 : 0);
            }
            myp[m] = s_rowpair[16 * m + (lane & 15)];
            __syncthreads();                                          // the tile is free for the next one
        }
        for (int b = b0; b < b1; ++b) {
            const int fd_e = fd_element(fd, fd_fixed, NC, b);
            const int vb = fd_e >= 0 ? fd_e / (2 * NC) : -1;
            const size_t ob = ((size_t)b * t.n_pairs + (size_t)itg) * LR;
            double* gp = t.out + ob;
            const bool aligned = (ob & 1) == 0;                       // (16 LR doubles per tile: every tile of the run starts as the run does)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (m >= n_tiles) break;
                const int n_el = min(16, n_valid - 16 * m) * LR;
                double* gm = gp + m * 16 * LR;
                const bool touched = __builtin_amdgcn_ballot_w64((int)(myp[m] & 0xffffu) == vb || (int)(myp[m] >> 16) == vb) != 0;
#pragma unroll
                for (int sx = 0; sx < kSlots; ++sx) {
                    const int e0 = 2 * (tid + 256 * sx), e1 = e0 + 1;
                    bool w0 = e0 < n_el, w1 = e1 < n_el;
                    if (touched) {                                    // (the pieces' pairs: looked up in LDS, only in the batch rows that need it)
                        const int q0 = qrow[m][sx] & 15, q1 = min(15, q0 + (qrow[m][sx] >> 4));
                        const unsigned pa_ = s_rowpair[16 * m + q0], pb_ = s_rowpair[16 * m + q1];
                        w0 = w0 && (int)(pa_ & 0xffffu) != vb && (int)(pa_ >> 16) != vb;
                        w1 = w1 && (int)(pb_ & 0xffffu) != vb && (int)(pb_ >> 16) != vb;
                    }
                    if (w0 && w1 && aligned) store_nt2(gm + e0, v0[m][sx], v1[m][sx]);
                    else {
                        if (w0) store_nt(gm + e0, v0[m][sx]);
                        if (w1) store_nt(gm + e1, v1[m][sx]);
                    }
                }
            }
        }
        return;
    }
#endif
    for (int cg = 0; 8 * cg < NT; ++cg) {
        const int c0 = 128 * cg, cw = min(LR - c0, 128);
        double bfr[2][E::KS];
        coop_load_bfrag<L>(t.Tf, NT, cg, wave, lane, bfr);            // (waited for here: no load is pending while the stores stream)
        for (int m = 0; m < n_tiles; ++m) {
            const int rows = min(16, n_valid - 16 * m);
            coop_tile<L>(img, m, bfr, t.offset, tile, cw, cg, NT, wave, lane);
            __syncthreads();
            const int n_el = rows * cw;
            double v0[kSlots], v1[kSlots];
            int off0[kSlots], off1[kSlots], pr0[kSlots], pr1[kSlots];  // offsets in the batch row's run; (i | j << 16) of each element's pair, -1: none
#pragma unroll
            for (int sx = 0; sx < kSlots; ++sx) {
                const int e0 = 2 * (tid + 256 * sx), e1 = e0 + 1;
                v0[sx] = v1[sx] = 0.0; pr0[sx] = pr1[sx] = -1; off0[sx] = off1[sx] = 0;
                if (e0 < n_el) { const int q = e0 / cw; off0[sx] = q * LR + c0 + (e0 - q * cw); pr0[sx] = (int)s_rowpair[16 * m + q]; v0[sx] = tile[e0]; }
                if (e1 < n_el) { const int q = e1 / cw; off1[sx] = q * LR + c0 + (e1 - q * cw); pr1[sx] = (int)s_rowpair[16 * m + q]; v1[sx] = tile[e1]; }
            }
            __syncthreads();                                          // the tile is free for the next one
            const unsigned myp = s_rowpair[16 * m + (lane & 15)];     // the tile's 16 pairs, for the test "does row b touch this tile at all"
            const size_t prow0 = (size_t)itg + 16 * m;
            for (int b = b0; b < b1; ++b) {
                const int fd_e = fd_element(fd, fd_fixed, NC, b);
                const int vb = fd_e >= 0 ? fd_e / (2 * NC) : -1;
                const size_t ob = ((size_t)b * t.n_pairs + prow0) * LR;
                double* gp = t.out + ob;
                const int odd = (int)(ob & 1);
                const bool touched = __builtin_amdgcn_ballot_w64((int)(myp & 0xffffu) == vb || (int)(myp >> 16) == vb) != 0;
                if (!touched) {
#pragma unroll
                    for (int sx = 0; sx < kSlots; ++sx) {
                        if (pr1[sx] >= 0 && off1[sx] == off0[sx] + 1 && ((off0[sx] ^ odd) & 1) == 0) store_nt2(gp + off0[sx], v0[sx], v1[sx]);
                        else {
                            if (pr0[sx] >= 0) store_nt(gp + off0[sx], v0[sx]);
                            if (pr1[sx] >= 0) store_nt(gp + off1[sx], v1[sx]);
                        }
                    }
                } else {
#pragma unroll
                    for (int sx = 0; sx < kSlots; ++sx) {
                        const bool w0 = pr0[sx] >= 0 && (pr0[sx] & 0xffff) != vb && (pr0[sx] >> 16) != vb;
                        const bool w1 = pr1[sx] >= 0 && (pr1[sx] & 0xffff) != vb && (pr1[sx] >> 16) != vb;
                        if (w0 && w1 && off1[sx] == off0[sx] + 1 && ((off0[sx] ^ odd) & 1) == 0) store_nt2(gp + off0[sx], v0[sx], v1[sx]);
                        else {
                            if (w0) store_nt(gp + off0[sx], v0[sx]);
                            if (w1) store_nt(gp + off1[sx], v1[sx]);
                        }
                    }
                }
            }
        }
    }
}

// F: a perturbed row's part -- the elevated separation rows of every pair that contains vehicle v of batch row b, 32 pairs
// to a pass (two tiles), every finished row one 16-byte-wide store instruction of one wave.  The chain per element is the
// matrix instruction's, as in the batch kernels and in the S kind: the brute-force sweep's bits.
// img: [32][PA], tile: [16][min(LR, 128)] doubles of LDS behind the staged row.
template <int NC>
__device__ __forceinline__ void tsep_elev_rows_of_vehicle(const TsepXYParams& t, const double2* xy, const int vpq, const int b,
                                                          const int n_veh, const int v, double* img, double* tile)
{
    constexpr int L = 2 * NC - 1;
    using E = ElevMfma<L>;
    const int LR = L + t.R, NT = (LR + 15) >> 4;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_to = t.n_tobj > 0 ? t.n_tobj : n_veh;
    const int n_rows = n_to - 1;
    __shared__ int s_q[32];                                            // position of the pass's pairs in the lexicographic pair list
    for (int cg = 0; 8 * cg < NT; ++cg) {
        const int c0 = 128 * cg, cw = min(LR - c0, 128);
        double bfr[2][E::KS];
        coop_load_bfrag<L>(t.Tf, NT, cg, wave, lane, bfr);
        for (int u00 = 0; u00 < n_rows; u00 += 32) {
            __syncthreads();                                          // the image and s_q of the previous pass have been read
            if (tid < 32) {
                const int u0 = min(u00 + tid, n_rows - 1);            // (lanes beyond the last pair: a valid pair, never stored)
                const int u = u0 < v ? u0 : u0 + 1;
                const int i = min(u, v), j = max(u, v);
                s_q[tid] = i * (2 * n_to - i - 1) / 2 + (j - i - 1);
                tsep_image_row<NC>(t, xy, vpq, tsep_slot(t, i), tsep_slot(t, j), img, tid);
            }
            __syncthreads();
            const int n_pass = min(32, n_rows - u00);
            for (int m = 0; 16 * m < n_pass; ++m) {
                const int rows = min(16, n_pass - 16 * m);
                coop_tile<L>(img, m, bfr, t.offset, tile, cw, cg, NT, wave, lane);
                __syncthreads();
                for (int r = wave; r < rows; r += 4) {
                    const size_t o = ((size_t)b * t.n_pairs + (size_t)s_q[16 * m + r]) * LR + c0;
                    const double* src = tile + r * cw;
                    const int sh = (int)(o & 1);                      // make the 16-byte stores 16-byte aligned
                    const int e0 = 2 * lane - sh, e1 = e0 + 1;        // (cw <= 128: 64 lanes x 2 cover cw + 1 elements)
                    if (e0 >= 0 && e1 < cw) store_nt2(t.out + o + e0, src[e0], src[e1]);
                    else if (e0 >= 0 && e0 < cw) store_nt(t.out + o + e0, src[e0]);
                    else if (e0 < 0 && e1 < cw) store_nt(t.out + o + e1, src[e1]);
                    if (sh && lane == 0 && cw == 128) store_nt(t.out + o + 127, src[127]);      // the one element 64 shifted pieces do not reach
                }
                __syncthreads();
            }
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
    double* tile = ELEV ? lds + ((p.stage_slots * S::VP + 1) & ~1) + wave * elev_mfma_wave_doubles(S::L, p.R)   // (16-byte aligned)
                        : lds + p.stage_slots * S::VP + wave * (p.tile_rows * S::TPF);

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
                if (p.sel_k > 0) {              // the item's sel_k smallest control points (separationRows='active')
                    Smallest4 sm;
#pragma unroll
                    for (int k = 0; k < L; ++k) sm.put(p.sign * cf[k] + p.offset, k);
                    if (mine) sm.store(p.out, p.sel_idx, (row + r) * p.sel_k, p.sel_k);
                } else {
                    double m = cf[0];
#pragma unroll
                    for (int k = 1; k < L; ++k) m = fmin(m, cf[k]);
                    if (mine) p.out[row + r] = p.sign * m + p.offset;
                }
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
            // elev(R) (bezier.py:469-495, 1127-1147) as the matrix product of the group's [64 x L] coefficients with the
            // elevation matrix, output transform folded in: out = offset + sum_j (sign c_j) T[j][k]
            double ch[L];
#pragma unroll
            for (int j = 0; j < L; ++j) ch[j] = p.sign * cf[j];
            if (MINONLY) {
                // only the row's minimum leaves the lane: the same chain per output column, a row of the dense table
                // as scalar operands
                const ctab_t Td = as_ctab(p.Td);
                if (p.sel_k > 0) {
                    Smallest4 sm;
                    for (int k = 0; k < LR; ++k) sm.put(elev_at<L>(ch, Td + k * L, p.offset), k);
                    if (mine) sm.store(p.out, p.sel_idx, (row + r) * p.sel_k, p.sel_k);
                } else {
                    double m = INFINITY;
                    for (int k = 0; k < LR; ++k) m = fmin(m, elev_at<L>(ch, Td + k * L, p.offset));
                    if (mine) p.out[row + r] = m;
                }
            } else {
                // (history at C5, R = 100, 2.26 GB per launch: lane = output column with LDS broadcasts 0.65 ms; lane = item with
                // 32-column chunks 0.87-1.24; lane = (row, 8 columns) with the binomial window in registers 0.52-0.55; this form: section 4.1 of DESIGN.md)
                elev_rows_mfma<L>(ch, mine, r, p.Tf, LR, p.offset, tile, p.out, row * LR, n_valid, lane);
            }
        }
    }
}

// =====================================================================================
//  DEG_ELEV > 0, the batch form: elevated separation rows with the elevation matrix STATIONARY in the waves' registers
// =====================================================================================
// normsq_elev_body's elevated form gives every wave its own 64 rows and all of the matrix (NTG x KS fragments, 96
// registers at degree 10): with the four tiles' A fragments it does not fit 256 registers, and every reload -- spilled
// register, fragment, pair index -- is a vector-memory LOAD in a loop that streams stores: s_waitcnt vmcnt counts both, so
// the wave drains its stores once per 16-row tile (0.445 ms at C5, matrix pipe 40 % busy).  Here a workgroup of four
// waves works on ONE 16-row tile at a time and wave w owns output tiles t = w, w + 4, ... of it: its share of the matrix is
// NTW x KS fragments (24 registers), loaded once.  A set is four 64-row groups: wave w forms the product coefficients of
// group w (lane = pair, kept in its registers); then, tile after tile, the owning wave puts the tile's 16 rows into a
// small A buffer, every wave runs its KS-instruction chains on them and writes its 16 x 16 blocks into the shared output
// tile -- the output run of the 16 rows, 128-byte aligned when the output is -- and all 256 threads copy it out as linear
// 16-byte stores.  A buffer and tile are double buffered: one workgroup barrier per tile.  Nothing is loaded from memory
// inside the loop (pair indices wait in LDS), so no wave ever waits for a store.
template <int LIN>
struct ElevCoop {
    using E = ElevMfma<LIN>;
    static constexpr int ABUF = 16 * E::PA;               // doubles of one A buffer: 16 rows, pitch PA
    static __host__ __device__ constexpr int tile_doubles(int LR) { return (16 * LR + 1) & ~1; }
    // doubles of LDS behind the staged objects: 2 A buffers, 2 tiles; the pair indices (4 bytes each) follow
    static __host__ __device__ constexpr int lds_doubles(int LR) { return 2 * ABUF + 2 * tile_doubles(LR); }
};

// `total` doubles of an LDS run -> gout[e0 ..), all threads of the workgroup, 16-byte stores
__device__ __forceinline__ void store_run_wg(const double* __restrict__ tile, double* __restrict__ gout, const size_t e0, const int total)
{
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int shift = (int)(e0 & 1);
    if (shift == 0 && (total & 1) == 0 && (reinterpret_cast<size_t>(tile) & 15) == 0) {
        const d2_t* t2 = reinterpret_cast<const d2_t*>(tile);
        d2_t* g2 = reinterpret_cast<d2_t*>(gout + e0);
        const int np = total >> 1;
        for (int m0 = tid; m0 < np; m0 += 4 * nthr) {
            d2_t v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) if (m0 + u * nthr < np) v[u] = t2[m0 + u * nthr];
#pragma unroll
            for (int u = 0; u < 4; ++u) if (m0 + u * nthr < np) g2[m0 + u * nthr] = v[u];
        }
        return;
    }
    // (a run that starts on an odd element, or an LDS run that is only 8-byte aligned: 16-byte stores all the same, their halves
    // read one by one; four pieces per trip as above)
    const int npairs = (total + shift + 1) >> 1;
    for (int m0 = tid; m0 < npairs; m0 += 4 * nthr) {
        double x[4], y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int a0 = 2 * (m0 + u * nthr) - shift, a1 = a0 + 1;
            x[u] = (a0 >= 0 && a0 < total) ? tile[a0] : 0.0;
            y[u] = a1 < total ? tile[a1] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int a0 = 2 * (m0 + u * nthr) - shift, a1 = a0 + 1;
            if (a0 >= 0 && a1 < total) { d2_t v; v.x = x[u]; v.y = y[u]; *reinterpret_cast<d2_t*>(gout + e0 + a0) = v; }
            else if (a0 >= 0 && a0 < total) gout[e0 + a0] = x[u];
            else if (a0 < 0 && a1 < total) gout[e0 + a1] = y[u];
        }
    }
}


// (b, w) = evaluation row and workgroup index inside the row, as normsq_elev_body; MODE 0 (pairs), every object of the row
// staged (p.stage_all), 2n + R + 1 <= 64 NTW columns, object ids below 65536.  blockDim.x == 256.
template <int NC, int DIM, int NTW>
__device__ __forceinline__ void sep_elev_coop_body(const NsParams& p, const int b, const int w, double* lds)
{
    using S = NsShape<NC, DIM>;
    constexpr int L = S::L;
    using E = ElevMfma<L>;
    using C = ElevCoop<L>;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int LR = L + p.R, NT = (LR + 15) >> 4;
    const int chunk = kWave * p.groups_per_wg;
    const int it0 = p.item_begin + w * chunk;
    const int it_end = min(p.item_begin + p.item_count, it0 + chunk);
    if (it0 >= it_end) return;
    const int n_items = it_end - it0;

    double* vl = lds;
    double* abuf = lds + ((p.stage_slots * S::VP + 1) & ~1);
    double* tiles = abuf + 2 * C::ABUF;
    const int tile_d = C::tile_doubles(LR);
    unsigned* pr = reinterpret_cast<unsigned*>(tiles + 2 * tile_d);          // [n_items] i | j << 16

    const double* Yrow = p.fd ? p.Y : p.Y + (size_t)b * p.n_veh * S::VLEN;
    const int fd_e = fd_element(p.fd, p.fd_fixed, NC, b);
    stage_objects<NC, DIM>(vl, Yrow, p.obs, p.n_veh, 0, p.n_obj, 0, threadIdx.x, blockDim.x, fd_e, p.fd_h);
    for (int e = threadIdx.x; e < n_items; e += blockDim.x) {
        const int2 ij = p.pairs[it0 + e];
        pr[e] = (unsigned)ij.x | ((unsigned)ij.y << 16);
    }
    double bfr[NTW][E::KS];
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const double* f = p.Tf + (size_t)min(wave + 4 * i, NT - 1) * (E::KS * kWave);
#pragma unroll
        for (int sx = 0; sx < E::KS; ++sx) bfr[i][sx] = f[sx * kWave + lane];
    }
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int sx = 0; sx < E::KS; ++sx) pin_reg(bfr[i][sx]);
    __syncthreads();

    const int n_groups = (n_items + kWave - 1) >> 6;
    const size_t row0 = (size_t)b * p.item_count + (size_t)(it0 - p.item_begin);
    int cnt = 0;                                          // tiles so far: parity of the A buffer and of the output tile
    for (int g0 = 0; g0 < n_groups; g0 += 4) {
        // ---- this wave's group of the set: product coefficients, sign applied, in registers
        double ch[L];
        {
            const int e = min((g0 + wave) * kWave + lane, n_items - 1);      // (waves beyond the last group, lanes beyond the last pair: a valid pair, never stored)
            const unsigned pij = pr[e];
            const double* vi = vl + (int)(pij & 0xffffu) * S::VP;
            const double* vj = vl + (int)(pij >> 16) * S::VP;
            double a[DIM][NC];
#pragma unroll
            for (int q = 0; q < DIM; ++q)
#pragma unroll
                for (int c = 0; c < NC; ++c) a[q][c] = vi[q * NC + c] - vj[q * NC + c];
            double cf[L];
            normsq_coeffs<NC, DIM>(a, as_ctab(p.W2), cf);
#pragma unroll
            for (int j = 0; j < L; ++j) ch[j] = p.sign * cf[j];
        }
        const int set_tiles = 4 * min(4, n_groups - g0);
        // rows of the set's first tile (group g0, rows 0..15: wave 0, lanes 0..15)
        if (wave == 0 && lane < 16) elev_store_image_row<L>(abuf + (cnt & 1) * C::ABUF, lane, ch);
        __syncthreads();
        for (int mt = 0; mt < set_tiles; ++mt, ++cnt) {
            const int gg = mt >> 2, m = mt & 3;
            const int rows = min(16, n_items - (g0 + gg) * kWave - 16 * m);      // uniform over the workgroup
            double* tile = tiles + (cnt & 1) * tile_d;
            // the NEXT tile's rows into the other A buffer (its last readers passed the previous barrier)
            if (mt + 1 < set_tiles && wave == ((mt + 1) >> 2) && (lane >> 4) == ((mt + 1) & 3))
                elev_store_image_row<L>(abuf + ((cnt + 1) & 1) * C::ABUF, lane & 15, ch);
            if (rows > 0) {
                double afr[E::KS];
                elev_load_afrag<L>(abuf + (cnt & 1) * C::ABUF, 0, lane, afr);
#pragma unroll
                for (int i = 0; i < NTW; ++i) {
                    const int t = wave + 4 * i;
                    if (t < NT) elev_acc_to_tile(elev_mfma_tile<L>(afr, bfr[i], p.offset), t, tile, LR, lane);
                }
            }
            __syncthreads();
            if (rows > 0) store_run_wg(tile, p.out, (row0 + (size_t)((g0 + gg) * kWave + 16 * m)) * LR, rows * LR);
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
    if (b + p.fd - 1 > 0) {
        const int free_cols = NC - 2 * p.fd_fixed, kq = b + p.fd - 2, pr = kq / free_cols, pc = p.fd_fixed + (kq - pr * free_cols);
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
// sub (dynamics_elev_group, mode 2): >= 0: the workgroup's four waves share the 16 items 16 sub .. 16 sub + 15 of the group,
//           each wave a quarter of the column tiles -- a stream workgroup then repeats row 0's matrix products for 16 vehicles
//           and four times the rows instead of 64 vehicles; -1: wave w takes items 16 w .. 16 w + 15 (everything else).
// fd_map (mode 2): >= 0: the AngParams::fd that names the rows of the emission map when the group's items are read with
//           fd = 0 (the view's unperturbed row as a one-row batch -- a row range whose local row 0 is a perturbed row).
// mode 1: item_begin = the first perturbed local row (1, or 0 in such a range), item_end = how many there are.
struct DynEmit {
    int mode, item_begin, item_end, b0, b1;
    int* s_map;
    int sub = -1;
    int fd_map = -1;
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
        const int b = min(group * kWave + lane, em->item_end - 1) + em->item_begin;
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
                const int fd_e = fd_element(em.fd_map >= 0 ? em.fd_map : p.fd, p.fd_fixed, NC, b);
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

// ---- angular rate / speed arithmetic in separable form (late round 4) -----------------------------------------------------
// u_j = C(n, j) a_j turns the product of two degree-n Bernstein curves into a plain convolution of the u's,
//     C(2n, k) (a b)_k = sum_j ua_j ub_{k-j},
// so the degree-2n curves of optimization.py:603-605 are kept as raw_k = C(2n, k) x coefficient, their squares at degree
// 4n are plain folded convolutions of the raws, and the quotient num.cpts / den.cpts (optimization.py:608) needs no binomial
// at all -- the same 1 / C(4n, k) stands on both sides -- while DEG_ELEV > 0 wants exactly C(4n, k) x coefficient as the
// operand of its convolution form.  One fused multiply-add per term where the weighted sums took two to four instructions.
// Used by the DEG_ELEV > 0 groups (dynamics_elev_group: phase A 2000 -> 1100 instructions per lane; C5's fused launch 0.594 ->
// 0.559 ms, same box, interleaved).  The DEG_ELEV = 0 groups keep the weighted sums: there the same change bought C3 nothing
// (its dynamics groups fill the sweep's tail) and cost C4 1.6 % (16 control points: the four scaled derivative arrays on top
// of a body that already spills under the tiled sweep's 128 registers).  C(n, .) and 1 / C(2n, .) sit behind the plain
// product weights (capi.cpp plain_product_weights).
template <int NC>
__device__ __forceinline__ void ang_scale(const double (&a)[NC], const ctab_t Cn, double (&u)[NC])
{
#pragma unroll
    for (int j = 0; j < NC; ++j) u[j] = Cn[j] * a[j];
}

// raw_k = C(2n, k) (xD^2 + yD^2)_k from the scaled derivatives
template <int NC>
__device__ __forceinline__ void ang_raw_den(const double (&ux)[NC], const double (&uy)[NC], double (&raw)[2 * NC - 1])
{
    constexpr int N = NC - 1, L2 = 2 * N + 1;
#pragma unroll
    for (int k = 0; k < L2; ++k) {
        const int jlo = k - N > 0 ? k - N : 0;
        double so = 0.0;
#pragma unroll
        for (int j = jlo; 2 * j < k; ++j) {
            so = j == jlo ? ux[j] * ux[k - j] : fma(ux[j], ux[k - j], so);
            so = fma(uy[j], uy[k - j], so);
        }
        if ((k & 1) == 0) {
            const int h = k >> 1;
            const double dg = fma(uy[h], uy[h], ux[h] * ux[h]);
            raw[k] = (2 * jlo < k) ? fma(2.0, so, dg) : dg;
        } else raw[k] = so + so;
    }
}

// raw_k = C(2n, k) (yDD xD - xDD yD)_k
template <int NC>
__device__ __forceinline__ void ang_raw_num(const double (&uyDD)[NC], const double (&uxD)[NC], const double (&uxDD)[NC],
                                            const double (&uyD)[NC], double (&raw)[2 * NC - 1])
{
    constexpr int N = NC - 1, L2 = 2 * N + 1;
#pragma unroll
    for (int k = 0; k < L2; ++k) {
        const int jlo = k - N > 0 ? k - N : 0, jhi = N < k ? N : k;
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int j = jlo; j <= jhi; ++j) {
            s1 = j == jlo ? uyDD[j] * uxD[k - j] : fma(uyDD[j], uxD[k - j], s1);
            s2 = j == jlo ? uxDD[j] * uyD[k - j] : fma(uxDD[j], uyD[k - j], s2);
        }
        raw[k] = s1 - s2;
    }
}

// coefficient k of the square of a raw curve of LIN coefficients: C(2 (LIN - 1), k) x (the curve's square)_k
template <int LIN>
__device__ __forceinline__ double fold_square_at(const double (&u)[LIN], const int k)
{
    constexpr int M = LIN - 1;
    const int jlo = k - M > 0 ? k - M : 0;
    double so = 0.0;
#pragma unroll
    for (int j = jlo; 2 * j < k; ++j) so = j == jlo ? u[j] * u[k - j] : fma(u[j], u[k - j], so);
    if ((k & 1) == 0) {
        const int h = k >> 1;
        const double dg = u[h] * u[h];
        return (2 * jlo < k) ? fma(2.0, so, dg) : dg;
    }
    return so + so;
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
    int* flags;                      // obtg_ctx_set_ang_rate_order(2): flags[0] counts, flags[1 ..] lists the items whose speed curve
                                     // comes near zero (their angular-rate rows are recomputed in double-double); else nullptr
};

// LDS doubles of dynamics_elev_group: the two binomial rows as the matrix instruction's B operand reads them, and 1/C(2n+R, .)
__host__ __device__ constexpr int dyn_elev_lds_doubles(int n, int R)
{
    const int L2 = 2 * n + 1, L4 = 4 * n + 1;
    const int pad4 = 4 * ((L4 + 3) / 4) + 4, pad2 = 4 * ((L2 + 3) / 4) + 4;
    const int nt4 = (L4 + 4 * R + 15) / 16, nt2 = (L2 + R + 15) / 16;
    return (pad4 + 16 * nt4) + (pad2 + 16 * nt2) + 16 * nt2;
}
// ... and of the area behind them in which a shared stream workgroup (DynEmit::sub) collects its 16 items' rows
__host__ __device__ constexpr int dyn_elev_stage_doubles(int n, int R)
{
    const int a = 16 * (4 * n + 1 + 4 * R), sp = 2 * 16 * (2 * n + 1 + R);
    return a > sp ? a : sp;
}

// Angular rate and speed rows with DEG_ELEV = R > 0 for the 64 items of `group`, four waves; wave w takes items 16 w ..
// 16 w + 15.  Phase A (registers): the degree-4n numerator and denominator and the degree-2n speed curve of the lane's
// item from the ORIGINAL control points, as k_dynamics2 forms them -- four lanes per item, lane (c, kg) = item c of the
// tile, coefficients j = kg mod 4: exactly the A operand of v_mfma_f64_16x16x4_f64, no transposition.  Phase B: the two
// elevations by 4R (and the speed curve's by R) as matrix products.  Elevation is a binomially scaled convolution,
//     elev(a, Q)_k = (1 / C(P+Q, k)) sum_j [C(P, j) a_j] C(Q, k-j),
// so with the coefficients pre-scaled by C(P, j) the B operand is a TOEPLITZ matrix, B[j][k] = C(Q, k-j): fragment
// (k-step s, column tile t) is the table entry 16 t + c - 4 s - kg of one row of a few hundred doubles, which waits in
// LDS -- nothing is read from memory while the wave streams its stores (a vector-memory load there would wait for every
// store issued before it: s_waitcnt vmcnt counts both).  In the quotient num_k / den_k the factor 1 / C(P+Q, k) cancels
// and is never applied; the row C(4R, .) is scaled by a power of two that keeps C(4R, .) x C(4n, .) x values far from
// overflow.  22 matrix instructions per 16 x 16 block of quotients against 2 x 41 x 16 v_fma_f64 per lane before
// (k_dynamics_elev 0.20 ms at C5); results leave straight from the accumulators, 128-byte row segments.
template <int NC>
__device__ __forceinline__ void dynamics_elev_group(const AngElevParams& q, double* lds, const int group, const DynEmit* em = nullptr)
{
    constexpr int N = NC - 1, L2 = 2 * N + 1, L4 = 4 * N + 1;
    constexpr int KS2 = (L2 + 3) / 4, KS4 = (L4 + 3) / 4, PAD2 = 4 * KS2 + 4, PAD4 = 4 * KS4 + 4;
    const AngParams& p = q.a;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, kg = lane >> 4;
    const int L2R = L2 + q.R, L4R = L4 + 4 * q.R, NT2 = (L2R + 15) >> 4, NT4 = (L4R + 15) >> 4;
    const bool mapped = em && em->mode != 0;
    // ---- tables -> LDS: hl4[PAD4 + m] = C(4R, m) 2^-e, hl2[PAD2 + m] = C(R, m) (zero outside the row), inv2l[k] = 1 / C(2n+R, k)
    double* hl4 = lds;
    double* hl2 = hl4 + PAD4 + 16 * NT4;
    double* inv2l = hl2 + PAD2 + 16 * NT2;
    {
        const int tid = threadIdx.x, nthr = blockDim.x;
        const double* row4 = q.cv4 + L4;
        const int rl4 = 2 * (L4 - 1) + 4 * q.R + 1 + kElevBlock;
        for (int i = tid; i < PAD4 + 16 * NT4; i += nthr) { const int qd = i - PAD4 + (L4 - 1); hl4[i] = (qd >= 0 && qd < rl4) ? row4[qd] : 0.0; }
        const double* row2 = q.cv2 + L2;
        const int rl2 = 2 * (L2 - 1) + q.R + 1 + kElevBlock;
        for (int i = tid; i < PAD2 + 16 * NT2; i += nthr) { const int qd = i - PAD2 + (L2 - 1); hl2[i] = (qd >= 0 && qd < rl2) ? row2[qd] : 0.0; }
        const double* inv2 = row2 + rl2;
        for (int i = tid; i < 16 * NT2; i += nthr) inv2l[i] = i < L2R ? inv2[i] : 0.0;
    }
    const bool split = mapped && em->sub >= 0;          // (wave-uniform)
    const int wi = split ? em->sub : wave;               // which 16 items of the group are this wave's
    const int t_first = split ? wave : 0, t_step = split ? 4 : 1;
    int it0, n_valid;
    const int item = dyn_item_of_lane<NC>(p, em, group, 16 * wi + c, it0, n_valid);
    const int b = item / p.n_veh;
    if (mapped && wave == 0) {
        int it0l, nvl;
        const int item_l = dyn_item_of_lane<NC>(p, em, group, lane, it0l, nvl);
        dyn_emit_prepare<NC>(p, *em, item_l, it0l, nvl, lane);
    }
    // ---- phase A: degree-4n numerator and denominator (as k_dynamics2), pre-scaled by C(4n, k); the speed curve by C(2n, j).
    // The denominator side first, then the numerator side, a scheduling barrier between them: interleaved, the two sides'
    // degree-2n curves and derivatives are live together and the kernel needs 235 registers (two workgroups per CU).
    double afn[KS4], afd[KS4], afs[KS2];
    {
        double x[NC], y[NC];
        load_item_xy<NC>(p, item, b, x, y);
        const double val = (double)N / p.tf[b];
        double xD[NC], yD[NC];
        diff_elev1<NC>(x, val, xD);
        diff_elev1<NC>(y, val, yD);
        const ctab_t Cn = as_ctab(p.Wn) + L2 * NC;             // C(n, .) behind the plain product weights
        // the lane's share of an operand: coefficients j = 4 s + kg (zero past the end)
        // (the four candidates are computed by every lane and pinned: left to itself the compiler sinks each candidate's
        // arithmetic AND its table load into a branch of its own -- vector loads with a wait each, a hundred per wave)
        auto pick = [&](double (&v)[4]) {
#pragma unroll
            for (int u = 0; u < 4; ++u) pin_reg(v[u]);
            return kg == 0 ? v[0] : kg == 1 ? v[1] : kg == 2 ? v[2] : v[3];
        };
        double uxD[NC], uyD[NC];
        ang_scale<NC>(xD, Cn, uxD);
        ang_scale<NC>(yD, Cn, uyD);
        {
            // C(2n, k) x den1_k IS the speed curve's operand of the convolution form, its folded square C(4n, k) x den_k the denominator's
            double den1[L2];
            ang_raw_den<NC>(uxD, uyD, den1);
#pragma unroll
            for (int sx = 0; sx < KS2; ++sx) {
                double v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = (4 * sx + u < L2) ? den1[4 * sx + u < L2 ? 4 * sx + u : 0] : 0.0;
                afs[sx] = pick(v);
            }
#pragma unroll
            for (int sx = 0; sx < KS4; ++sx) {
                double v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = (4 * sx + u < L4) ? fold_square_at<L2>(den1, 4 * sx + u < L4 ? 4 * sx + u : 0) : 0.0;
                afd[sx] = pick(v);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            double xDD[NC], yDD[NC];
            diff_elev1<NC>(xD, val, xDD);
            diff_elev1<NC>(yD, val, yDD);
            double uxDD[NC], uyDD[NC];
            ang_scale<NC>(xDD, Cn, uxDD);
            ang_scale<NC>(yDD, Cn, uyDD);
            double num1[L2];
            ang_raw_num<NC>(uyDD, uxD, uxDD, uyD, num1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int sx = 0; sx < KS4; ++sx) {
                double v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = (4 * sx + u < L4) ? fold_square_at<L2>(num1, 4 * sx + u < L4 ? 4 * sx + u : 0) : 0.0;
                afn[sx] = pick(v);
            }
        }
    }
    __syncthreads();                                   // the tables (and the emission map) are in LDS
    // value of (row of the group, column) -> where the emission mode wants it
    const size_t row_stride = (size_t)p.n_veh;
    auto emit = [&](double* __restrict__ out, const int LROW, const int row, const int col, const double v) {
        if (row >= n_valid || col >= LROW) return;
        if (!mapped) { out[(size_t)(it0 + row) * LROW + col] = v; return; }
        if (em->mode == 1) {
            const int it = em->s_map[row];
            if (it >= 0) out[(size_t)it * LROW + col] = v;
            return;
        }
        const int nb = em->b1 - em->b0;
        double* o = out + ((size_t)em->b0 * p.n_veh + it0 + row) * LROW + col;
        for (int j = 0; j < nb; ++j, o += row_stride * LROW) {
            const int m = em->s_map[j];
            if (m != -2 && m != row) *o = v;
        }
    };
    const int row_base = 16 * wi + kg;                 // this lane's rows: row_base + 4 r
    // Shared stream workgroups (DynEmit::sub): the 16 items' rows are collected in LDS behind the tables -- consecutive
    // vehicles' rows are one contiguous run of every batch row -- and leave as whole-workgroup runs of 16-byte stores, one
    // or two per batch row (two: the row advances one of these vehicles and its own workgroup writes that one).
    const bool staged = split && em->mode == 2;
    double* stage = inv2l + 16 * NT2;                  // [16][L4R] doubles (the speed rows before: [16][L2R], twice)
    const int sub_rows = staged ? min(16, n_valid - 16 * wi) : 0;
    auto stream_staged = [&](const double* src, double* __restrict__ out, const int LROW) {
        const int nb = em->b1 - em->b0;
        for (int j = 0; j < nb; ++j) {
            const int m = em->s_map[j];
            if (m == -2) continue;                                  // (tf[b] is not tf[0]: the row has workgroups of its own)
            const int skip = m - 16 * wi;                           // the advanced vehicle's row among these 16, if any
            const size_t e0 = ((size_t)(em->b0 + j) * p.n_veh + it0 + 16 * wi) * LROW;
            if (skip < 0 || skip >= sub_rows) store_run_wg(src, out, e0, sub_rows * LROW);
            else {
                if (skip > 0) store_run_wg(src, out, e0, skip * LROW);
                if (skip + 1 < sub_rows) store_run_wg(src + (skip + 1) * LROW, out, e0 + (size_t)(skip + 1) * LROW, (sub_rows - skip - 1) * LROW);
            }
        }
    };
    // ---- speed rows = elev(den1, R), both requested bounds from the same accumulators
    const bool flagging = q.flags != nullptr && p.out != nullptr && !mapped;
    if (p.out_speed || flagging) {
        const double* bp = hl2 + PAD2 + c - kg;
        double mn[4] = { INFINITY, INFINITY, INFINITY, INFINITY }, mx[4] = { -INFINITY, -INFINITY, -INFINITY, -INFINITY };
        for (int t = t_first; t < NT2; t += t_step) {
            v4d_t acc;
            acc[0] = acc[1] = acc[2] = acc[3] = 0.0;
#pragma unroll
            for (int sx = 0; sx < KS2; ++sx) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(afs[sx], bp[16 * t - 4 * sx], acc, 0, 0, 0);
            const int col = 16 * t + c;
            const double iv = inv2l[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double e = acc[r] * iv;
                if (col < L2R) { mn[r] = fmin(mn[r], e); mx[r] = fmax(mx[r], e); }
                if (p.out_speed && staged) {
                    if (col < L2R) {
                        stage[(kg + 4 * r) * L2R + col] = p.sp_sign * e + p.sp_offset;
                        if (p.out_speed2) stage[(16 + kg + 4 * r) * L2R + col] = p.sp2_sign * e + p.sp2_offset;
                    }
                } else if (p.out_speed) {
                    emit(p.out_speed, L2R, row_base + 4 * r, col, p.sp_sign * e + p.sp_offset);
                    if (p.out_speed2) emit(p.out_speed2, L2R, row_base + 4 * r, col, p.sp2_sign * e + p.sp2_offset);
                }
            }
        }
        if (p.out_speed && staged) {
            __syncthreads();
            stream_staged(stage, p.out_speed, L2R);
            if (p.out_speed2) stream_staged(stage + 16 * L2R, p.out_speed2, L2R);
            __syncthreads();                                        // the angular-rate rows take the area over
        }
        if (flagging) {
            // obtg_ctx_set_ang_rate_order(2).  Near a stop: the elevated control points of |v|^2 -- tight to the curve at
            // DEG_ELEV of any size -- fall below 2 % of their largest (or are not positive).  The elevated denominator
            // (|v|^2)^2 is then a sum of terms ten thousand times itself; the row goes on the list of the double-double pass.
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) { mn[r] = fmin(mn[r], __shfl_xor(mn[r], m)); mx[r] = fmax(mx[r], __shfl_xor(mx[r], m)); }
                const int row = row_base + 4 * r;
                if (c == 0 && row < n_valid && !(mn[r] > 2e-2 * mx[r])) q.flags[1 + atomicAdd(q.flags, 1)] = it0 + row;
            }
        }
    }
    // ---- angular rate: elevate numerator and denominator by 4R, divide element by element (optimization.py:608)
    if (p.out) {
        const double* bp = hl4 + PAD4 + c - kg;
        for (int t = t_first; t < NT4; t += t_step) {
            v4d_t an, ad;
            an[0] = an[1] = an[2] = an[3] = 0.0;
            ad[0] = ad[1] = ad[2] = ad[3] = 0.0;
#pragma unroll
            for (int sx = 0; sx < KS4; ++sx) {
                const double bv = bp[16 * t - 4 * sx];
                an = __builtin_amdgcn_mfma_f64_16x16x4f64(afn[sx], bv, an, 0, 0, 0);
                ad = __builtin_amdgcn_mfma_f64_16x16x4f64(afd[sx], bv, ad, 0, 0, 0);
            }
            const int col = 16 * t + c;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double v = p.w2 - an[r] / ad[r];
                if (staged) { if (col < L4R) stage[(kg + 4 * r) * L4R + col] = v; }
                else emit(p.out, L4R, row_base + 4 * r, col, v);
            }
        }
        if (staged) {
            __syncthreads();
            stream_staged(stage, p.out, L4R);
        }
    }
}

}  // namespace obtg
