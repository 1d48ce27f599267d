// GJK convex-hull distance sweeps and the curve<->curve / curve<->polygon branch & bound.
//
// Reference: gjk/gjk.py:230-681 (gjkNew and helpers), bezier.py:1283-1558 (_minDist,
// _minDist2Poly, _upperbound, _upperboundPoly), bezier.py:985-1027 (deCasteljauSplit),
// optimization.py:109-133 (spatialSeparationConstraints pair loop).
//
// This translation unit is compiled with -ffp-contract=off: the reference's branch
// decisions must be reproduced bit for bit (support indices are the parity target).
//
// Work decomposition: one hull pair per lane.  gjkNew is a short, divergent state machine
// (about five support sweeps over 2(n+1) points); the only expensive step, the support
// scan, is common to every simplex case, so the loop is shaped "small divergent update,
// then convergent supportPts".  For the batched swarm sweep a workgroup stages the control
// polygons of one evaluation row in LDS (structure-of-arrays per vehicle, odd pitch) and
// its lanes walk consecutive pairs of that row.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <hip/hip_ext.h>

#include "gjk_device.h"
#include "gjk_true.h"
#include "obtg_internal.h"
// the Bernstein sweep body that k_pair_sweep runs beside the GJK workgroups: same contraction
// setting as in bern_kernels.hip, so that both units produce the same arithmetic
#pragma clang fp contract(fast)
#include "bern_device.h"
#pragma clang fp contract(off)

constexpr int kSweepChunk = 864;       // hull pairs per workgroup of the plain sweeps (see sweep_shape)
constexpr int kPairSweepChunk = 1280;  // ... and of the one-launch pair sweep

namespace obtg {

using gjk::Ctx;
using gjk::MemGlobal;
using gjk::MemLds;
using gjk::Poly;
using gjk::Result;
using gjk::V3;

// -------------------------------------------------------------------------------------
//  The sweep loop.  Every iteration performs ONE doSimplex step for every lane that holds a
//  pair: a short divergent simplex update, then the support scan as convergent code.  A lane
//  whose pair finishes (collision, converged minimumDistance, cap) emits its result and pulls
//  the next pair of the workgroup's chunk from an LDS counter, so lanes stay busy although
//  gjkNew's trip count varies from 3 to 26 support scans per pair.
// -------------------------------------------------------------------------------------
template <class Mem, bool PLANAR, class Setup, class Emit>
__device__ __forceinline__ void gjk_sweep(int c_end, int* s_next, int max_iter, int md_cap, Setup setup, Emit emit)
{
    const int lane = threadIdx.x & (kWave - 1);
    int k = -1;
    bool exhausted = false;
    Ctx<Mem> g;
    gjk::Simplex s, old;
    gjk::Checkpoint chk;
    V3 dir{ 1.0, 0.0, 0.0 };
    int phase = 0, it = 0, rr = 0;
    s.keys = 0;
    s.A = gjk::Vert{ V3{ 0, 0, 0 }, 0, 0 };
    chk.start(s, dir);
    s.B = s.A; s.C = s.A; s.D = s.A; old = s;
    const double qnan = __builtin_nan("");
    for (;;) {
        const unsigned long long want = __ballot(k < 0 && !exhausted);
        if (want) {
            const int leader = __ffsll((long long)want) - 1;
            int base = 0;
            if (lane == leader) base = atomicAdd(s_next, __popcll(want));
            base = __shfl(base, leader);
            if (k < 0 && !exhausted) {
                const int my = base + __popcll(want & ((1ull << lane) - 1ull));
                if (my < c_end) {
                    k = my;
                    setup(k, g);
                    g.n_support = 0;
                    s.keys = 0; dir = V3{ 1.0, 0.0, 0.0 };
                    // (matches_old_batched fetches poly-1 points at the indices of absent entries too: none may be left from
                    // the previous pair, whose first hull can have had more points than this one's)
                    s.A.i1 = 0; s.B.i1 = 0; s.C.i1 = 0; s.D.i1 = 0;
                    phase = 0; it = 0; rr = 0;
                } else exhausted = true;
            }
        }
        if (__ballot(k >= 0) == 0ull) break;
        if (k >= 0) {
            if (phase == 1) old = s;
            if (gjk::simplex_update(s, dir)) {
                gjk::support_pts<Mem, PLANAR>(g, dir, s.A);
                s.keys |= gjk::kA;
            }
            bool done = false;
            Result r;
            r.c1 = V3{ qnan, qnan, qnan }; r.c2 = r.c1; r.dist = qnan;
            r.flag = -1; r.status = OBTG_ST_OK;
            if (phase == 0) {
                ++it;
                if (s.keys & gjk::kColl) { r.flag = 0; done = true; }
                else if (gjk::dotb(s.A.v, dir) < 0) { phase = 1; rr = 0; chk.start(s, dir); }
                else if (it >= max_iter) { r.flag = -1; r.status = OBTG_ST_MAXITER; done = true; }
            } else {
                ++rr;
                if (gjk::matches_old_batched(g, old, s.A.v)) {
                    gjk::closest_from_simplex(g, old, r);
                    r.flag = 1; done = true;
                } else if (chk.step(s, dir)) { r.flag = 1; r.status = OBTG_ST_CYCLE; done = true; }
                else if (rr >= md_cap) { r.flag = 1; r.status = OBTG_ST_MD_CAP; done = true; }
            }
            if (done) {
                r.n_support = g.n_support;
                emit(k, r);
                k = -1;
            }
        }
    }
}

// planar inputs: the same loop on the 2-D state machine of gjk_device.h
template <class Mem, class Setup, class Emit>
__device__ __forceinline__ void gjk_sweep_planar(int c_end, int* s_next, int max_iter, int md_cap, Setup setup, Emit emit)
{
    const int lane = threadIdx.x & (kWave - 1);
    int k = -1;
    bool exhausted = false;
    Ctx<Mem> g;
    gjk::Simplex2 s, old;
    gjk::V2 dir{ 1.0, 0.0 };
    int phase = 0, it = 0, rr = 0;
    s.keys = 0;
    s.A = gjk::Vert2{ gjk::V2{ 0, 0 }, 0 };
    s.B = s.A; s.C = s.A; old = s;
    const double qnan = __builtin_nan("");
    for (;;) {
        const unsigned long long want = __ballot(k < 0 && !exhausted);
        if (want) {
            const int leader = __ffsll((long long)want) - 1;
            int base = 0;
            if (lane == leader) base = atomicAdd(s_next, __popcll(want));
            base = __shfl(base, leader);
            if (k < 0 && !exhausted) {
                const int my = base + __popcll(want & ((1ull << lane) - 1ull));
                if (my < c_end) {
                    k = my;
                    setup(k, g);
                    g.n_support = 0;
                    s.keys = 0; dir = gjk::V2{ 1.0, 0.0 };
                    phase = 0; it = 0; rr = 0;
                } else exhausted = true;
            }
        }
        if (__ballot(k >= 0) == 0ull) break;
        if (k >= 0) {
            if (phase == 1) old = s;
            gjk::simplex_update2(s, dir);
            gjk::support_pts2(g, dir, s.A);
            s.keys |= gjk::kA;
            bool done = false;
            Result r;
            r.c1 = V3{ qnan, qnan, qnan }; r.c2 = r.c1; r.dist = qnan;
            r.flag = -1; r.status = OBTG_ST_OK;
            if (phase == 0) {
                ++it;
                if (s.keys & gjk::kColl) { r.flag = 0; done = true; }
                else if (gjk::dotb2(s.A.v, dir) < 0) { phase = 1; rr = 0; }
                else if (it >= max_iter) { r.flag = -1; r.status = OBTG_ST_MAXITER; done = true; }
            } else {
                ++rr;
                if (gjk::matches_old2(g, old, s.A.v)) {
                    gjk::closest_from_simplex(g, gjk::lift(old), r);
                    r.flag = 1; done = true;
                } else if (rr >= md_cap) { r.flag = 1; r.status = OBTG_ST_MD_CAP; done = true; }
            }
            if (done) {
                r.n_support = g.n_support;
                emit(k, r);
                k = -1;
            }
        }
    }
}

template <class Mem, bool PLANAR, class Setup, class Emit>
__device__ __forceinline__ void gjk_sweep_any(int c_end, int* s_next, int max_iter, int md_cap, Setup setup, Emit emit)
{
    if (PLANAR) gjk_sweep_planar<Mem>(c_end, s_next, max_iter, md_cap, setup, emit);
    else gjk_sweep<Mem, false>(c_end, s_next, max_iter, md_cap, setup, emit);
}

// -------------------------------------------------------------------------------------
//  generic pair list on global SoA point sets:  poly a = x[K] y[K] z[K] at soa + 3*off[a]
// -------------------------------------------------------------------------------------
struct GjkPairsParams {
    const double* __restrict__ soa;
    const int* __restrict__ off;
    const int* __restrict__ pa;
    const int* __restrict__ pb;
    int n_pairs, max_iter, md_cap, chunk;
    int* __restrict__ flag;
    double* __restrict__ p1;
    double* __restrict__ p2;
    double* __restrict__ dist;
    short* trace;
    int trace_cap;
    int* nsup;
    int* status;
};

template <bool PLANAR>
__global__ __launch_bounds__(256) void k_gjk_pairs(const GjkPairsParams p)
{
    __shared__ int s_next;
    const int c0 = blockIdx.x * p.chunk, c1 = min(p.n_pairs, c0 + p.chunk);
    if (threadIdx.x == 0) s_next = c0;
    __syncthreads();
    gjk_sweep_any<MemGlobal, PLANAR>(
        c1, &s_next, p.max_iter, p.md_cap,
        [&](int k, Ctx<MemGlobal>& g) {
            const int a = p.pa[k], b = p.pb[k];
            g.mem = MemGlobal{ p.soa };
            const int oa = p.off[a], ob = p.off[b];
            const int Ka = p.off[a + 1] - oa, Kb = p.off[b + 1] - ob;
            g.P1 = Poly{ 3 * oa, Ka, Ka, 1 };
            g.P2 = Poly{ 3 * ob, Kb, Kb, 1 };
            g.trace = p.trace ? p.trace + (size_t)k * p.trace_cap * 2 : nullptr;
            g.trace_cap = p.trace_cap;
        },
        [&](int k, const Result& r) {
            p.flag[k] = r.flag;
            p.p1[3 * k] = r.c1.x; p.p1[3 * k + 1] = r.c1.y; p.p1[3 * k + 2] = r.c1.z;
            p.p2[3 * k] = r.c2.x; p.p2[3 * k + 1] = r.c2.y; p.p2[3 * k + 2] = r.c2.z;
            p.dist[k] = r.dist;
            if (p.nsup) p.nsup[k] = r.n_support;
            if (p.status) p.status[k] = r.status;
        });
}

// -------------------------------------------------------------------------------------
//  batched swarm sweep: hulls of the vehicles of row b (LDS) + static polygons (LDS)
// -------------------------------------------------------------------------------------
struct GjkSwarmParams {
    const double* __restrict__ Y;      // [B][n_veh*dim][nc]
    const double* __restrict__ poly;   // SoA polygons
    const int* __restrict__ poly_off;  // [n_poly+1]
    const int* __restrict__ pa;
    const int* __restrict__ pb;
    int n_veh, dim, nc, n_poly, n_poly_pts, n_pairs, wgs_per_row, vp, chunk;
    int max_iter, md_cap;
    const unsigned char* chg;          // FIXUP kernels: [B][n_veh], 1 = vehicle differs from row 0
    // TILED kernels (large rows): host-built chunk metadata, see build_tiles()
    const int* chunk_off;              // [n_chunks+1] pair ranges in tile-major order
    const int* order;                  // [n_pairs] original pair index of each sorted position
    const unsigned* pslots;            // [n_pairs] LDS slots (a | b << 16) of each sorted position
    const int* cobj_off;               // [n_chunks+1] object-list ranges
    const int* cobjs;                  // object ids per chunk, slot order
    int max_objs;                      // most objects any chunk stages
    const int2* chunk_ij;              // [n_chunks] pair sweep: origin (ti0, tj0) of the tile whose separation rows the chunk writes, ti0 < 0: none
    int tile_a;                        // tile height
    // plain sweep (MODE 0): trip-count history used to order each workgroup's pairs, see the kernel
    int B;
    const unsigned char* len_in;       // nullable, [.][n_pairs] support-scan counts of the previous sweep
    int len_in_stride;                 // n_pairs (same batch shape as last time) or 0 (every row reads row 0)
    unsigned char* len_out;            // nullable, [B][n_pairs]
    int fd, fd_fixed;                  // fd != 0: Y is ONE row; row b >= 1 = Y with its (b-1)-th free control point
    double fd_h;                       //          advanced by fd_h (the rows obtg_fd_batch_dev writes), formed while staging
    TsepXYParams ts;                   // pair sweep: the row's temporal-separation block (ts.out != nullptr)
    AngParams dyn;                     // pair sweep: speed / angular-rate groups run by the grid's LAST workgroups (dyn.out != nullptr)
    int dyn_first_block;               //             first of them
    int ts_tile_rows;
    unsigned long long* timeline;      // diagnostics (OBTG_TIMELINE): per workgroup [start, end, hw id, -] on the 100 MHz clock
    int* __restrict__ flag;
    double* __restrict__ p1;
    double* __restrict__ p2;
    double* __restrict__ dist;
    int* nsup;
    int* status;
    int chg_from_fd = 0;               // MODE 1: the changed vehicle of row b is the one the view's row b advances (no mask array):
    const int* vp_off = nullptr;       //         its pairs come from the per-vehicle lists (obtg_ctx::d_vp_off / d_vp_idx)
    const int* vp_idx = nullptr;
    int emit_scalar = 0;               // results may go to LDS through generic pointers: 8-byte stores only
    int refill_min = 1;                // planar sweeps: idle lanes of a wave wait until this many can refill together
    int hist_shift = 0;                // planar sweeps: the history order's bins hold scan counts c with equal c >> hist_shift (0: one count per bin)
    int passes = 1;                    // MODE 0: chunks a workgroup takes one after the other (w-th workgroup of a row: chunks w*passes ..)
    const double* obs = nullptr;       // pair sweep (MODE 0): the context's point obstacles [n_obs][2], staged behind the hull
    int n_obs = 0;                     //         objects as constant curves for the separation rows that name them
};

template <bool PLANAR>
__global__ __launch_bounds__(256) void k_gjk_swarm(const GjkSwarmParams p)
{
    extern __shared__ double lds[];
    __shared__ int s_next;
    const int b = blockIdx.x / p.wgs_per_row, w = blockIdx.x - b * p.wgs_per_row;
    const int vlen = p.dim * p.nc;
    double* vl = lds;                        // [n_veh][vp]
    double* pl = lds + p.n_veh * p.vp;       // polygons, SoA, 3*n_poly_pts doubles
    const double* Yrow = p.fd ? p.Y : p.Y + (size_t)b * p.n_veh * vlen;
    const int fd_e = fd_element(p.fd, p.fd_fixed, p.nc, b);
    for (int e = threadIdx.x; e < p.n_veh * vlen; e += blockDim.x) {
        const int v = e / vlen, r = e - v * vlen;
        const double val = Yrow[e];
        vl[v * p.vp + r] = (e == fd_e) ? val + p.fd_h : val;
    }
    for (int e = threadIdx.x; e < 3 * p.n_poly_pts; e += blockDim.x) pl[e] = p.poly[e];
    const int c0 = w * p.chunk, c1 = min(p.n_pairs, c0 + p.chunk);
    if (threadIdx.x == 0) s_next = c0;
    __syncthreads();
    const int polybase = p.n_veh * p.vp;
    const size_t obase = (size_t)b * p.n_pairs;
    gjk_sweep_any<MemLds, PLANAR>(
        c1, &s_next, p.max_iter, p.md_cap,
        [&](int k, Ctx<MemLds>& g) {
            const int a = p.pa[k], bb = p.pb[k];
            g.mem = MemLds{ lds };
            if (a < p.n_veh) g.P1 = Poly{ a * p.vp, p.nc, p.nc, p.dim == 3 };
            else { const int o = p.poly_off[a - p.n_veh], K = p.poly_off[a - p.n_veh + 1] - o; g.P1 = Poly{ polybase + 3 * o, K, K, 1 }; }
            if (bb < p.n_veh) g.P2 = Poly{ bb * p.vp, p.nc, p.nc, p.dim == 3 };
            else { const int o = p.poly_off[bb - p.n_veh], K = p.poly_off[bb - p.n_veh + 1] - o; g.P2 = Poly{ polybase + 3 * o, K, K, 1 }; }
            g.trace = nullptr; g.trace_cap = 0;
        },
        [&](int k, const Result& r) {
            const size_t o = obase + k;
            p.flag[o] = r.flag;
            p.p1[3 * o] = r.c1.x; p.p1[3 * o + 1] = r.c1.y; p.p1[3 * o + 2] = r.c1.z;
            p.p2[3 * o] = r.c2.x; p.p2[3 * o + 1] = r.c2.y; p.p2[3 * o + 2] = r.c2.z;
            p.dist[o] = r.dist;
            if (p.nsup) p.nsup[o] = r.n_support;
            if (p.status) p.status[o] = r.status;
        });
}

// -------------------------------------------------------------------------------------
//  Planar swarm sweep, fixed point count: the fast path of the 2-D configurations.
//  Every object of a row (vehicle hulls AND polygons) is staged in LDS as x[NC] y[NC] with one
//  odd pitch; polygons with fewer than NC vertices are padded with copies of their vertex 0,
//  which can never win the strict '>' of the support scan (gjk.py:109), so indices are
//  unchanged.  The scan is then a fully unrolled 2*NC-point loop with immediate LDS offsets.
//  Phase 1 runs the state machine and leaves a 16-byte record per pair in LDS (final simplex
//  as support indices); phase 2 turns records into closest points / distance with one lane
//  per pair, convergent and with coalesced output stores.
// -------------------------------------------------------------------------------------
template <int NC>
struct PlanarShape {
    static constexpr int VPQ = NC | 1;   // object pitch in (x, y) points of 16 bytes, odd
};

// LDS image of the staged objects as the general closest-point code addresses it (gjk_device.h
// `point`: coordinate c of point k of a set at `base` is mem(base + c*cs + k)): cs = 1 << 24
struct MemLdsXY {
    const double* l;
    __device__ __forceinline__ double operator()(int idx) const { return l[((idx & 0xffffff) << 1) | (idx >> 24)]; }
};

// The support scan over two staged objects.  Points are (x, y) pairs of 16 bytes, so one
// ds_read_b128 fetches a point (ds_read2_b64, which the compiler picks for two separate doubles,
// moves half the bytes per LDS cycle).  Value and index of the running maximum are tracked
// separately: `c > m` decides the index exactly as gjk.py:109 does, max() carries the value
// (equal up to the sign of zero, which no comparison sees).
template <int NC>
__device__ __forceinline__ void support_fixed(const double2* __restrict__ o1, const double2* __restrict__ o2,
                                              const gjk::V2& d, gjk::Vert2& out)
{
    const double ndx = -d.x, ndy = -d.y;
    int i1 = 0, i2 = 0;
    const double2 f1 = o1[0], f2 = o2[0];
    double m1 = f1.x * d.x + f1.y * d.y;
    double m2 = f2.x * ndx + f2.y * ndy;
#pragma unroll
    for (int i = 1; i < NC; ++i) {
        const double2 q1 = o1[i], q2 = o2[i];
        const double c1 = q1.x * d.x + q1.y * d.y;
        const double c2 = q2.x * ndx + q2.y * ndy;
        i1 = c1 > m1 ? i : i1;
        i2 = c2 > m2 ? i : i2;
        m1 = __builtin_fmax(m1, c1);
        m2 = __builtin_fmax(m2, c2);
    }
    out.ii = gjk::pack_ii(i1, i2);
    const double2 w1 = o1[i1], w2 = o2[i2];
    out.v = gjk::V2{ w1.x - w2.x, w1.y - w2.y };
}

// MODE 0: the sweep proper.  Workgroup w of a row owns the pairs [w*chunk, (w+1)*chunk) of the list;
//         every object of the row is staged, LDS slot == object id.
//         * Scheduling by history: gjkNew's trip count per pair varies 3..26 support scans, and a wave
//           pays for the refill path whenever ANY lane finishes.  The previous sweep leaves its
//           per-pair scan counts in p.len_in (SLSQP evaluates f(x) right before the finite-difference
//           batch around x, and consecutive iterates are close, so the counts carry over); each
//           workgroup counting-sorts its pairs by descending count, so the lanes of a wave start
//           pairs of equal length together, finish together, refill together, and the longest pairs
//           are not left for the tail.  Only the ORDER of evaluation changes: results are written
//           per pair and are identical for any order.
//         * XCD-aware ids: consecutive workgroup ids go round-robin to the 8 XCDs (one L2 each).  The
//           W workgroups of a row stage the same row and write neighbouring output ranges, so
//           id -> (row, w) keeps a row on one XCD: rows 8g .. 8g+7 take ids 8gW .. 8(g+1)W-1 with
//           row = 8g + id % 8, w = (id / 8) % W.
// MODE 1: finite-difference de-duplication pass (FIXUP), one workgroup per row b >= 1: only pairs
//         with a hull that differs from row 0 (mask p.chg) are evaluated; everything else was
//         filled in beforehand by k_bcast_row0.  The pair list is walked in segments of
//         p.chunk candidates; the changed ones are compacted into an LDS list.
// MODE 2: large rows (TILED).  The host has sorted the pair list tile-major (8 x 64 blocks of the
//         pair matrix) and cut it into chunks that touch at most p.max_objs objects; a workgroup
//         stages only its chunk's objects (p.cobjs) and the pairs carry LDS slots (p.pslots);
//         results go to the pairs' original positions (p.order).
template <int MODE>
__host__ __device__ constexpr size_t planar_lds_bytes(int cap_obj, int vpq, int chunk)
{
    // objects | rec int2[chunk], plist u32[chunk] | ext int[2 cap_obj] | MODE 1: list int[chunk]; MODE 0, 2: ord u16[chunk]
    // (14 bytes per pair for the sweeps: five workgroups of 1264 pairs and 72 objects per CU)
    return 16 * (size_t)cap_obj * vpq + 12 * (size_t)chunk + 8 * (size_t)cap_obj + (MODE == 1 ? 4 : 2) * (size_t)chunk;
}

#define OBTG_SWEEP_THREADS 256      // 512-thread workgroups (half as many stagings per row) measured no faster
// Occupancy: the sweep is sensitive to waves per SIMD (C3: 2 / 3 / 4 / 5 waves = 0.58 / 0.17 / 0.153 /
// 0.146 ms).  Five waves need <= 96 VGPRs (7 spilled at NC = 11) and <= 32 KB of LDS per workgroup,
// hence 864-pair chunks; six waves (80 VGPRs, 24 spilled) lose again.
constexpr int kSweepWavesPerSimd = 5;
// The one-launch pair sweep (separation block, dynamics groups in its tail) does better at FOUR waves per SIMD and two
// workgroups per C3 row: 128 VGPRs (no spills, also none in the dynamics groups), a third fewer stagings and sorts per
// row, 32-row transposition passes.  Interleaved runs on one box: 5 waves / 864-pair chunks 0.1872 ms, 4 / 864
// 0.1855, 4 / 1280 0.1819 -- while the plain gjkNew sweep still prefers five (0.129 against 0.142 ms).
// (Round 3 measured the one-launch form at five waves per SIMD again, now with 14-byte records so that 1264-pair
// chunks fit five times into a CU's LDS and ONE workgroup per row in two passes is one round of the chip: 0.183 - 0.187
// ms, the same as four waves -- but under the 96-VGPR bound the speed / angular-rate groups spill, and the PMC traffic
// of the launch goes from 630 MB to 722 MB against 647 MB algorithmic (profiles/r03_a_*).  Same speed, cleaner traffic: four.)
#ifndef OBTG_PS_WAVES
#define OBTG_PS_WAVES 4
#endif
constexpr int kPairSweepWavesPerSimd = OBTG_PS_WAVES;
// packed support indices (i1 | i2 << 16) as the 10-bit record form i1 | i2 << 5 (indices < 32 in the fixed-count sweeps)
__device__ __forceinline__ int rec5(int ii) { return (ii & 31) | ((ii >> 11) & (31 << 5)); }

// (b_in, w_in): row / workgroup-in-row when the caller has already decoded them (>= 0: the one-launch
// pair sweep), else decoded from blockIdx here.
// TS: the kernel also writes temporal-separation rows (the pair sweeps); the plain sweeps compile without that code
template <int NC, int MODE, bool TS = false>
__device__ __forceinline__ void gjk_planar_body(const GjkSwarmParams& p, double2* xy, const int b_in, const int w_in)
{
    constexpr bool SWEEP = MODE == 0, FIXUP = MODE == 1, TILED = MODE == 2;
    using gjk::V2;
    using gjk::Vert2;
    using gjk::Simplex2;
    constexpr int VPQ = PlanarShape<NC>::VPQ;
    double* lds = reinterpret_cast<double*>(xy);                   // xy: [cap_obj][VPQ] points
    // Issue priority: a workgroup's gjkNew phases (VALU bound, on the launch's critical path) run at priority 3, its
    // Bernstein phases (whose stores drain in the background) at 0, so a SIMD that holds both kinds of waves feeds the
    // state machines first.  Pair sweep 1 - 2.5 % faster in interleaved runs on one box (0.1736 -> 0.1720 ms, 0.1763 ->
    // 0.1718 ms on another); the reverse assignment changes nothing.
    if (TS) __builtin_amdgcn_s_setprio(3);
    __shared__ int s_next;
    __shared__ int s_nlist;
    __shared__ int s_hist[256];
    __shared__ unsigned short s_rowslot[32], s_colslot[kWave];   // TILED pair sweep: LDS slots of the tile's row / column vehicles
    int b, w;
    if (b_in >= 0) {
        b = b_in; w = w_in;
    } else if (SWEEP) {
        const int per = 8 * p.wgs_per_row;
        const int grp = (int)blockIdx.x / per, g = (int)blockIdx.x - grp * per;
        b = grp * 8 + (g & 7);
        w = g >> 3;
        if (b >= p.B) return;
    } else if (FIXUP) {
        b = (int)blockIdx.x + 1; w = 0;
    } else {
        b = (int)(blockIdx.x / p.wgs_per_row); w = (int)(blockIdx.x - b * p.wgs_per_row);
    }
    // c0 .. c1: positions this workgroup walks (MODE 0: local indices l of its current pass, pair k = cbase + l)
    int c0 = SWEEP ? 0 : (TILED ? p.chunk_off[w] : w * p.chunk);
    int cbase = 0;                                  // MODE 0: first pair of the chunk in hand (see the pass loop)
#define OWN(w_, l_, W_) (cbase + (l_))
    int c1 = SWEEP ? 0 : (TILED ? p.chunk_off[w + 1] : min(p.n_pairs, c0 + p.chunk));
    const int obj0 = TILED ? p.cobj_off[w] : 0;
    const int n_obj = TILED ? p.cobj_off[w + 1] - obj0 : p.n_veh + p.n_poly;     // staged objects
    const int n_obs_staged = (TS && SWEEP) ? p.n_obs : 0;                          // point obstacles of the separation rows
    const int cap_obj = TILED ? p.max_objs : n_obj + n_obs_staged;                 // LDS slots reserved
    // per-pair records of phase 1 -> phase 2 (8 bytes):  r01.x = (flag+1) | status << 2 | keys << 4 | n_support << 8,
    // r01.y = the final simplex as support indices, five bits each: A.i1 | A.i2 << 5 | B.i1 << 10 | B.i2 << 15 | C.i1 << 20 | C.i2 << 25
    static_assert(NC <= 32, "five-bit support indices in the phase-1 records");
    int2* r01 = reinterpret_cast<int2*>(xy + cap_obj * VPQ);
    unsigned* plist = reinterpret_cast<unsigned*>(r01 + p.chunk);   // [chunk] packed slots (a | b << 16) per position
    int* ext = reinterpret_cast<int*>(plist + p.chunk);            // [cap_obj][2]: (first argmax x, first argmin x)
    int* list = ext + 2 * cap_obj;                                  // FIXUP: compacted pair indices of a segment
    unsigned short* ord = reinterpret_cast<unsigned short*>(list);  // MODE 0, 2: position -> local index l

    // ---- stage vehicles (rows x, y of the evaluation row) and padded polygons
    const double* Yrow = p.fd ? p.Y : p.Y + (size_t)b * p.n_veh * 2 * NC;
    const int fd_e = fd_element(p.fd, p.fd_fixed, NC, b);   // element of the row that this evaluation row advances by fd_h
    // (rows x[NC], y[NC] in memory -> point-major (x, y) in LDS)
    if (TILED) {
        for (int e = threadIdx.x; e < n_obj * 2 * NC; e += blockDim.x) {
            const int sl = e / (2 * NC), r = e - sl * (2 * NC), q = r / NC, k = r - q * NC;
            const int obj = p.cobjs[obj0 + sl];
            if (TS && r == 0 && p.ts.out != nullptr) {
                const int2 ij = p.chunk_ij[w];
                if (ij.x >= 0) {
                    if (obj >= ij.x && obj < ij.x + p.tile_a) s_rowslot[obj - ij.x] = (unsigned short)sl;
                    if (obj >= ij.y && obj < ij.y + kWave) s_colslot[obj - ij.y] = (unsigned short)sl;
                }
            }
            double val;
            if (obj < p.n_veh) { const int ee = obj * 2 * NC + r; val = Yrow[ee]; if (ee == fd_e) val += p.fd_h; }
            else {
                const int o = obj - p.n_veh;
                const int off = p.poly_off[o], K = p.poly_off[o + 1] - off;
                val = p.poly[3 * off + q * K + (k < K ? k : 0)];
            }
            lds[2 * (sl * VPQ + k) + q] = val;
        }
    } else {
        for (int e = threadIdx.x; e < p.n_veh * 2 * NC; e += blockDim.x) {
            const int v = e / (2 * NC), r = e - v * (2 * NC), q = r / NC, k = r - q * NC;
            const double val = Yrow[e];
            lds[2 * (v * VPQ + k) + q] = (e == fd_e) ? val + p.fd_h : val;
        }
        for (int e = threadIdx.x; e < p.n_poly * 2 * NC; e += blockDim.x) {
            const int o = e / (2 * NC), r = e - o * (2 * NC), q = r / NC, k = r - q * NC;
            const int off = p.poly_off[o], K = p.poly_off[o + 1] - off;
            lds[2 * ((p.n_veh + o) * VPQ + k) + q] = p.poly[3 * off + q * K + (k < K ? k : 0)];
        }
        for (int e = threadIdx.x; e < n_obs_staged * 2 * NC; e += blockDim.x) {       // constant curves (optimization.py:86-98)
            const int o = e / (2 * NC), r = e - o * (2 * NC), q = r / NC, k = r - q * NC;
            lds[2 * ((n_obj + o) * VPQ + k) + q] = p.obs[o * 2 + q];
        }
    }
    __syncthreads();

    // ---- pair sweep: this workgroup's share of the row's temporal-separation block, from the objects
    // just staged.  Its stores are not waited for: they drain while the lanes run gjkNew below (the
    // Bernstein block is HBM-write bound, gjkNew VALU bound).  The transposition tile borrows the
    // LDS that phase 1 uses afterwards.
    // Workgroups differ in how much of their share they write before computing (0..3 groups per wave,
    // the rest afterwards), so that every CU has both kinds of work at all times.
    const int ts_before = (b + w) & 3;      // groups per wave written before the gjkNew phases (0..3), the rest after
    const int2 ts_ij = (TS && TILED && p.ts.out != nullptr) ? p.chunk_ij[w] : make_int2(-1, -1);
    if (TS && TILED && ts_ij.x >= 0 && ts_before > 0) {
        __builtin_amdgcn_s_setprio(0);
        tsep_tile_from_xy<NC>(p.ts, xy, VPQ, b, p.n_veh, ts_ij.x, ts_ij.y, p.tile_a, s_rowslot, s_colslot,
                              reinterpret_cast<double*>(r01), p.ts_tile_rows, 0, ts_before);
        __builtin_amdgcn_s_setprio(3);
        __syncthreads();
    }
    if (TS && SWEEP && p.ts.out != nullptr && ts_before > 0) {
        __builtin_amdgcn_s_setprio(0);
        tsep_groups_from_xy<NC>(p.ts, xy, VPQ, b, w, p.wgs_per_row, reinterpret_cast<double*>(r01), p.ts_tile_rows,
                                0, ts_before);
        __builtin_amdgcn_s_setprio(3);
        __syncthreads();
    }

    // ---- the first two doSimplex steps of every pair use the fixed directions (1,0,0) and
    // (-1,-0,-0) (gjk.py:247, 544): their support scans depend on one object only, so they are
    // done once per object and row.  x*1 + y*0 == x and x*(-1) + y*(-0) == -x exactly, hence
    // "first index of max x" / "first index of min x" are the reference's answers.
    for (int o = threadIdx.x; o < n_obj; o += blockDim.x) {
        const double2* q = xy + o * VPQ;
        int imx = 0, imn = 0;
        double mx = q[0].x * 1.0 + q[0].y * 0.0, mn = q[0].x * -1.0 + q[0].y * -0.0;
#pragma unroll
        for (int i = 1; i < NC; ++i) {
            const double c1 = q[i].x * 1.0 + q[i].y * 0.0, c2 = q[i].x * -1.0 + q[i].y * -0.0;
            if (c1 > mx) { mx = c1; imx = i; }
            if (c2 > mn) { mn = c2; imn = i; }
        }
        ext[2 * o] = imx; ext[2 * o + 1] = imn;
    }
    __syncthreads();
    const bool shortcut = p.max_iter >= 3 && p.md_cap >= 2;
    const unsigned char* chg = (FIXUP && !p.chg_from_fd) ? p.chg + (size_t)b * p.n_veh : nullptr;
    const int fd_veh = fd_e >= 0 ? fd_e / (2 * NC) : -1;          // the vehicle row b's advanced control point belongs to

    // MODE 0: a workgroup takes p.passes chunks of its row one after the other on the one staging of the row (fewer,
    // longer workgroups; see sweep_shape() for why that is not the default shape)
    const int fix_base = (FIXUP && p.chg_from_fd && fd_veh >= 0) ? p.vp_off[fd_veh] : 0;
    const int fix_count = FIXUP ? (p.chg_from_fd ? (fd_veh >= 0 ? p.vp_off[fd_veh + 1] - fix_base : 0) : p.n_pairs) : 0;
    for (int seg0 = 0; seg0 < (FIXUP ? fix_count : (SWEEP ? p.passes * p.chunk : 1)); seg0 += p.chunk) {
    if (SWEEP) {
        cbase = w * p.passes * p.chunk + seg0;
        c1 = max(0, min(p.n_pairs - cbase, p.chunk));
        if (c1 == 0) break;                               // uniform
    }
    if (FIXUP) {
        if (threadIdx.x == 0) s_nlist = 0;
        __syncthreads();
        const int seg1 = min(fix_count, seg0 + p.chunk);
        if (p.chg_from_fd) {
            // the pairs of the row's vehicle, straight from its list
            for (int q = seg0 + (int)threadIdx.x; q < seg1; q += blockDim.x) list[q - seg0] = p.vp_idx[fix_base + q];
            if (threadIdx.x == 0) s_nlist = seg1 - seg0;
        } else {
            for (int q = seg0 + (int)threadIdx.x; q < seg1; q += blockDim.x) {
                const int a = p.pa[q], bb = p.pb[q];
                if ((a < p.n_veh && chg[a]) || (bb < p.n_veh && chg[bb])) list[atomicAdd(&s_nlist, 1)] = q;
            }
        }
        __syncthreads();
        c0 = 0; c1 = s_nlist;
        __syncthreads();
        if (c1 == 0) continue;                            // uniform: nothing changed in this segment
    }
    // the chunk's (a, b) object ids go to LDS once: the refill path must not wait on global memory
    if (SWEEP || TILED) {
        // local index l = position - c0 in list order; pair id (history / output index) and packed slots of l
        const int n_loc = c1 - c0;
        auto pair_of = [&](int l) { return SWEEP ? OWN(w, l, 0) : p.order[c0 + l]; };
        auto slots_of = [&](int l, int kq) {
            return SWEEP ? ((unsigned)p.pa[kq] | ((unsigned)p.pb[kq] << 16)) : p.pslots[c0 + l];
        };
        if (p.len_in) {
            // counting sort of this workgroup's pairs by descending scan count of the previous sweep
            const unsigned char* len = p.len_in + (size_t)b * p.len_in_stride;
            if (threadIdx.x < 256) s_hist[threadIdx.x] = 0;
            __syncthreads();
            for (int l = threadIdx.x; l < n_loc; l += blockDim.x) atomicAdd(&s_hist[255 - (len[pair_of(l)] >> p.hist_shift)], 1);
            __syncthreads();
            if (threadIdx.x < kWave) {
                const int lane = threadIdx.x;
                const int v0 = s_hist[4 * lane], v1 = s_hist[4 * lane + 1], v2 = s_hist[4 * lane + 2], v3 = s_hist[4 * lane + 3];
                const int sum = v0 + v1 + v2 + v3;
                int incl = sum;
#pragma unroll
                for (int d = 1; d < kWave; d <<= 1) {
                    const int t = __shfl_up(incl, d);
                    if (lane >= d) incl += t;
                }
                const int excl = incl - sum;
                s_hist[4 * lane] = excl; s_hist[4 * lane + 1] = excl + v0;
                s_hist[4 * lane + 2] = excl + v0 + v1; s_hist[4 * lane + 3] = excl + v0 + v1 + v2;
            }
            __syncthreads();
            for (int l = threadIdx.x; l < n_loc; l += blockDim.x) {
                const int kq = pair_of(l);
                const int pos = atomicAdd(&s_hist[255 - (len[kq] >> p.hist_shift)], 1);
                ord[pos] = (unsigned short)l;
                plist[pos] = slots_of(l, kq);
            }
        } else {
            for (int l = threadIdx.x; l < n_loc; l += blockDim.x) {
                ord[l] = (unsigned short)l;
                plist[l] = slots_of(l, pair_of(l));
            }
        }
    } else {
        for (int q = c0 + (int)threadIdx.x; q < c1; q += blockDim.x) {
            const int kq = list[q];
            plist[q - c0] = (unsigned)p.pa[kq] | ((unsigned)p.pb[kq] << 16);
        }
    }
    if (threadIdx.x == 0) { s_next = c0; if (SWEEP || TILED) s_nlist = 0; }
    __syncthreads();

    // ---- phase 1: state machine with lane refill
    {
        const int lane = threadIdx.x & (kWave - 1);
        int k = -1, slot = 0;
        bool exhausted = false;
        const double2* o1 = xy;
        const double2* o2 = xy;
        Simplex2 s, old;
        V2 dir{ 1.0, 0.0 };
        int phase = 0, it = 0, rr = 0, nsup = 0;
        s.keys = 0;
        s.A = Vert2{ V2{ 0, 0 }, 0 };
        s.B = s.A; s.C = s.A; old = s;
        for (;;) {
            const unsigned long long want = __ballot(k < 0 && !exhausted);
            // Refill in batches: the refill path and the two-point step that every fresh pair starts with are paid by
            // the whole wave whenever ANY lane refills, so idle lanes wait until p.refill_min of them are idle (or
            // nothing else runs); see tools/refill_sim.py for the trade against idle lane-rounds.
            if (want && (__popcll(want) >= p.refill_min || __ballot(k >= 0) == 0ull)) {
                const int leader = __ffsll((long long)want) - 1;
                int base = 0;
                if (lane == leader) base = atomicAdd(&s_next, __popcll(want));
                base = __shfl(base, leader);
                if (k < 0 && !exhausted) {
                    const int my = base + __popcll(want & ((1ull << lane) - 1ull));
                    if (my < c1) {
                        k = my;
                        slot = (SWEEP || TILED) ? (int)ord[k - c0] : k - c0;
                        const unsigned ab = plist[k - c0];
                        const int a = (int)(ab & 0xffffu), bb = (int)(ab >> 16);
                        o1 = xy + a * VPQ;
                        o2 = xy + bb * VPQ;
                        s.keys = 0; dir = V2{ 1.0, 0.0 };
                        phase = 0; it = 0; rr = 0; nsup = 0;
                        if (shortcut) {
                            // state after doSimplex #1 (0pt) and #2 (1pt)
                            const int ax = ext[2 * a], an = ext[2 * a + 1], bx = ext[2 * bb], bn = ext[2 * bb + 1];
                            const double2 pax = o1[ax], pbn = o2[bn], pan = o1[an], pbx = o2[bx];
                            const Vert2 A1{ V2{ pax.x - pbn.x, pax.y - pbn.y }, gjk::pack_ii(ax, bn) };
                            const Vert2 A2{ V2{ pan.x - pbx.x, pan.y - pbx.y }, gjk::pack_ii(an, bx) };
                            const bool md1 = gjk::dotb2(A1.v, dir) < 0;       // exit of iteration 1 (gjk.py:260)
                            dir = gjk::neg2(dir);
                            s.B = A1; s.A = A2; s.keys = gjk::kA | gjk::kB;
                            nsup = 2;
                            if (!md1) {
                                it = 2;
                                if (gjk::dotb2(A2.v, dir) < 0) { phase = 1; rr = 0; }
                            } else {
                                // iteration 2 ran inside minimumDistance with old = {A1}
                                phase = 1; rr = 1;
                                const bool m = gjk::eq2(A2.v, A1.v) ||
                                    (A1.v.x == 0.0 && A1.v.y == 0.0 && A2.v.x == pax.x && A2.v.y == pax.y &&
                                     A2.v.x == pbn.x && A2.v.y == pbn.y);
                                if (m) {
                                    r01[slot] = make_int2((1 + 1) | (OBTG_ST_OK << 2) | (gjk::kA << 4) | (nsup << 8),
                                                          rec5(A1.ii));
                                    k = -1;
                                }
                            }
                        }
                    } else exhausted = true;
                }
            }
            if (__ballot(k >= 0) == 0ull) break;
            if (k >= 0) {
                if (phase == 1) old = s;
                gjk::simplex_update2(s, dir);
                support_fixed<NC>(o1, o2, dir, s.A);
                s.keys |= gjk::kA;
                ++nsup;
                int flag = -2, status = OBTG_ST_OK;       // -2: not finished
                if (phase == 0) {
                    ++it;
                    if (s.keys & gjk::kColl) flag = 0;
                    else if (gjk::dotb2(s.A.v, dir) < 0) { phase = 1; rr = 0; }
                    else if (it >= p.max_iter) { flag = -1; status = OBTG_ST_MAXITER; }
                } else {
                    ++rr;
                    // converged iff the new A equals any value of the old dict (gjk.py:281-294)
                    bool m = false;
                    if ((old.keys & gjk::kA) && (gjk::eq2(s.A.v, old.A.v) ||
                        (old.A.v.x == 0.0 && old.A.v.y == 0.0 && s.A.v.x == o1[old.A.i1()].x && s.A.v.y == o1[old.A.i1()].y &&
                         s.A.v.x == o2[old.A.i2()].x && s.A.v.y == o2[old.A.i2()].y))) m = true;
                    if ((old.keys & gjk::kB) && (gjk::eq2(s.A.v, old.B.v) ||
                        (old.B.v.x == 0.0 && old.B.v.y == 0.0 && s.A.v.x == o1[old.B.i1()].x && s.A.v.y == o1[old.B.i1()].y &&
                         s.A.v.x == o2[old.B.i2()].x && s.A.v.y == o2[old.B.i2()].y))) m = true;
                    if ((old.keys & gjk::kC) && (gjk::eq2(s.A.v, old.C.v) ||
                        (old.C.v.x == 0.0 && old.C.v.y == 0.0 && s.A.v.x == o1[old.C.i1()].x && s.A.v.y == o1[old.C.i1()].y &&
                         s.A.v.x == o2[old.C.i2()].x && s.A.v.y == o2[old.C.i2()].y))) m = true;
                    if (m) flag = 1;
                    else if (rr >= p.md_cap) { flag = 1; status = OBTG_ST_MD_CAP; }
                }
                if (flag != -2) {
                    r01[slot] = make_int2((flag + 1) | (status << 2) | ((old.keys & 7) << 4) | (min(nsup, 0xffffff) << 8),
                                          rec5(old.A.ii) | (rec5(old.B.ii) << 10) | (rec5(old.C.ii) << 20));
                    k = -1;
                }
            }
        }
    }
    __syncthreads();

    // ---- phase 2: closest points / distance from the recorded simplices (gjk.py:299-360)
    // The rare exit "origin over the triangle's interior" needs the general 3-D evaluation (about three times the
    // instructions of the other exits together); about one pair in a hundred takes it, i.e. every other wave would
    // pay for it.  Those pairs are set aside (their local index goes to `ord`, which phase 1 no longer needs) and
    // evaluated afterwards, densely packed.
    constexpr bool DEFER = SWEEP || TILED;
    const size_t obase = (size_t)b * p.n_pairs;
    const double qnan = __builtin_nan("");
    auto emit = [&](int kk, int flag, int status, int n_scans, const Result& r) {
        const size_t o = obase + kk;
        p.flag[o] = flag;
        // 24-byte records: one 16-byte (8-byte aligned) and one 8-byte store each
        typedef double d2u_t __attribute__((ext_vector_type(2), aligned(8)));
        if (p.emit_scalar) {
            p.p1[3 * o] = r.c1.x; p.p1[3 * o + 1] = r.c1.y; p.p1[3 * o + 2] = r.c1.z;
            p.p2[3 * o] = r.c2.x; p.p2[3 * o + 1] = r.c2.y; p.p2[3 * o + 2] = r.c2.z;
        } else {
            d2u_t xy1, xy2;
            xy1.x = r.c1.x; xy1.y = r.c1.y; xy2.x = r.c2.x; xy2.y = r.c2.y;
            *reinterpret_cast<d2u_t*>(p.p1 + 3 * o) = xy1; p.p1[3 * o + 2] = r.c1.z;
            *reinterpret_cast<d2u_t*>(p.p2 + 3 * o) = xy2; p.p2[3 * o + 2] = r.c2.z;
        }
        p.dist[o] = r.dist;
        if (p.nsup) p.nsup[o] = n_scans;
        if (p.status) p.status[o] = status;
        if ((SWEEP || TILED) && p.len_out) p.len_out[o] = (unsigned char)min(n_scans, 255);
    };
    auto pair_id = [&](int k) { return SWEEP ? OWN(w, k, p.wgs_per_row) : (TILED ? p.order[k] : (FIXUP ? list[k] : k)); };
    // the general evaluation of a recorded three-point simplex
    // slots of local index l in list order (phase 2 is convergent: the sweeps read them from the lists again instead of
    // keeping a second LDS copy)
    auto slots_nat = [&](int l) {
        if (SWEEP) { const int kq = cbase + l; return (unsigned)p.pa[kq] | ((unsigned)p.pb[kq] << 16); }
        if (TILED) return p.pslots[c0 + l];
        return plist[l];
    };
    auto general_exit = [&](int sa, int sb, int keys, int rq1, Result& r) {
        const int ia1 = rq1 & 31, ia2 = (rq1 >> 5) & 31, ib1 = (rq1 >> 10) & 31, ib2 = (rq1 >> 15) & 31;
        const int ic1 = (rq1 >> 20) & 31, ic2 = (rq1 >> 25) & 31;
        Ctx<MemLdsXY> g;
        g.mem = MemLdsXY{ lds };
        g.P1 = Poly{ sa * VPQ, 1 << 24, NC, 0 };
        g.P2 = Poly{ sb * VPQ, 1 << 24, NC, 0 };
        g.trace = nullptr; g.trace_cap = 0; g.n_support = 0;
        gjk::Simplex s;
        s.keys = keys;
        s.A = gjk::Vert{ gjk::sub(gjk::point(g.mem, g.P1, ia1), gjk::point(g.mem, g.P2, ia2)), ia1, ia2 };
        s.B = gjk::Vert{ gjk::sub(gjk::point(g.mem, g.P1, ib1), gjk::point(g.mem, g.P2, ib2)), ib1, ib2 };
        s.C = gjk::Vert{ gjk::sub(gjk::point(g.mem, g.P1, ic1), gjk::point(g.mem, g.P2, ic2)), ic1, ic2 };
        s.D = s.A;
        gjk::closest_from_simplex(g, s, r);
    };
    for (int k = c0 + (int)threadIdx.x; k < c1; k += blockDim.x) {
        const int kk = pair_id(k);
        const unsigned ab2 = slots_nat(k - c0);
        const int sa = (int)(ab2 & 0xffffu), sb = (int)(ab2 >> 16);
        const int2 rq = r01[k - c0];
        const int rq0 = rq.x, rq1 = rq.y;
        const int flag = (rq0 & 3) - 1, status = (rq0 >> 2) & 3, keys = (rq0 >> 4) & 7, n_scans = (int)((unsigned)rq0 >> 8);
        Result r;
        r.c1 = V3{ qnan, qnan, qnan }; r.c2 = r.c1; r.dist = qnan;
        if (flag == 1 && status == OBTG_ST_OK) {
            // planar restatement of gjk.py:299-360 (z terms are exact zeros, see gjk_device.h):
            // pick the partner vertex O of the closest feature, then ONE segment evaluation
            const double2* q1 = xy + sa * VPQ;
            const double2* q2 = xy + sb * VPQ;
            const int ia1 = rq1 & 31, ia2 = (rq1 >> 5) & 31, ib1 = (rq1 >> 10) & 31, ib2 = (rq1 >> 15) & 31;
            const int ic1 = (rq1 >> 20) & 31, ic2 = (rq1 >> 25) & 31;
            const V2 a1{ q1[ia1].x, q1[ia1].y }, a2{ q2[ia2].x, q2[ia2].y };
            const V2 A = gjk::sub2(a1, a2);
            int which = 0;                    // 0: point A, 1: segment A-B, 2: segment A-C, 3: plane
            if (keys & gjk::kC) {
                const V2 b1{ q1[ib1].x, q1[ib1].y }, b2{ q2[ib2].x, q2[ib2].y };
                const V2 c1{ q1[ic1].x, q1[ic1].y }, c2{ q2[ic2].x, q2[ic2].y };
                const V2 B = gjk::sub2(b1, b2), C = gjk::sub2(c1, c2);
                const V2 A0 = gjk::neg2(A), AB = gjk::sub2(B, A), AC = gjk::sub2(C, A);
                const double w = gjk::cz(AB, AC);
                const V2 t1{ -(w * AC.y), w * AC.x };
                const V2 t2{ AB.y * w, -(AB.x * w) };
                which = (gjk::dotb2(t1, A0) >= 0) ? 2 : ((gjk::dotb2(t2, A0) >= 0) ? 1 : 3);
            } else if (keys & gjk::kB) which = 1;
            if (which != 3) {
                // point A (which == 0) or segment A-O: one radicand per lane, ONE square root for the wave
                // (the three exits of gjk.py:343-358 / 397-437 differ in what is under the root only)
                double t = 0.0, rad = gjk::dotb2(A, A);                  // np.linalg.norm(A) (gjk.py:352)
                V2 o1 = a1, o2 = a2;
                if (which != 0) {
                    const int io1 = which == 2 ? ic1 : ib1, io2 = which == 2 ? ic2 : ib2;
                    o1 = V2{ q1[io1].x, q1[io1].y }; o2 = V2{ q2[io2].x, q2[io2].y };
                    const V2 O = gjk::sub2(o1, o2);
                    rad = gjk::dot2(A, A);                                  // identical points (gjk.py:417-419)
                    if (!gjk::eq2(A, O)) {                                  // weightedOriginToLine (gjk.py:397-437)
                        const V2 v = gjk::sub2(O, A);
                        t = -gjk::dot2(v, A) / gjk::dot2(v, v);
                        if (t > 1) t = 1; else if (t < 0) t = 0;
                        const V2 cp{ (1 - t) * A.x + t * O.x, (1 - t) * A.y + t * O.y };
                        rad = gjk::dot2(cp, cp);
                    }
                }
                r.dist = __builtin_sqrt(rad);
                if (which == 0) {
                    r.c1 = V3{ a1.x, a1.y, 0.0 };
                    r.c2 = V3{ a2.x, a2.y, 0.0 };
                } else {
                    r.c1 = V3{ (1 - t) * a1.x + t * o1.x, (1 - t) * a1.y + t * o1.y, 0.0 };
                    r.c2 = V3{ (1 - t) * a2.x + t * o2.x, (1 - t) * a2.y + t * o2.y, 0.0 };
                }
            } else if (DEFER) {
                ord[atomicAdd(&s_nlist, 1)] = (unsigned short)(k - c0);
                continue;
            } else {
                general_exit(sa, sb, keys, rq1, r);
            }
        }
        emit(kk, flag, status, n_scans, r);
    }
    if (DEFER) {
        __syncthreads();
        const int n_def = s_nlist;
        for (int q = (int)threadIdx.x; q < n_def; q += blockDim.x) {
            const int l = (int)ord[q], k = c0 + l;
            const unsigned ab2 = slots_nat(l);
            const int2 rq = r01[l];
            Result r;
            r.c1 = V3{ qnan, qnan, qnan }; r.c2 = r.c1; r.dist = qnan;
            general_exit((int)(ab2 & 0xffffu), (int)(ab2 >> 16), (rq.x >> 4) & 7, rq.y, r);
            emit(pair_id(k), 1, OBTG_ST_OK, (int)((unsigned)rq.x >> 8), r);
        }
    }
    if (FIXUP || SWEEP) __syncthreads();                 // records / lists are reused by the next segment or pass
    }
    if (TS && TILED && ts_ij.x >= 0) {
        __syncthreads();                                     // phase 2 has read the records the tile overwrites
        __builtin_amdgcn_s_setprio(0);
        tsep_tile_from_xy<NC>(p.ts, xy, VPQ, b, p.n_veh, ts_ij.x, ts_ij.y, p.tile_a, s_rowslot, s_colslot,
                              reinterpret_cast<double*>(r01), p.ts_tile_rows, ts_before, 1 << 30);
    }
    if (TS && SWEEP && p.ts.out != nullptr) {
        __syncthreads();                                     // phase 2 has read the records the tile overwrites
        __builtin_amdgcn_s_setprio(0);
        tsep_groups_from_xy<NC>(p.ts, xy, VPQ, b, w, p.wgs_per_row, reinterpret_cast<double*>(r01), p.ts_tile_rows,
                                ts_before, 1 << 30);
    }
}

template <int NC, int MODE>
__global__ __launch_bounds__(MODE == 0 ? OBTG_SWEEP_THREADS : 256,
                             (MODE == 0 && NC <= 11) ? kSweepWavesPerSimd : (MODE == 2 ? 4 : 1))   // tiled: four waves per SIMD (tile_height() counts on it)
void k_gjk_swarm_planar(const GjkSwarmParams p)
{
    extern __shared__ double2 xy_dyn[];
    gjk_planar_body<NC, MODE>(p, xy_dyn, -1, -1);
}

// Diagnostics (OBTG_TIMELINE=<file>, launch_pair_sweep): when each workgroup of a launch started and when its last wave
// left, on the constant 100 MHz clock, and where it ran (HW_ID: CU / SE, XCC_ID) -- the picture of the grid's rounds
// and of its tail.  One scalar branch per workgroup when off.
struct TimelineScope {
    unsigned long long* t;
    __device__ __forceinline__ explicit TimelineScope(unsigned long long* base) : t(base)
    {
        if (t) {
            t += 4 * (size_t)blockIdx.x;
            if (threadIdx.x == 0) {
                t[0] = wall_clock64();
                t[2] = (unsigned long long)__builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | (31 << 11)) |
                       ((unsigned long long)__builtin_amdgcn_s_getreg((20 /*XCC_ID*/) | (0 << 6) | (31 << 11)) << 32);
            }
        }
    }
    __device__ __forceinline__ ~TimelineScope()
    {
        if (t && (threadIdx.x & (kWave - 1)) == 0) atomicMax(t + 1, wall_clock64());
    }
};

// the same grid when it also writes the temporal-separation blocks (p.ts.out set): own symbol, so that
// profiles tell the pair sweep from the plain GJK sweep.
// With p.dyn set the launch is the whole constraint evaluation of the batch: its last workgroups run the speed /
// angular-rate groups (see below).  Round 1 tried a dynamics workgroup per ROW inside the rows' id range: under the
// 96-VGPR bound the inlined body cost the gjkNew loop 16 scratch accesses (0.268 ms for the launch; out of line:
// 0.283 ms) against 0.183 + 0.024 ms as two launches.
template <int NC>
__global__ __launch_bounds__(OBTG_SWEEP_THREADS, NC <= 11 ? kPairSweepWavesPerSimd : 1)
void k_pair_sweep(const GjkSwarmParams p)
{
    extern __shared__ double2 xy_dyn[];
    TimelineScope tl(p.timeline);
    if (p.dyn.out != nullptr && (int)blockIdx.x >= p.dyn_first_block) {
        // the speed / angular-rate evaluation of the batch, one 64-vehicle group per workgroup, on the grid's last
        // block ids: they are dispatched as the sweep's workgroups drain and fill the slots the tail leaves empty
        // (on the FIRST block ids the launch takes 0.201 instead of 0.187 ms).  Under the sweep's 96-VGPR bound this
        // body spills (43 dwords); the gjkNew loop is not touched by it (no scratch access at all in this build).
        // (two waves work, two leave: with all four on the group -- dynamics2_group<.., W4 = true>, each squaring half the
        // coefficients -- a workgroup is shorter but the degree-2n curves are formed twice, and the launch is bound by the
        // chip's aggregate issue rate: 0.1820 against 0.1806 ms)
        // (16 control points: the two-wave form spills 800 bytes per lane under this kernel's 128 VGPRs, the four-wave form
        // half of that: C4 25.9 against 26.3 ms)
        if (NC <= 11 && threadIdx.x >= 2 * kWave) return;
        dynamics2_group<NC, true, (NC > 11)>(p.dyn, reinterpret_cast<double*>(xy_dyn), (int)blockIdx.x - p.dyn_first_block);
        return;
    }
    gjk_planar_body<NC, 0, true>(p, xy_dyn, -1, -1);
}

// -------------------------------------------------------------------------------------
//  Structured finite-difference step in ONE launch (SURVEY.md 8(f) item 1 for the whole step; obtg_constraint_sweep_fd_structured_dev).
//  The rows of an SLSQP finite-difference batch are ONE row of control points with one element advanced: row b >= 1
//  differs from row 0 in one vehicle, so all but N-1 of its separation pairs, N-1+M of its hull pairs and one of its
//  vehicles repeat row 0's results bit for bit.  Inside an FD view the rows are DEFINED that way (row b = the view's row
//  with its (b-1)-th free control point advanced), so which entries repeat is known without comparing anything, and the
//  step becomes four kinds of workgroups in one grid with no dependency between them:
//    S  (group g, row range): evaluates the 64-pair group g of ROW 0's separation block once and streams it into every
//       row of its range, leaving out the entries of pairs that contain the row's own vehicle;
//    G  (chunk c, row range): runs gjkNew on chunk c of ROW 0's hull pairs (results in LDS) and streams flag / closest
//       points / distance / status (/ scan count) into every row of its range, with the same exception;
//    F  (row b >= 1): stages row b, evaluates exactly the pairs the streams left out -- the hull pairs of its vehicle
//       (the de-duplication pass of the plain sweep, its mask computed from b) and that vehicle's separation rows;
//    D  the speed / angular-rate rows, by the same rule per vehicle: row 0's 64-vehicle groups streamed into row ranges,
//       the advanced vehicle of every row b >= 1 (64 of them to a group), and -- tf is an input per row -- every row whose
//       tf is not bit for bit tf[0] in full (bern_device.h DynEmit).
//  Every output element is written exactly once, by the workgroup kind that owns it: the result equals the brute-force
//  sweep's bit for bit (same device functions per pair), the launch is bound by its 611 MB of stores.
// -------------------------------------------------------------------------------------
struct StructuredParams {
    GjkSwarmParams g;                  // the sweep's parameters inside the view (g.fd set, g.Y = the view's row)
    int n_sep_groups, sep_rows_per;    // S: groups of the separation block, rows per range
    int gjk_chunks, gjk_chunk_pairs, gjk_rows_per;   // G
    int fix_chunk;                     // F: candidates per segment of the fix-up pass
    int n_kind[4];                     // workgroups of each kind (S, F, G, D).  The grid interleaves them: of every 16 block ids,
    int per16[4];                      // per16[k] belong to kind k, in the order pat[] (host: shares by expected work), so that
    unsigned char pat[16], rank[16];   // streams (HBM bound) and state machines (latency bound) share the CUs all along
    const double* cv4;                 // DEG_ELEV > 0: the dynamics groups' elevation tables (AngElevParams) and DEG_ELEV
    const double* cv2;
    int R;
    int dyn_groups_per_row, dyn_rows_per, dyn_streams, dyn_fix_groups;   // D: see the kernel
    int dyn_split;                     // D, DEG_ELEV > 0: stream workgroups of 16 vehicles (DynEmit::sub) instead of 64
    int first_pert;                    // the view's first perturbed local row: 1 (the view starts at the batch's row 0) or 0 (a later row range)
};

// ELEV: DEG_ELEV > 0 -- the separation groups are elevated (tsep_elev_group_stream), the fix-up rows too, and the
// dynamics groups are k_dynamics_elev's.  Two workgroups per CU: the S, F and G kinds need 121 / 120 / 82 registers, but the
// dynamics groups' row-mapped emission spills 80 dwords under the 168 of three per CU, and the launch is slower for it
// (C5, one box, interleaved: 0.699 against 0.564 ms; round 3's form of the S and F kinds at two per CU: 0.73-0.79).
// WPC: workgroups per CU the register budget is set for (4: rows of up to 40 KB of LDS; 1: large rows -- C4's 256 vehicles
// of 16 points are 70 KB -- where LDS leaves room for one workgroup anyway and the streams dominate).
template <int NC, bool ELEV, int WPC = (ELEV ? 2 : 4)>
__global__ __launch_bounds__(256, WPC) void k_step_fd_structured(const StructuredParams sp)
{
    extern __shared__ double2 xy_dyn[];
    constexpr int VPQ = PlanarShape<NC>::VPQ;
    constexpr int L = 2 * NC - 1;
    const GjkSwarmParams& p = sp.g;
    TimelineScope tl(p.timeline);
    const int n_obj = p.n_veh + p.n_poly;
    double* lds = reinterpret_cast<double*>(xy_dyn);
    // kind (0 S, 1 F, 2 G, 3 D) and index inside the kind from the block id
    // (block id mod 8 is the XCD: the pattern is rotated by one slot per group of 16, or every XCD would see one or two
    // kinds only and the XCD with the dearest kind would finish last)
    const int grp = (int)blockIdx.x >> 4, slot = ((int)blockIdx.x + grp) & 15;
    const int kind = sp.pat[slot];
    const int id = grp * sp.per16[kind] + sp.rank[slot];
    if (id >= sp.n_kind[kind]) return;
    if (kind == 3) {
        // ---- D: row 0's vehicle groups streamed into row ranges | the advanced vehicles, 64 rows to a group | rows whose
        //      tf is not tf[0] (a finite-difference row of tf itself), eight rows to a workgroup, in full
        __shared__ int s_dmap[kWave];
        const int gd = sp.dyn_groups_per_row;
        DynEmit em{ 0, 0, 0, 0, 0, s_dmap };
        // base_row: the group's items are the vehicles of the view's UNPERTURBED row (the streams' source) -- inside a row
        // range that is not the view's local row 0, so the items are read as a materialised one-row batch (fd = 0) while the
        // emission maps keep the view's rows (DynEmit::fd_map)
        auto run = [&](int group, bool base_row = false) {
            if (ELEV) {
                AngElevParams q;
                q.a = p.dyn; q.cv4 = sp.cv4; q.cv2 = sp.cv2; q.R = sp.R; q.flags = nullptr;
                if (base_row) { em.fd_map = q.a.fd; q.a.fd = 0; }
                dynamics_elev_group<NC>(q, lds, group, &em);
            } else {
                AngParams a = p.dyn;
                if (base_row) { em.fd_map = a.fd; a.fd = 0; }
                dynamics2_group<NC, false, true>(a, lds, group, &em);
            }
        };
        if (id < sp.dyn_streams) {
            // ELEV: a stream workgroup takes 16 of the group's vehicles (DynEmit::sub) and four times the rows
            const bool split = ELEV && sp.dyn_split;
            const int gs = split ? 4 * gd : gd;
            const int r = id / gs, u = id - r * gs;
            em.mode = 2; em.item_begin = 0; em.item_end = p.n_veh;
            em.b0 = r * sp.dyn_rows_per; em.b1 = min(p.B, em.b0 + sp.dyn_rows_per);
            if (split) {
                em.sub = u & 3;
                if (64 * (u >> 2) + 16 * em.sub >= p.n_veh) return;      // (a last group with fewer than 64 vehicles)
                run(u >> 2, true);
            } else run(u, true);
            return;
        }
        if (id < sp.dyn_streams + sp.dyn_fix_groups) {
            em.mode = 1; em.item_begin = sp.first_pert; em.item_end = p.B - sp.first_pert;
            run(id - sp.dyn_streams);
            return;
        }
        const int x = id - sp.dyn_streams - sp.dyn_fix_groups, xr = x / gd, gv = x - xr * gd;
        for (int b = 1 + 8 * xr; b < min(p.B, 9 + 8 * xr); ++b) {
            if (same_bits(p.dyn.tf[b], p.dyn.tf[0])) continue;
            em.mode = 0; em.item_begin = b * p.n_veh; em.item_end = em.item_begin + p.n_veh;
            __syncthreads();                                 // the tile of the previous row has been read out
            run(gv);
        }
        return;
    }
    if (kind == 1) {
        // ---- F: row b's own pairs
        const int b = id + sp.first_pert;
        GjkSwarmParams q = p;
        q.chunk = sp.fix_chunk; q.chg_from_fd = 1; q.passes = 1; q.len_in = nullptr; q.len_out = nullptr;
        gjk_planar_body<NC, 1, false>(q, xy_dyn, b, 0);
        __syncthreads();
        if (p.n_obs > 0) {             // point obstacles: constant curves behind the hull objects, where the body's records were
            for (int e = threadIdx.x; e < p.n_obs * 2 * NC; e += blockDim.x) {
                const int o = e / (2 * NC), r = e - o * (2 * NC), qd = r / NC, k = r - qd * NC;
                lds[2 * ((n_obj + o) * VPQ + k) + qd] = p.obs[o * 2 + qd];
            }
            __syncthreads();
        }
        const int fd_e = fd_element(p.fd, p.fd_fixed, NC, b);
        if (fd_e >= 0) {
            if (ELEV) {
                // coefficient image [32][PA] and output tile behind the staged objects (the body's records are dead)
                double* img = reinterpret_cast<double*>(xy_dyn + (n_obj + p.n_obs) * VPQ);
                tsep_elev_rows_of_vehicle<NC>(p.ts, xy_dyn, VPQ, b, p.n_veh, fd_e / (2 * NC), img, img + 32 * ElevMfma<L>::PA);
            } else tsep_rows_of_vehicle<NC>(p.ts, xy_dyn, VPQ, b, p.n_veh, fd_e / (2 * NC));
        }
        return;
    }
    if (kind == 2) {
        // ---- G: chunk c of row 0's hull pairs, streamed into the rows of range r
        const int t = id, c = t % sp.gjk_chunks, r = t / sp.gjk_chunks;
        const int cp = sp.gjk_chunk_pairs;
        const int k0 = c * cp, n_valid = max(0, min(cp, p.n_pairs - k0));
        // results in LDS behind the body's own arrays
        char* base = reinterpret_cast<char*>(xy_dyn) + ((planar_lds_bytes<0>(n_obj, VPQ, cp) + 15) / 16) * 16;
        double* r_p1 = reinterpret_cast<double*>(base);
        double* r_p2 = r_p1 + 3 * cp;
        double* r_dist = r_p2 + 3 * cp;
        int* r_flag = reinterpret_cast<int*>(r_dist + cp);
        int* r_stat = r_flag + cp;
        int* r_nsup = r_stat + cp;
        GjkSwarmParams q = p;
        q.fd = 0; q.B = 1; q.chunk = cp; q.passes = 1; q.wgs_per_row = sp.gjk_chunks; q.len_in = nullptr; q.len_out = nullptr;
        q.emit_scalar = 1;
        q.flag = r_flag - k0; q.p1 = r_p1 - 3 * (size_t)k0; q.p2 = r_p2 - 3 * (size_t)k0; q.dist = r_dist - k0;
        q.status = r_stat - k0; q.nsup = r_nsup - k0;
        gjk_planar_body<NC, 0, false>(q, xy_dyn, 0, c);
        __syncthreads();
        const int b0 = r * sp.gjk_rows_per, b1 = min(p.B, b0 + sp.gjk_rows_per);
        // every thread streams one pair into every n_sub-th row of the range
        const int n_sub = max(1, (int)blockDim.x / max(n_valid, 1));
        for (int t = (int)threadIdx.x; t < n_valid * n_sub; t += (int)blockDim.x) {
            const int l = t % n_valid, sub = t / n_valid;
            const int kq = k0 + l;
            const int a = p.pa[kq], bb = p.pb[kq];
            const int fl = r_flag[l], st = r_stat[l], ns = r_nsup[l];
            const double di = r_dist[l];
            const double p1x = r_p1[3 * l], p1y = r_p1[3 * l + 1], p1z = r_p1[3 * l + 2];
            const double p2x = r_p2[3 * l], p2y = r_p2[3 * l + 1], p2z = r_p2[3 * l + 2];
            for (int b = b0 + sub; b < b1; b += n_sub) {
                const int fd_e = fd_element(p.fd, p.fd_fixed, NC, b);
                const int vb = fd_e >= 0 ? fd_e / (2 * NC) : -1;
                if (a == vb || bb == vb) continue;                   // row b's own workgroup evaluates this pair
                const size_t o = (size_t)b * p.n_pairs + kq;
                p.flag[o] = fl;
                p.p1[3 * o] = p1x; p.p1[3 * o + 1] = p1y; p.p1[3 * o + 2] = p1z;
                p.p2[3 * o] = p2x; p.p2[3 * o + 1] = p2y; p.p2[3 * o + 2] = p2z;
                p.dist[o] = di;
                if (p.nsup) p.nsup[o] = ns;
                if (p.status) p.status[o] = st;
            }
        }
        return;
    }
    // ---- S: group g of row 0's separation block, streamed into the rows of range r
    {
        const int g = id % sp.n_sep_groups, r = id / sp.n_sep_groups;
        for (int e = threadIdx.x; e < p.n_veh * 2 * NC; e += blockDim.x) {
            const int v = e / (2 * NC), rr = e - v * (2 * NC), qd = rr / NC, k = rr - qd * NC;
            lds[2 * (v * VPQ + k) + qd] = p.Y[e];
        }
        for (int e = threadIdx.x; e < p.n_obs * 2 * NC; e += blockDim.x) {     // point obstacles: slot == object id here (no polygons staged)
            const int o = e / (2 * NC), rr = e - o * (2 * NC), qd = rr / NC, k = rr - qd * NC;
            lds[2 * ((p.n_veh + o) * VPQ + k) + qd] = p.obs[o * 2 + qd];
        }
        __syncthreads();
        if (ELEV) {
            // the group's coefficient image [64][PA] behind the staged row, the 16-row output tile behind that
            double* img = reinterpret_cast<double*>(xy_dyn + (p.n_veh + p.n_obs) * VPQ);
            tsep_elev_group_stream<NC>(p.ts, xy_dyn, VPQ, g, img, img + kWave * ElevMfma<L>::PA, r * sp.sep_rows_per,
                                       min(p.B, (r + 1) * sp.sep_rows_per), p.fd, p.fd_fixed);
            return;
        }
        // [64][L]: the output run of the group.  It takes the place of the staged row: only the first wave reads that,
        // and has read it by the time it writes the tile (one wave's LDS accesses complete in order)
        double* tile = lds;
        __shared__ int s_nv;
        __shared__ unsigned s_gpair[kWave];                          // the group's pairs, i | j << 16: one load per lane, looked up below
        if (threadIdx.x < kWave) {
            const int nv = tsep_group_to_tile<NC>(p.ts, xy_dyn, VPQ, g, tile);
            if (threadIdx.x == 0) s_nv = nv;
            const int2 ij = p.ts.pairs[min(g * kWave + (int)threadIdx.x, p.ts.n_pairs - 1)];
            s_gpair[threadIdx.x] = (unsigned)ij.x | ((unsigned)ij.y << 16);
        }
        __syncthreads();
        const int n_el = s_nv * L;                                   // doubles in the run
        constexpr int kSlots = (kWave * L / 2 + 255) / 256;          // 16-byte pieces per thread
        double v0[kSlots], v1[kSlots];
        int pr0[kSlots], pr1[kSlots];                                // (i | j << 16) of the pair each element belongs to, -1: none
#pragma unroll
        for (int s = 0; s < kSlots; ++s) {
            const int m = (int)threadIdx.x + 256 * s, e0 = 2 * m, e1 = e0 + 1;
            v0[s] = v1[s] = 0.0; pr0[s] = pr1[s] = -1;
            if (e0 < n_el) { pr0[s] = (int)s_gpair[e0 / L]; v0[s] = tile[e0]; }
            if (e1 < n_el) { pr1[s] = (int)s_gpair[e1 / L]; v1[s] = tile[e1]; }
        }
        const int b0 = r * sp.sep_rows_per, b1 = min(p.B, b0 + sp.sep_rows_per);
        for (int b = b0; b < b1; ++b) {
            const int fd_e = fd_element(p.fd, p.fd_fixed, NC, b);
            const int vb = fd_e >= 0 ? fd_e / (2 * NC) : -1;
            const size_t ob = ((size_t)b * p.ts.n_pairs + (size_t)g * kWave) * L;
            double* o = p.ts.out + ob;
            const bool aligned = (ob & 1) == 0;                     // rows of an odd length start on odd elements
#pragma unroll
            for (int s = 0; s < kSlots; ++s) {
                const int m = (int)threadIdx.x + 256 * s;
                const bool w0 = pr0[s] >= 0 && (pr0[s] & 0xffff) != vb && (pr0[s] >> 16) != vb;
                const bool w1 = pr1[s] >= 0 && (pr1[s] & 0xffff) != vb && (pr1[s] >> 16) != vb;
                if (w0 && w1 && aligned) store_nt2(o + 2 * m, v0[s], v1[s]);
                else if (w0 && w1) { store_nt(o + 2 * m, v0[s]); store_nt(o + 2 * m + 1, v1[s]); }
                else if (w0) store_nt(o + 2 * m, v0[s]);
                else if (w1) store_nt(o + 2 * m + 1, v1[s]);
            }
        }
    }
}

// The pair sweep of LARGE rows (hulls of a row beyond 48 KB of LDS, C4) as one launch: the tiled sweep whose chunks
// also write the temporal-separation rows of their TA x 64 tile of the pair matrix (tsep_tile_from_xy) -- a chunk has
// exactly that tile's vehicles staged.  Needs a hull pair list that holds every vehicle pair (build_tiles checks).
template <int NC>
__global__ __launch_bounds__(256, 4) void k_pair_sweep_tiled(const GjkSwarmParams p)
{
    extern __shared__ double2 xy_dyn[];
    if (p.dyn.out != nullptr && (int)blockIdx.x >= p.dyn_first_block) {
        // the speed / angular-rate groups of the batch as the grid's last workgroups, as in k_pair_sweep: the whole
        // evaluation of a large swarm (C4) is this one launch
        // (16 control points: the two-wave form spills 800 bytes per lane under this kernel's 128 VGPRs, the four-wave form
        // half of that: C4 25.9 against 26.3 ms)
        if (NC <= 11 && threadIdx.x >= 2 * kWave) return;
        dynamics2_group<NC, true, (NC > 11)>(p.dyn, reinterpret_cast<double*>(xy_dyn), (int)blockIdx.x - p.dyn_first_block);
        return;
    }
    gjk_planar_body<NC, 2, true>(p, xy_dyn, -1, -1);
}

// -------------------------------------------------------------------------------------
//  3-D swarm sweep, fixed point count (the SwarmOfAerialVehicles shapes: every object has <= NC
//  points, any z).  Same organisation as the planar sweep -- objects staged point-major, unrolled
//  support scan, lane refill in trip-count order of the previous sweep, XCD-aware ids, phase 1
//  leaves records, phase 2 evaluates the closest points convergently -- on the full 3-D state machine
//  of gjk_device.h (simplex_update, the Brent cycle detector, closest_from_simplex).
//  LDS point = (x, y, z, 0): 32 bytes, object pitch 4 NC + 2 doubles (odd in 16-byte units).
// -------------------------------------------------------------------------------------
template <int NC>
__device__ __forceinline__ void support_fixed3(const double* __restrict__ o1, const double* __restrict__ o2, const V3& d,
                                               gjk::Vert& out)
{
    const double ndx = -d.x, ndy = -d.y, ndz = -d.z;
    int i1 = 0, i2 = 0;
    double m1 = o1[0] * d.x + o1[1] * d.y + o1[2] * d.z;
    double m2 = o2[0] * ndx + o2[1] * ndy + o2[2] * ndz;
#pragma unroll
    for (int i = 1; i < NC; ++i) {
        const double2 a = *reinterpret_cast<const double2*>(o1 + 4 * i), bq = *reinterpret_cast<const double2*>(o2 + 4 * i);
        const double c1 = a.x * d.x + a.y * d.y + o1[4 * i + 2] * d.z;
        const double c2 = bq.x * ndx + bq.y * ndy + o2[4 * i + 2] * ndz;
        i1 = c1 > m1 ? i : i1;
        i2 = c2 > m2 ? i : i2;
        m1 = __builtin_fmax(m1, c1);
        m2 = __builtin_fmax(m2, c2);
    }
    out.i1 = i1; out.i2 = i2;
    out.v = V3{ o1[4 * i1] - o2[4 * i2], o1[4 * i1 + 1] - o2[4 * i2 + 1], o1[4 * i1 + 2] - o2[4 * i2 + 2] };
}

__host__ __device__ constexpr size_t sweep3d_lds_bytes(int n_obj, int nc, int chunk)
{
    // objects | r0 r1 r2 int[chunk] each | plist, pnat int[chunk] | ext int[2 n_obj] | ord u16[chunk]
    return 8 * (size_t)n_obj * (4 * nc + 2) + 20 * (size_t)chunk + 8 * (size_t)n_obj + 2 * (size_t)chunk + 16;
}

template <int NC>
__device__ __forceinline__ void gjk_swarm3d_body(const GjkSwarmParams& p, double* lds)
{
    using gjk::Simplex;
    using gjk::Vert;
    constexpr int PITCH = 4 * NC + 2;
    __shared__ int s_next;
    __shared__ int s_hist[256];
    const int per = 8 * p.wgs_per_row;
    const int grp = (int)blockIdx.x / per, g8 = (int)blockIdx.x - grp * per;
    const int b = grp * 8 + (g8 & 7), w = g8 >> 3;
    if (b >= p.B) return;
    const int n_obj = p.n_veh + p.n_poly;
    const int c1 = max(0, min(p.n_pairs - w * p.chunk, p.chunk));
    int* r0 = reinterpret_cast<int*>(lds + n_obj * PITCH);
    int* r1 = r0 + p.chunk;
    int* r2 = r1 + p.chunk;
    unsigned* plist = reinterpret_cast<unsigned*>(r2 + p.chunk);
    unsigned* pnat = plist + p.chunk;
    int* ext = reinterpret_cast<int*>(pnat + p.chunk);
    unsigned short* ord = reinterpret_cast<unsigned short*>(ext + 2 * n_obj);

    // ---- stage: vehicles (rows x, y[, z] of the evaluation row) and padded polygons, point-major
    const int vlen = p.dim * NC;
    const double* Yrow = p.fd ? p.Y : p.Y + (size_t)b * p.n_veh * vlen;
    const int fd_e = fd_element(p.fd, p.fd_fixed, NC, b);
    for (int e = threadIdx.x; e < n_obj * 4 * NC; e += blockDim.x) {
        const int o = e / (4 * NC), r = e - o * (4 * NC), k = r >> 2, q = r & 3;
        double val = 0.0;
        if (o < p.n_veh) {
            if (q < p.dim) { const int ee = o * vlen + q * NC + k; val = Yrow[ee]; if (ee == fd_e) val += p.fd_h; }
        } else if (q < 3) {
            const int off = p.poly_off[o - p.n_veh], K = p.poly_off[o - p.n_veh + 1] - off;
            val = p.poly[3 * off + q * K + (k < K ? k : 0)];       // padded with copies of vertex 0
        }
        lds[o * PITCH + r] = val;
    }
    __syncthreads();
    // first two doSimplex steps: supports along (1,0,0) and (-1,-0,-0) depend on one object only
    for (int o = threadIdx.x; o < n_obj; o += blockDim.x) {
        const double* q = lds + o * PITCH;
        int imx = 0, imn = 0;
        double mx = q[0] * 1.0 + q[1] * 0.0 + q[2] * 0.0, mn = q[0] * -1.0 + q[1] * -0.0 + q[2] * -0.0;
#pragma unroll
        for (int i = 1; i < NC; ++i) {
            const double v1 = q[4 * i] * 1.0 + q[4 * i + 1] * 0.0 + q[4 * i + 2] * 0.0;
            const double v2 = q[4 * i] * -1.0 + q[4 * i + 1] * -0.0 + q[4 * i + 2] * -0.0;
            if (v1 > mx) { mx = v1; imx = i; }
            if (v2 > mn) { mn = v2; imn = i; }
        }
        ext[2 * o] = imx; ext[2 * o + 1] = imn;
    }
    // pair slots, in trip-count order of the previous sweep
    {
        auto slots_of = [&](int l) { const int kq = w * p.chunk + l; return (unsigned)p.pa[kq] | ((unsigned)p.pb[kq] << 16); };
        if (p.len_in) {
            const unsigned char* len = p.len_in + (size_t)b * p.len_in_stride + (size_t)w * p.chunk;
            s_hist[threadIdx.x] = 0;
            __syncthreads();
            for (int l = threadIdx.x; l < c1; l += blockDim.x) atomicAdd(&s_hist[255 - len[l]], 1);
            __syncthreads();
            if (threadIdx.x < kWave) {
                const int lane = threadIdx.x;
                const int v0 = s_hist[4 * lane], v1 = s_hist[4 * lane + 1], v2 = s_hist[4 * lane + 2], v3 = s_hist[4 * lane + 3];
                const int sum = v0 + v1 + v2 + v3;
                int incl = sum;
#pragma unroll
                for (int dd = 1; dd < kWave; dd <<= 1) {
                    const int t = __shfl_up(incl, dd);
                    if (lane >= dd) incl += t;
                }
                const int excl = incl - sum;
                s_hist[4 * lane] = excl; s_hist[4 * lane + 1] = excl + v0;
                s_hist[4 * lane + 2] = excl + v0 + v1; s_hist[4 * lane + 3] = excl + v0 + v1 + v2;
            }
            __syncthreads();
            for (int l = threadIdx.x; l < c1; l += blockDim.x) {
                const int pos = atomicAdd(&s_hist[255 - len[l]], 1);
                const unsigned ab = slots_of(l);
                ord[pos] = (unsigned short)l; plist[pos] = ab; pnat[l] = ab;
            }
        } else {
            for (int l = threadIdx.x; l < c1; l += blockDim.x) { ord[l] = (unsigned short)l; plist[l] = pnat[l] = slots_of(l); }
        }
    }
    if (threadIdx.x == 0) s_next = 0;
    __syncthreads();
    const bool shortcut = p.max_iter >= 3 && p.md_cap >= 2;

    // ---- phase 1
    {
        const int lane = threadIdx.x & (kWave - 1);
        int k = -1, slot = 0, sa = 0, sb = 0;
        bool exhausted = false;
        const double* o1 = lds;
        const double* o2 = lds;
        Simplex s, old;
        gjk::Checkpoint chk;
        V3 dir{ 1.0, 0.0, 0.0 };
        int phase = 0, it = 0, rr = 0, nsup = 0;
        s.keys = 0;
        s.A = Vert{ V3{ 0, 0, 0 }, 0, 0 };
        s.B = s.A; s.C = s.A; s.D = s.A; old = s;
        chk.start(s, dir);
        for (;;) {
            const unsigned long long want = __ballot(k < 0 && !exhausted);
            if (want) {
                const int leader = __ffsll((long long)want) - 1;
                int base = 0;
                if (lane == leader) base = atomicAdd(&s_next, __popcll(want));
                base = __shfl(base, leader);
                if (k < 0 && !exhausted) {
                    const int my = base + __popcll(want & ((1ull << lane) - 1ull));
                    if (my < c1) {
                        k = my;
                        slot = (int)ord[k];
                        const unsigned ab = plist[k];
                        sa = (int)(ab & 0xffffu); sb = (int)(ab >> 16);
                        o1 = lds + sa * PITCH; o2 = lds + sb * PITCH;
                        s.keys = 0; dir = V3{ 1.0, 0.0, 0.0 };
                        phase = 0; it = 0; rr = 0; nsup = 0;
                        if (shortcut) {
                            const int ax = ext[2 * sa], an = ext[2 * sa + 1], bx = ext[2 * sb], bn = ext[2 * sb + 1];
                            const Vert A1{ V3{ o1[4 * ax] - o2[4 * bn], o1[4 * ax + 1] - o2[4 * bn + 1], o1[4 * ax + 2] - o2[4 * bn + 2] }, ax, bn };
                            const Vert A2{ V3{ o1[4 * an] - o2[4 * bx], o1[4 * an + 1] - o2[4 * bx + 1], o1[4 * an + 2] - o2[4 * bx + 2] }, an, bx };
                            const bool md1 = gjk::dotb(A1.v, dir) < 0;
                            dir = gjk::neg(dir);
                            s.B = A1; s.A = A2; s.keys = gjk::kA | gjk::kB;
                            nsup = 2;
                            if (!md1) {
                                it = 2;
                                if (gjk::dotb(A2.v, dir) < 0) { phase = 1; rr = 0; chk.start(s, dir); }
                            } else {
                                // iteration 2 ran inside minimumDistance with old = {A: A1}
                                phase = 1; rr = 1;
                                const bool m = gjk::eq(A2.v, A1.v) ||
                                    (A1.v.x == 0.0 && A1.v.y == 0.0 && A1.v.z == 0.0 &&
                                     A2.v.x == o1[4 * ax] && A2.v.y == o1[4 * ax + 1] && A2.v.z == o1[4 * ax + 2] &&
                                     A2.v.x == o2[4 * bn] && A2.v.y == o2[4 * bn + 1] && A2.v.z == o2[4 * bn + 2]);
                                if (m) {
                                    r0[slot] = (1 + 1) | (OBTG_ST_OK << 2) | (gjk::kA << 4) | (nsup << 10);
                                    r1[slot] = A1.i1 | (A1.i2 << 8);
                                    r2[slot] = 0;
                                    k = -1;
                                } else {
                                    // the checkpoint of minimumDistance is the state at its entry: {A: A1}, dir (1,0,0)
                                    Simplex e1; e1.keys = gjk::kA; e1.A = A1; e1.B = A1; e1.C = A1; e1.D = A1;
                                    chk.start(e1, V3{ 1.0, 0.0, 0.0 });
                                    if (chk.step(s, dir)) {   // cannot repeat after one round; keeps the counters aligned
                                        r0[slot] = (1 + 1) | (OBTG_ST_CYCLE << 2) | (nsup << 10); r1[slot] = 0; r2[slot] = 0; k = -1;
                                    }
                                }
                            }
                        }
                    } else exhausted = true;
                }
            }
            if (__ballot(k >= 0) == 0ull) break;
            if (k >= 0) {
                if (phase == 1) old = s;
                int flag = -2, status = OBTG_ST_OK;
                if (gjk::simplex_update(s, dir)) {
                    support_fixed3<NC>(o1, o2, dir, s.A);
                    s.keys |= gjk::kA;
                    ++nsup;
                }
                if (phase == 0) {
                    ++it;
                    if (s.keys & gjk::kColl) flag = 0;
                    else if (gjk::dotb(s.A.v, dir) < 0) { phase = 1; rr = 0; chk.start(s, dir); }
                    else if (it >= p.max_iter) { flag = -1; status = OBTG_ST_MAXITER; }
                } else {
                    ++rr;
                    // matches_old through per-object bases: the stored indices address o1 / o2 directly
                    auto pts_eq = [&](const Vert& ov, bool has, bool haspts) {
                        if (has && gjk::eq(s.A.v, ov.v)) return true;
                        return haspts && s.A.v.x == o1[4 * ov.i1] && s.A.v.y == o1[4 * ov.i1 + 1] && s.A.v.z == o1[4 * ov.i1 + 2] &&
                               s.A.v.x == o2[4 * ov.i2] && s.A.v.y == o2[4 * ov.i2 + 1] && s.A.v.z == o2[4 * ov.i2 + 2];
                    };
                    const bool m = pts_eq(old.A, old.keys & gjk::kA, old.keys & gjk::kA) ||
                                   pts_eq(old.B, old.keys & gjk::kB, old.keys & gjk::kB) ||
                                   pts_eq(old.C, old.keys & gjk::kC, old.keys & gjk::kC) ||
                                   pts_eq(old.D, old.keys & gjk::kD, old.keys & gjk::kDpts) ||
                                   ((old.keys & gjk::kColl) && s.A.v.x == 1.0 && s.A.v.y == 1.0 && s.A.v.z == 1.0);
                    if (m) flag = 1;
                    else if (chk.step(s, dir)) { flag = 1; status = OBTG_ST_CYCLE; }
                    else if (rr >= p.md_cap) { flag = 1; status = OBTG_ST_MD_CAP; }
                }
                if (flag != -2) {
                    r0[slot] = (flag + 1) | (status << 2) | ((old.keys & 63) << 4) | (min(nsup, 0x3fffff) << 10);
                    r1[slot] = old.A.i1 | (old.A.i2 << 8) | (old.B.i1 << 16) | (old.B.i2 << 24);
                    r2[slot] = old.C.i1 | (old.C.i2 << 8) | (old.D.i1 << 16) | (old.D.i2 << 24);
                    k = -1;
                }
            }
        }
    }
    __syncthreads();

    // ---- phase 2: closest points / distance from the recorded simplices (gjk.py:299-360), one lane per pair
    const size_t obase = (size_t)b * p.n_pairs + (size_t)w * p.chunk;
    const double qnan = __builtin_nan("");
    for (int l = threadIdx.x; l < c1; l += blockDim.x) {
        const unsigned ab2 = pnat[l];
        const double* q1 = lds + (int)(ab2 & 0xffffu) * PITCH;
        const double* q2 = lds + (int)(ab2 >> 16) * PITCH;
        const int w0 = r0[l], w1 = r1[l], w2 = r2[l];
        const int flag = (w0 & 3) - 1, status = (w0 >> 2) & 3, keys = (w0 >> 4) & 63, n_scans = (int)((unsigned)w0 >> 10);
        Result r;
        r.c1 = V3{ qnan, qnan, qnan }; r.c2 = r.c1; r.dist = qnan;
        if (flag == 1 && status == OBTG_ST_OK) {
            // gjk_device.h addresses coordinate c of point k of a set at `base` as mem(base + c*cs + k): with
            // cs = 1 << 24 and bit 23 of the base telling the two objects apart, both resolve into LDS
            struct MemTwo {
                const double* a; const double* b2;
                __device__ __forceinline__ double operator()(int idx) const
                {
                    const int kpt = idx & 0x7fffff, c = (idx >> 24) & 3;
                    return (idx & 0x800000) ? b2[4 * kpt + c] : a[4 * kpt + c];
                }
            };
            Ctx<MemTwo> g;
            g.mem = MemTwo{ q1, q2 };
            g.P1 = Poly{ 0, 1 << 24, NC, 1 };
            g.P2 = Poly{ 0x800000, 1 << 24, NC, 1 };
            g.trace = nullptr; g.trace_cap = 0; g.n_support = 0;
            auto vert = [&](int i1, int i2) {
                return Vert{ V3{ q1[4 * i1] - q2[4 * i2], q1[4 * i1 + 1] - q2[4 * i2 + 1], q1[4 * i1 + 2] - q2[4 * i2 + 2] }, i1, i2 };
            };
            Simplex s;
            s.keys = keys;
            s.A = vert(w1 & 0xff, (w1 >> 8) & 0xff);
            s.B = vert((w1 >> 16) & 0xff, (w1 >> 24) & 0xff);
            s.C = vert(w2 & 0xff, (w2 >> 8) & 0xff);
            s.D = vert((w2 >> 16) & 0xff, (w2 >> 24) & 0xff);
            gjk::closest_from_simplex(g, s, r);
        }
        const size_t o = obase + l;
        p.flag[o] = flag;
        p.p1[3 * o] = r.c1.x; p.p1[3 * o + 1] = r.c1.y; p.p1[3 * o + 2] = r.c1.z;
        p.p2[3 * o] = r.c2.x; p.p2[3 * o + 1] = r.c2.y; p.p2[3 * o + 2] = r.c2.z;
        p.dist[o] = r.dist;
        if (p.nsup) p.nsup[o] = n_scans;
        if (p.status) p.status[o] = status;
        if (p.len_out) p.len_out[o] = (unsigned char)min(n_scans, 255);
    }
}

template <int NC>
__global__ __launch_bounds__(256) void k_gjk_swarm_3d(const GjkSwarmParams p)
{
    extern __shared__ double2 xyz_dyn[];
    gjk_swarm3d_body<NC>(p, reinterpret_cast<double*>(xyz_dyn));
}

// The 3-D sweep as the ONE launch of an evaluation batch (the SwarmOfAerialVehicles shapes: a step is otherwise three
// launch-latency-sized kernels).  Workgroup (row b, part w) first runs its part of the row's temporal-separation block
// and of its speed rows -- normsq_elev_body, the stand-alone kernels' code on its own staging area behind the sweep's
// LDS, hence the same bits -- and then the gjkNew sweep; the Bernstein stores drain meanwhile.
template <int NC>
__global__ __launch_bounds__(256) void k_pair_sweep_3d(const GjkSwarmParams p, const NsParams ts, const NsParams sp,
                                                       const int ns_offset /* doubles */)
{
    extern __shared__ double2 xyz_dyn[];
    double* lds = reinterpret_cast<double*>(xyz_dyn);
    const int per = 8 * p.wgs_per_row;
    const int grp = (int)blockIdx.x / per, g8 = (int)blockIdx.x - grp * per;
    const int b = grp * 8 + (g8 & 7), w = g8 >> 3;
    if (b >= p.B) return;
    if (ts.out != nullptr) normsq_elev_body<NC, 3, 0, false, false>(ts, b, w, lds + ns_offset);
    if (sp.out != nullptr) {
        __syncthreads();
        normsq_elev_body<NC, 3, 1, false, false>(sp, b, w, lds + ns_offset);
    }
    gjk_swarm3d_body<NC>(p, lds);
}

// which vehicles of row b differ (bitwise) from row 0: chg[b][v]
__global__ void k_changed_objects(const double* __restrict__ Y, int B, int n_veh, int vlen, unsigned char* __restrict__ chg)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * n_veh) return;
    const int b = t / n_veh, v = t - b * n_veh;
    const unsigned long long* r0 = reinterpret_cast<const unsigned long long*>(Y) + (size_t)v * vlen;
    const unsigned long long* rb = r0 + (size_t)b * n_veh * vlen;
    bool diff = false;
    for (int i = 0; i < vlen; ++i) diff |= r0[i] != rb[i];
    chg[t] = diff ? 1 : 0;
}

// rows 1..B-1 of a [B][row_len] array := row 0 (streaming; blockIdx.y picks a block of 16 rows)
template <class T>
__global__ void k_bcast_row0(T* __restrict__ a, size_t row_len, int B)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= row_len) return;
    const T v = a[e];
    const int b0 = 1 + (int)blockIdx.y * 16, b1 = min(B, b0 + 16);
    for (int b = b0; b < b1; ++b) __builtin_nontemporal_store(v, a + (size_t)b * row_len + e);
}

// -------------------------------------------------------------------------------------
//  _minDist / _minDist2Poly: depth-first branch & bound with an explicit per-lane stack
// -------------------------------------------------------------------------------------
constexpr int kMdMaxK = 32;   // control points per curve supported by the branch & bound kernels

// numpy add.reduce on a contiguous float64 vector: 0 + pairwise_sum (8 partial sums for n >= 8)
__device__ __forceinline__ double np_sum(const double* a, int n)
{
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
        r0 += a[i]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
        r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; ++i) res += a[i];
    return res;
}

// bezier.py:1320-1351: parameter of a hull closest point on curve c[3][K]
__device__ double hull_param(const double* c, int K, const V3& cl)
{
    for (int i = 0; i < K; ++i)
        if (c[i] == cl.x && c[K + i] == cl.y && c[2 * K + i] == cl.z) return (double)i / (double)(K - 1);
    double e[kMdMaxK], q[kMdMaxK], W[kMdMaxK];
    for (int i = 0; i < K; ++i) {
        const double dx = cl.x - c[i], dy = cl.y - c[K + i], dz = cl.z - c[2 * K + i];
        double s = 0.0;
        s += dx * dx; s += dy * dy; s += dz * dz;
        e[i] = __builtin_sqrt(s);
    }
    for (int i = 0; i < K; ++i) {
        for (int j = 0; j < i; ++j) q[j] = e[i] / e[j];
        const double s1 = np_sum(q, i);
        for (int j = i + 1; j < K; ++j) q[j - i - 1] = e[i] / e[j];
        const double s2 = np_sum(q, K - i - 1);
        W[i] = 1 / (1 + s1 + s2);
    }
    for (int i = 0; i < K; ++i) q[i] = W[i] * (double)i / (double)K;
    return np_sum(q, K);
}

__device__ __forceinline__ double norm_seq(double ax, double ay, double az, double bx, double by, double bz)
{
    const double dx = ax - bx, dy = ay - by, dz = az - bz;
    double s = 0.0;
    s += dx * dx; s += dy * dy; s += dz * dz;
    return __builtin_sqrt(s);
}

// bezier.py:985-1027 deCasteljauSplit on one coordinate row; half = 0: left piece, 1: right piece
// (the reference reverses the "right" list, bezier.py:563)
__device__ void split_row(const double* src, int K, double t, int half, double* dst)
{
    double tmp[kMdMaxK];
    for (int i = 0; i < K; ++i) tmp[i] = src[i];
    int idx = 0;
    for (int sz = K; sz > 1; --sz) {
        if (half == 0) dst[idx] = tmp[0]; else dst[K - 1 - idx] = tmp[sz - 1];
        idx++;
        for (int i = 0; i < sz - 1; ++i) tmp[i] = (1 - t) * tmp[i] + t * tmp[i + 1];
    }
    if (half == 0) dst[K - 1] = tmp[0]; else dst[0] = tmp[0];
}

// frame layout (doubles): c1[3K] c2[3K] then scalars
enum { F_T1 = 0, F_T2, F_T1L, F_T1H, F_T2L, F_T2H, F_ALPHA, F_RT1, F_RT2, F_STATE, F_NSCAL };

struct MdParams {
    const double* __restrict__ curves;   // [n_curves][3][K]
    const int* __restrict__ pa;
    const int* __restrict__ pb;
    int n_pairs, K, max_iter, md_cap, max_depth, max_nodes;
    double eps;
    double* stack;                        // [n_pairs][max_depth][frame] (the wave form: one per WORKER wave)
    double* __restrict__ res;             // [n_pairs][3]
    int* __restrict__ info;               // [n_pairs][4]
    const int* __restrict__ order;        // wave form: the pairs in the order they are handed out (nullptr = list order)
    int* queue;                           // wave form: the next slot of `order` (zeroed before the launch)
};

__global__ __launch_bounds__(64) void k_min_dist(const MdParams p)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= p.n_pairs) return;
    const int K = p.K, FR = 6 * K + F_NSCAL;
    double* st = p.stack + (size_t)k * p.max_depth * FR;
    const double* ca = p.curves + (size_t)p.pa[k] * 3 * K;
    const double* cb = p.curves + (size_t)p.pb[k] * 3 * K;
    for (int i = 0; i < 3 * K; ++i) { st[i] = ca[i]; st[3 * K + i] = cb[i]; }
    {
        double* sc = st + 6 * K;
        sc[F_T1L] = 0; sc[F_T1H] = 1; sc[F_T2L] = 0; sc[F_T2H] = 1;
        sc[F_ALPHA] = INFINITY; sc[F_STATE] = 0;
    }
    int depth = 0;           // index of the current frame; cnt = depth + 1
    int nodes = 0, calls = 0, dmax = 0, status = OBTG_MD_OK;
    double r0 = INFINITY, r1 = -1, r2 = -1;   // value returned by the frame that just finished
    bool returning = false;
    for (;;) {
        double* f = st + (size_t)depth * FR;
        double* sc = f + 6 * K;
        int state = (int)sc[F_STATE];
        if (!returning && state == 0) {
            // ---- evaluate this node (bezier.py:1310-1374)
            if (depth + 1 > 1000) { r0 = r1 = r2 = -1; returning = true; depth--; if (depth < 0) break; continue; }
            if (nodes >= p.max_nodes) { status = OBTG_MD_NODE_CAP; break; }
            nodes++;
            if (depth + 1 > dmax) dmax = depth + 1;
            Ctx<MemGlobal> g;
            g.mem = MemGlobal{ f };
            g.P1 = Poly{ 0, K, K, 1 };
            g.P2 = Poly{ 3 * K, K, K, 1 };
            g.trace = nullptr; g.trace_cap = 0; g.n_support = 0;
            Result gr;
            gjk::run(g, p.max_iter, p.md_cap, gr);
            calls++;
            if (gr.status == OBTG_ST_MD_CAP || gr.status == OBTG_ST_CYCLE) { status = OBTG_MD_GJK_CAP; break; }
            double lb, t1, t2;
            if (gr.flag > 0) {
                lb = gr.dist;
                t1 = hull_param(f, K, gr.c1);
                t2 = hull_param(f + 3 * K, K, gr.c2);
            } else { t1 = 0.5; t2 = 0.5; lb = p.eps; }
            // _upperbound (bezier.py:1499-1516)
            const double* c1 = f; const double* c2 = f + 3 * K;
            double dd[4];
            dd[0] = norm_seq(c1[0], c1[K], c1[2 * K], c2[0], c2[K], c2[2 * K]);
            dd[1] = norm_seq(c1[0], c1[K], c1[2 * K], c2[K - 1], c2[2 * K - 1], c2[3 * K - 1]);
            dd[2] = norm_seq(c1[K - 1], c1[2 * K - 1], c1[3 * K - 1], c2[0], c2[K], c2[2 * K]);
            dd[3] = norm_seq(c1[K - 1], c1[2 * K - 1], c1[3 * K - 1], c2[K - 1], c2[2 * K - 1], c2[3 * K - 1]);
            int am = 0;
            for (int i = 1; i < 4; ++i) if (dd[i] < dd[am]) am = i;
            for (int i = 0; i < 4; ++i) if (dd[i] != dd[i]) { am = i; break; }
            const double ub = dd[am], t1loc = (am >> 1) ? 1.0 : 0.0, t2loc = (am & 1) ? 1.0 : 0.0;
            double alpha = sc[F_ALPHA], nT1, nT2;
            if (ub <= alpha) {
                alpha = ub;
                nT1 = (1 - t1loc) * sc[F_T1L] + t1loc * sc[F_T1H];
                nT2 = (1 - t2loc) * sc[F_T2L] + t2loc * sc[F_T2H];
            } else { nT1 = -1; nT2 = -1; }
            if (lb >= alpha * (1 - p.eps)) {
                r0 = alpha; r1 = nT1; r2 = nT2; returning = true; depth--;
                if (depth < 0) break;
                continue;
            }
            if (depth + 1 >= p.max_depth) { status = OBTG_MD_DEPTH_CAP; r0 = alpha; r1 = nT1; r2 = nT2; break; }
            if (t1 != t1) t1 = 0;    // Bezier.split: NaN -> 0 (bezier.py:555-557)
            if (t2 != t2) t2 = 0;
            sc[F_T1] = t1; sc[F_T2] = t2; sc[F_ALPHA] = alpha; sc[F_RT1] = nT1; sc[F_RT2] = nT2;
            state = 1; sc[F_STATE] = 1;
        }
        if (returning) {
            // a child of this frame finished: keep the better answer (bezier.py:1384-1406)
            if (r0 < sc[F_ALPHA]) { sc[F_ALPHA] = r0; sc[F_RT1] = r1; sc[F_RT2] = r2; }
            returning = false;
            state = (int)sc[F_STATE];
        }
        if (state >= 5) {
            r0 = sc[F_ALPHA]; r1 = sc[F_RT1]; r2 = sc[F_RT2]; returning = true; depth--;
            if (depth < 0) break;
            continue;
        }
        // ---- descend into child state-1: (c3,c5) (c3,c6) (c4,c5) (c4,c6)
        {
            const int ch = state - 1, h1 = ch >> 1, h2 = ch & 1;
            const double t1 = sc[F_T1], t2 = sc[F_T2];
            double* nf = f + FR;
            for (int c = 0; c < 3; ++c) {
                split_row(f + c * K, K, t1, h1, nf + c * K);
                split_row(f + 3 * K + c * K, K, t2, h2, nf + 3 * K + c * K);
            }
            double* ns = nf + 6 * K;
            const double t1len = sc[F_T1H] - sc[F_T1L], t2len = sc[F_T2H] - sc[F_T2L];
            const double m1 = sc[F_T1L] + t1 * t1len, m2 = sc[F_T2L] + t2 * t2len;
            ns[F_T1L] = h1 ? m1 : sc[F_T1L]; ns[F_T1H] = h1 ? sc[F_T1H] : m1;
            ns[F_T2L] = h2 ? m2 : sc[F_T2L]; ns[F_T2H] = h2 ? sc[F_T2H] : m2;
            ns[F_ALPHA] = sc[F_ALPHA]; ns[F_STATE] = 0;
            sc[F_STATE] = state + 1;
            depth++;
        }
    }
    p.res[3 * k] = r0; p.res[3 * k + 1] = r1; p.res[3 * k + 2] = r2;
    if (p.info) { p.info[4 * k] = nodes; p.info[4 * k + 1] = calls; p.info[4 * k + 2] = dmax; p.info[4 * k + 3] = status; }
}

// -------------------------------------------------------------------------------------
//  _minDist with one pair per WAVEFRONT.  The recursion of bezier.py:1283-1408 is sequential per pair
//  (every child is pruned against the alpha its elder siblings returned), so the parallelism inside a
//  pair is inside a node: the support scans of gjkNew (support_pts_wave), the two closest-point
//  parameters (one curve per half-wave, one hull point per lane) and the six de Casteljau rows of a
//  child.  Everything else runs redundantly in all lanes with wave-uniform control flow, on the same
//  device functions as k_min_dist, so results, node and GJK-call counts are those of the one-lane form.
//  LDS per pair: current node's curves, the next child's curves, the scalar part of every stack frame;
//  the curve part of the frames stays in the global stack (written once per child, re-read when the
//  walk comes back to a node to split its next child).
// -------------------------------------------------------------------------------------
// np_sum over q(0..n-1) without materialising the vector (same association as np_sum above)
template <class F>
__device__ __forceinline__ double np_sum_f(int n, F q)
{
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; ++i) r += q(i);
        return r;
    }
    double r0 = q(0), r1 = q(1), r2 = q(2), r3 = q(3), r4 = q(4), r5 = q(5), r6 = q(6), r7 = q(7);
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
        r0 += q(i); r1 += q(i + 1); r2 += q(i + 2); r3 += q(i + 3);
        r4 += q(i + 4); r5 += q(i + 5); r6 += q(i + 6); r7 += q(i + 7);
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; ++i) res += q(i);
    return res;
}

// split_row on LDS rows: `work` (K doubles) is the lane's scratch row, so nothing goes to private memory
__device__ __forceinline__ void split_row_lds(const double* src, int K, double t, int half, double* dst, double* work)
{
    for (int i = 0; i < K; ++i) work[i] = src[i];
    int idx = 0;
    for (int sz = K; sz > 1; --sz) {
        if (half == 0) dst[idx] = work[0]; else dst[K - 1 - idx] = work[sz - 1];
        idx++;
        for (int i = 0; i < sz - 1; ++i) work[i] = (1 - t) * work[i] + t * work[i + 1];
    }
    if (half == 0) dst[K - 1] = work[0]; else dst[0] = work[0];
}

// The lane above's value (lane 63: zero): one DPP move per half of the double, no LDS round trip.
__device__ __forceinline__ double wave_next_lane(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// deCasteljauSplit (bezier.py:985-1027) of three coordinate rows at once, LEVEL-parallel: lane (r, i) = r * K + i holds
// control point i of row r; a level is one neighbour exchange and the reference's `(1 - t) * w[i] + t * w[i + 1]` in every
// lane, K - 1 levels in all -- the same operation per element as split_row_lds, so the same bits, but a chain of K - 1
// dependent steps instead of K (K - 1) / 2 LDS round trips in ONE lane (round 5: the splits were a fifth of a node's
// latency, and a pair's search is a serial chain of up to max_nodes nodes).  Left piece: point L is lane 0's value at
// level L; right piece (stored reversed, bezier.py:563): point i is lane i's value at the last level it takes part in.
// Needs 3 K <= 64; rl / il = lane / K, lane % K (rl >= 3: the lane idles).
// One level: every lane stores -- the lane whose value is a point of the piece at this level to its place, the others to a
// slot of their own in `dump` (64 doubles of scratch) -- so that a level has no divergent region: a select on the address,
// one ds_write, two DPP moves, two multiplies and an add.
template <int KC>          // KC > 0: K is that constant and the levels are unrolled
__device__ __forceinline__ void split_rows3_wave_t(const double* src, int K, double t, int half, double* dst, int rl, int il,
                                                   double* dump)
{
    if (KC > 0) K = KC;
    const bool valid = rl < 3;
    double w = valid ? src[rl * K + il] : 0.0;
    const double u = 1 - t;
    // who records at level L: the left piece's point L is lane 0's value, the right piece's point i is lane i's value at
    // level K - 1 - i; the last level's value (lane 0) is point K - 1 of the left piece / point 0 of the right one
    const bool every = valid && half == 0 && il == 0;
    const int my_level = (valid && half != 0) ? K - 1 - il : -1;
    double* out = dst + rl * K + (half == 0 ? 0 : il);
    double* mine = dump + (threadIdx.x & 63);
    auto level = [&](int L) {
        double* a = every ? out + L : (my_level == L ? out : mine);
        *a = w;
        const double up = wave_next_lane(w);
        w = u * w + t * up;
    };
    if constexpr (KC > 0) {
#pragma unroll
        for (int L = 0; L < KC - 1; ++L) level(L);
    } else {
        for (int L = 0; L < K - 1; ++L) level(L);
    }
    if (valid && il == 0) dst[rl * K + (half == 0 ? K - 1 : 0)] = w;
}

__device__ __forceinline__ void split_rows3_wave(const double* src, int K, double t, int half, double* dst, int rl, int il,
                                                 double* dump)
{
    switch (K) {        // (wave-uniform)
#define OBTG_CASE(NC_) case NC_: split_rows3_wave_t<NC_>(src, K, t, half, dst, rl, il, dump); return;
        OBTG_NC_DYN(OBTG_CASE)
#undef OBTG_CASE
        default: split_rows3_wave_t<0>(src, K, t, half, dst, rl, il, dump);
    }
}

__device__ __forceinline__ double hull_param_wave(const double* c, int K, const V3& cl, double* sh_e, double* sh_q, int li)
{
    // exact row match first (bezier.py:1320-1333): lowest matching index
    const bool hit = li < K && c[li] == cl.x && c[K + li] == cl.y && c[2 * K + li] == cl.z;
    const unsigned long long mall = __ballot(hit);
    const unsigned m = (threadIdx.x & 32) ? (unsigned)(mall >> 32) : (unsigned)mall;
    if (m) return (double)(__ffs((int)m) - 1) / (double)(K - 1);
    if (li < K) {
        const double dx = cl.x - c[li], dy = cl.y - c[K + li], dz = cl.z - c[2 * K + li];
        double s = 0.0;
        s += dx * dx; s += dy * dy; s += dz * dz;
        sh_e[li] = __builtin_sqrt(s);
    }
    wave_sync();
    if (li < K) {
        const double ei = sh_e[li];
        const double s1 = np_sum_f(li, [&](int j) { return ei / sh_e[j]; });
        const double s2 = np_sum_f(K - li - 1, [&](int j) { return ei / sh_e[li + 1 + j]; });
        const double W = 1 / (1 + s1 + s2);
        sh_q[li] = W * (double)li / (double)K;
    }
    wave_sync();
    return np_sum(sh_q, K);
}

// Round 5: the waves are WORKERS.  A pair's search costs anything from one gjkNew call to max_nodes nodes of four, and a
// launch of one single-wave workgroup per pair, all resident at once, lasts as long as the SIMD that happened to receive the
// most long pairs (C5-sized sweep, PMC: VALU 23 % busy over the launch, profiles/r05_mindist_pmc_before.txt).  Now a fixed
// number of waves per SIMD pull pairs from a queue -- in the order of `order`: the previous evaluation's node counts,
// descending (the host keeps them per pair list), so the long searches start first, spread over the chip, and the short ones
// fill in behind them.  Pairs are independent and every result is written per pair: which wave evaluates a pair, and when,
// changes nothing in res / info.
#ifndef OBTG_MD_MIN_WAVES
#define OBTG_MD_MIN_WAVES 2     // worker waves per SIMD the register allocation is held to (launch bound's second argument)
#endif
__global__ __launch_bounds__(64, OBTG_MD_MIN_WAVES) void k_min_dist_wave(const MdParams p)
{
    extern __shared__ double md_lds[];
    const int lane = threadIdx.x, half = lane >> 5, li = lane & 31;
    const int K = p.K, FR = 6 * K + F_NSCAL;
    const int rl = lane / K, il = lane - rl * K;          // (row, point) of the lane in the level-parallel splits
    double* st = p.stack + (size_t)blockIdx.x * p.max_depth * FR;
  for (;;) {
    // The pull must not hang on a lane-dependent branch.  The first form -- `if (lane == 0) slot = atomicAdd(queue, 1);
    // slot = __shfl(slot, 0);`, with `if (lane == 0) { write the results }` as the loop's last statement -- was compiled
    // into a loop in which the two lane-0 regions on either side of the back edge are ONE region and lanes 1..63 go round
    // without lane 0: their ds_bpermute reads an inactive lane (0), they start pair 0 again on their own, and the search,
    // which relies on lane 0's writes, never ends (ISA of that build: the depth-2 loop under BB2_3; the GPU suite hung in
    // its first minDist call).  So: every lane issues the atomic, lane 0 with the increment and the others with 0 -- correct
    // whether or not the compiler folds the 64 into one -- and the results below are stored by every lane (the values are
    // wave-uniform): no lane-dependent control flow at either end of the loop body.
    const int ticket = atomicAdd(p.queue, lane == 0 ? 1 : 0);
    const int slot = __builtin_amdgcn_readfirstlane(ticket);
    if (slot >= p.n_pairs) break;                // every worker ends here: the queue only grows
    const int k = p.order ? p.order[slot] : slot;
    __syncthreads();                             // (one wave per workgroup: the previous pair's LDS reads are done)
    double* cur = md_lds;                       // [6K] curves of the node being evaluated
    double* nxt = cur + 6 * K;                  // [6K] curves of the child being built
    double* sh_e = nxt + 6 * K;                 // [2][kMdMaxK]; with sh_q also the six scratch rows of a split
    double* sh_q = sh_e + 2 * kMdMaxK;          // [2][kMdMaxK] (+ 2 more rows so that 6 K doubles fit)
    double* scs = sh_q + 4 * kMdMaxK;           // [max_depth][F_NSCAL] frame scalars
    const double* ca = p.curves + (size_t)p.pa[k] * 3 * K;
    const double* cb = p.curves + (size_t)p.pb[k] * 3 * K;
    for (int i = lane; i < 3 * K; i += kWave) {
        const double a = ca[i], bq = cb[i];
        st[i] = a; st[3 * K + i] = bq; cur[i] = a; cur[3 * K + i] = bq;
    }
    if (lane == 0) {
        scs[F_T1L] = 0; scs[F_T1H] = 1; scs[F_T2L] = 0; scs[F_T2H] = 1;
        scs[F_ALPHA] = INFINITY; scs[F_STATE] = 0;
    }
    __syncthreads();
    int depth = 0, cur_depth = 0;     // cur_depth: which frame's curves `cur` holds
    int nodes = 0, calls = 0, dmax = 0, status = OBTG_MD_OK;
    double r0 = INFINITY, r1 = -1, r2 = -1;
    bool returning = false;
#ifdef OBTG_MD_TIMING      // (variant builds: where a node's clocks go -- info[] then carries phase totals in units of 1024 clocks)
    unsigned long long tm_gjk = 0, tm_eval = 0, tm_desc = 0, tm_fetch = 0, tm_t;
#define OBTG_TM(acc, t0) acc += __builtin_readcyclecounter() - (t0)
#else
#define OBTG_TM(acc, t0)
#endif
    for (;;) {
        double* f = st + (size_t)depth * FR;
        double* sc = scs + depth * F_NSCAL;
        int state = (int)sc[F_STATE];
        if (!returning && state == 0) {
            if (depth + 1 > 1000) { r0 = r1 = r2 = -1; returning = true; depth--; if (depth < 0) break; continue; }
            if (nodes >= p.max_nodes) { status = OBTG_MD_NODE_CAP; break; }
            nodes++;
            if (depth + 1 > dmax) dmax = depth + 1;
            // `cur` holds this frame: frame 0 from the prologue, every other one from the descend step
            Ctx<MemLds> g;
            g.mem = MemLds{ cur };
            g.P1 = Poly{ 0, K, K, 1 };
            g.P2 = Poly{ 3 * K, K, K, 1 };
            g.trace = nullptr; g.trace_cap = 0; g.n_support = 0;
            Result gr;
#ifdef OBTG_MD_TIMING
            tm_t = __builtin_readcyclecounter();
#endif
            gjk::run<MemLds, false, true>(g, p.max_iter, p.md_cap, gr);
            OBTG_TM(tm_gjk, tm_t);
#ifdef OBTG_MD_TIMING
            tm_t = __builtin_readcyclecounter();
#endif
            calls++;
            if (gr.status == OBTG_ST_MD_CAP || gr.status == OBTG_ST_CYCLE) { status = OBTG_MD_GJK_CAP; break; }
            double lb, t1, t2;
            if (gr.flag > 0) {
                lb = gr.dist;
                const double tp = hull_param_wave(cur + half * 3 * K, K, half ? gr.c2 : gr.c1, sh_e + half * kMdMaxK,
                                                  sh_q + half * kMdMaxK, li);
                t1 = __shfl(tp, 0); t2 = __shfl(tp, 32);
            } else { t1 = 0.5; t2 = 0.5; lb = p.eps; }
            const double* c1 = cur; const double* c2 = cur + 3 * K;
            double dd[4];
            dd[0] = norm_seq(c1[0], c1[K], c1[2 * K], c2[0], c2[K], c2[2 * K]);
            dd[1] = norm_seq(c1[0], c1[K], c1[2 * K], c2[K - 1], c2[2 * K - 1], c2[3 * K - 1]);
            dd[2] = norm_seq(c1[K - 1], c1[2 * K - 1], c1[3 * K - 1], c2[0], c2[K], c2[2 * K]);
            dd[3] = norm_seq(c1[K - 1], c1[2 * K - 1], c1[3 * K - 1], c2[K - 1], c2[2 * K - 1], c2[3 * K - 1]);
            int am = 0;
            for (int i = 1; i < 4; ++i) if (dd[i] < dd[am]) am = i;
            for (int i = 0; i < 4; ++i) if (dd[i] != dd[i]) { am = i; break; }
            const double ub = dd[am], t1loc = (am >> 1) ? 1.0 : 0.0, t2loc = (am & 1) ? 1.0 : 0.0;
            double alpha = sc[F_ALPHA], nT1, nT2;
            if (ub <= alpha) {
                alpha = ub;
                nT1 = (1 - t1loc) * sc[F_T1L] + t1loc * sc[F_T1H];
                nT2 = (1 - t2loc) * sc[F_T2L] + t2loc * sc[F_T2H];
            } else { nT1 = -1; nT2 = -1; }
            if (lb >= alpha * (1 - p.eps)) {
                r0 = alpha; r1 = nT1; r2 = nT2; returning = true; depth--;
                if (depth < 0) break;
                continue;
            }
            if (depth + 1 >= p.max_depth) { status = OBTG_MD_DEPTH_CAP; r0 = alpha; r1 = nT1; r2 = nT2; break; }
            if (t1 != t1) t1 = 0;
            if (t2 != t2) t2 = 0;
            wave_sync();
            if (lane == 0) {
                sc[F_T1] = t1; sc[F_T2] = t2; sc[F_ALPHA] = alpha; sc[F_RT1] = nT1; sc[F_RT2] = nT2; sc[F_STATE] = 1;
            }
            wave_sync();
            state = 1;
            OBTG_TM(tm_eval, tm_t);
        }
        if (returning) {
            if (r0 < sc[F_ALPHA]) {
                wave_sync();
                if (lane == 0) { sc[F_ALPHA] = r0; sc[F_RT1] = r1; sc[F_RT2] = r2; }
                wave_sync();
            }
            returning = false;
            state = (int)sc[F_STATE];
        }
        if (state >= 5) {
            r0 = sc[F_ALPHA]; r1 = sc[F_RT1]; r2 = sc[F_RT2]; returning = true; depth--;
            if (depth < 0) break;
            continue;
        }
        // ---- descend into child state-1: (c3,c5) (c3,c6) (c4,c5) (c4,c6)
        {
#ifdef OBTG_MD_TIMING
            tm_t = __builtin_readcyclecounter();
#endif
            if (cur_depth != depth) {            // the walk came back up: fetch this frame's curves again
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   // other lanes' stores of this frame have landed
                for (int i = lane; i < 6 * K; i += kWave) cur[i] = f[i];
                cur_depth = depth;
                wave_sync();
#ifdef OBTG_MD_TIMING
                if (cur[0] != cur[0]) tm_fetch += 1;      // (keeps the loads ahead of the stamp)
                OBTG_TM(tm_fetch, tm_t);
#endif
            }
            const int ch = state - 1, h1 = ch >> 1, h2 = ch & 1;
            const double t1 = sc[F_T1], t2 = sc[F_T2];
            double* nf = f + FR;
            if (3 * K <= kWave) {                 // rows 0..2: curve 1 (x, y, z), rows 3..5: curve 2; level-parallel
                split_rows3_wave(cur, K, t1, h1, nxt, rl, il, sh_e);                 // (sh_e: 64 doubles, idle here)
                split_rows3_wave(cur + 3 * K, K, t2, h2, nxt + 3 * K, rl, il, sh_e);
            } else if (lane < 6) {
                const bool second = lane >= 3;
                split_row_lds(cur + lane * K, K, second ? t2 : t1, second ? h2 : h1, nxt + lane * K, sh_e + lane * K);
            }
            wave_sync();
            for (int i = lane; i < 6 * K; i += kWave) nf[i] = nxt[i];
            const double t1len = sc[F_T1H] - sc[F_T1L], t2len = sc[F_T2H] - sc[F_T2L];
            const double m1 = sc[F_T1L] + t1 * t1len, m2 = sc[F_T2L] + t2 * t2len;
            const double a_in = sc[F_ALPHA];
            const double n1l = h1 ? m1 : sc[F_T1L], n1h = h1 ? sc[F_T1H] : m1;
            const double n2l = h2 ? m2 : sc[F_T2L], n2h = h2 ? sc[F_T2H] : m2;
            wave_sync();
            if (lane == 0) {
                double* ns = sc + F_NSCAL;
                ns[F_T1L] = n1l; ns[F_T1H] = n1h; ns[F_T2L] = n2l; ns[F_T2H] = n2h;
                ns[F_ALPHA] = a_in; ns[F_STATE] = 0;
                sc[F_STATE] = state + 1;
            }
            // the child is evaluated next: its curves become `cur`
            double* tsw = cur; cur = nxt; nxt = tsw;
            depth++;
            cur_depth = depth;
            wave_sync();
            OBTG_TM(tm_desc, tm_t);
        }
    }
#ifdef OBTG_MD_TIMING
    calls = (int)(tm_gjk >> 10); dmax = (int)(tm_eval >> 10); status = (int)(tm_desc >> 10) | ((int)(tm_fetch >> 10) << 16);
#endif
    // (every lane, the same values to the same addresses: see the pull above)
    p.res[3 * k] = r0; p.res[3 * k + 1] = r1; p.res[3 * k + 2] = r2;
    if (p.info) { p.info[4 * k] = nodes; p.info[4 * k + 1] = calls; p.info[4 * k + 2] = dmax; p.info[4 * k + 3] = status; }
  }
}

// ---- _minDist, four children at a time (round 5) -------------------------------------------------------------------
// bezier.py:1283-1408 calls itself on ALL four children of a node it does not cut off -- the cut (lb >= alpha (1 - eps)) is
// taken inside the child, after the child's own gjkNew call and end-point distances -- so a child's gjkNew result, its split
// parameters and its end-point bound depend on the child's curves alone, never on alpha: the four can be worked out
// together when their parent is split, and the depth-first walk (which does depend on alpha, child after child) then only
// reads them.  k_min_dist_wave spends a whole wavefront on one gjkNew call at a time and a pair's search is a serial chain
// of them (8.9 k of a node's 14.8 k clocks, profiles/r05_experiments); here one 16-lane row of the wavefront takes each
// child (K <= 16), the four state machines advance in lockstep (gjk::run_quarter), and a node's two splits produce BOTH
// pieces of both curves in one level-parallel pass.  The walk visits the same nodes in the same order with the same
// values: res and info are those of k_min_dist_wave (test_min_dist_quad_form_is_the_wave_form and the reference fixtures).
//
// A frame's blob: the four half curves of an EXPANDED node, c3 c4 (curve 1 left, right) c5 c6 (curve 2), then the records
// of its four children (c3,c5) (c3,c6) (c4,c5) (c4,c6).
enum { R_CAP = 0, R_LB, R_T1, R_T2, R_UB, R_AM, R_NREC };
enum { Q_NSCAL = F_NSCAL };               // frame scalars: k_min_dist_wave's (F_STATE: the next child, 0..4)
__host__ __device__ constexpr int md_quad_blob(int K) { return 12 * K + 4 * R_NREC; }
constexpr int kMdQuadMaxK = 16;
// The scalars of the frames below the walk's: the first kMdScsLds levels in LDS, deeper ones behind their frame's blob in the
// global stack (round 6: at max_depth 128 the scalars were 10 of a worker's 15.4 KB of LDS and held a CU to ten workers;
// a search that ends is a few levels deep, one that runs into the depth cap pays two global round trips per level below 32).
constexpr int kMdScsLds = 32;
__host__ __device__ constexpr int md_quad_frame(int K) { return md_quad_blob(K) + Q_NSCAL; }     // doubles of a frame in the global stack
// doubles of split-parameter scratch per 16-lane row: 2 x 16 used; 34 (not 32 = 64 banks) so that the four rows' broadcast
// reads of the same element fall into four banks (PMC: a quarter of the LDS cycles were conflicts with 32)
constexpr int kMdShRow = 34;

// deCasteljauSplit of rows [row0, row0 + nrows) of a node's six coordinate rows (rows 0..2: curve 1 at t1, rows 3..5:
// curve 2 at t2), BOTH pieces kept: as split_rows3_wave_t, where the left piece's point L is lane (r, 0)'s value at level L
// and the right piece's point i is lane (r, i)'s value at level K - 1 - i -- never the same lane at the same level before
// the last one, so a level is still one store per lane.
template <int KC>
__device__ __forceinline__ void split_both_t(const double* c1, const double* c2, int K, double t1, double t2, double* blob,
                                             int row0, int nrows, double* dump)
{
    if (KC > 0) K = KC;
    const int lane = threadIdx.x & 63;
    const int rq = lane / K, il = lane - rq * K, row = row0 + rq;
    const bool valid = rq < nrows;
    const bool second = row >= 3;
    const int r3 = second ? row - 3 : row;
    double w = valid ? (second ? c2 : c1)[r3 * K + il] : 0.0;
    const double t = second ? t2 : t1, u = 1 - t;
    double* outL = blob + (second ? 6 * K : 0) + r3 * K;          // left piece's row; the right piece's is 3 K further
    double* outR = outL + 3 * K + il;
    const bool first = valid && il == 0;
    const int my_level = valid ? K - 1 - il : -1;
    double* mine = dump + lane;
    auto level = [&](int L) {
        double* a = first ? outL + L : (my_level == L ? outR : mine);
        *a = w;
        const double up = wave_next_lane(w);
        w = u * w + t * up;
    };
    if constexpr (KC > 0) {
#pragma unroll
        for (int L = 0; L < KC - 1; ++L) level(L);
    } else {
        for (int L = 0; L < K - 1; ++L) level(L);
    }
    // the last level's value: point K - 1 of the left piece and point 0 of the right one
    double* a = first ? outL + (K - 1) : mine;
    double* b = first ? outR : mine;
    *a = w; *b = w;
}

__device__ __forceinline__ void split_both(const double* c1, const double* c2, int K, double t1, double t2, double* blob,
                                           double* dump)
{
    const bool one_pass = 6 * K <= kWave;
    switch (K) {        // (wave-uniform)
#define OBTG_CASE(NC_) \
    case NC_: \
        if (NC_ <= kMdQuadMaxK) { \
            if (one_pass) split_both_t<NC_>(c1, c2, K, t1, t2, blob, 0, 6, dump); \
            else { split_both_t<NC_>(c1, c2, K, t1, t2, blob, 0, 3, dump); split_both_t<NC_>(c1, c2, K, t1, t2, blob, 3, 3, dump); } \
            return; \
        } \
        break;
        OBTG_NC_DYN(OBTG_CASE)
#undef OBTG_CASE
        default: break;
    }
    if (one_pass) split_both_t<0>(c1, c2, K, t1, t2, blob, 0, 6, dump);
    else { split_both_t<0>(c1, c2, K, t1, t2, blob, 0, 3, dump); split_both_t<0>(c1, c2, K, t1, t2, blob, 3, 3, dump); }
}

// np_sum_f of two sequences side by side (their quotients are independent chains: the two divisions of a step overlap)
template <class FA, class FB>
__device__ __forceinline__ void np_sum_f2(int n, FA qa, FB qb, double& sa, double& sb)
{
    if (n < 8) {
        double ra = 0.0, rb = 0.0;
        for (int i = 0; i < n; ++i) { ra += qa(i); rb += qb(i); }
        sa = ra; sb = rb;
        return;
    }
    double a0 = qa(0), a1 = qa(1), a2 = qa(2), a3 = qa(3), a4 = qa(4), a5 = qa(5), a6 = qa(6), a7 = qa(7);
    double b0 = qb(0), b1 = qb(1), b2 = qb(2), b3 = qb(3), b4 = qb(4), b5 = qb(5), b6 = qb(6), b7 = qb(7);
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
        a0 += qa(i); a1 += qa(i + 1); a2 += qa(i + 2); a3 += qa(i + 3); a4 += qa(i + 4); a5 += qa(i + 5); a6 += qa(i + 6); a7 += qa(i + 7);
        b0 += qb(i); b1 += qb(i + 1); b2 += qb(i + 2); b3 += qb(i + 3); b4 += qb(i + 4); b5 += qb(i + 5); b6 += qb(i + 6); b7 += qb(i + 7);
    }
    double ra = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
    double rb = ((b0 + b1) + (b2 + b3)) + ((b4 + b5) + (b6 + b7));
    for (; i < n; ++i) { ra += qa(i); rb += qb(i); }
    sa = ra; sb = rb;
}

// hull_param_wave for the call of one 16-lane row, both curves at once: lane l of the row takes control point l of c1 AND of
// c2 (the weights of a point are a chain of K - 1 divisions; the two curves' chains overlap); sh_e / sh_q: 32 doubles of
// the row's own.  Per curve the operations are hull_param's (bezier.py:1320-1351).
__device__ __forceinline__ void hull_param_quarter2(const double* c1, const double* c2, int K, const V3& cl1, const V3& cl2,
                                                    double* sh_e, double* sh_q, double& t1, double& t2)
{
    const int lane = threadIdx.x & 63, l = lane & 15;
    // exact row match first: lowest matching index
    const bool hit1 = l < K && c1[l] == cl1.x && c1[K + l] == cl1.y && c1[2 * K + l] == cl1.z;
    const bool hit2 = l < K && c2[l] == cl2.x && c2[K + l] == cl2.y && c2[2 * K + l] == cl2.z;
    const unsigned m1 = (unsigned)(__ballot(hit1) >> (lane & 48)) & 0xffffu;
    const unsigned m2 = (unsigned)(__ballot(hit2) >> (lane & 48)) & 0xffffu;
    if (m1) t1 = (double)(__ffs((int)m1) - 1) / (double)(K - 1);
    if (m2) t2 = (double)(__ffs((int)m2) - 1) / (double)(K - 1);
    if (m1 && m2) return;
    if (l < K) {
        const double dx = cl1.x - c1[l], dy = cl1.y - c1[K + l], dz = cl1.z - c1[2 * K + l];
        double s = 0.0;
        s += dx * dx; s += dy * dy; s += dz * dz;
        sh_e[l] = __builtin_sqrt(s);
        const double ex = cl2.x - c2[l], ey = cl2.y - c2[K + l], ez = cl2.z - c2[2 * K + l];
        double u = 0.0;
        u += ex * ex; u += ey * ey; u += ez * ez;
        sh_e[16 + l] = __builtin_sqrt(u);
    }
    wave_sync();
    if (l < K) {
        const double ea = sh_e[l], eb = sh_e[16 + l];
        double a1, b1, a2, b2;
        np_sum_f2(l, [&](int j) { return ea / sh_e[j]; }, [&](int j) { return eb / sh_e[16 + j]; }, a1, b1);
        np_sum_f2(K - l - 1, [&](int j) { return ea / sh_e[l + 1 + j]; }, [&](int j) { return eb / sh_e[16 + l + 1 + j]; }, a2, b2);
        const double Wa = 1 / (1 + a1 + a2), Wb = 1 / (1 + b1 + b2);
        sh_q[l] = Wa * (double)l / (double)K;
        sh_q[16 + l] = Wb * (double)l / (double)K;
    }
    wave_sync();
    if (!m1) t1 = np_sum(sh_q, K);
    if (!m2) t2 = np_sum(sh_q + 16, K);
}

// hull_param_quarter2 for a control-point count known at compile time.  The weights of point l are 1 / (1 + sum_{j<l} e_l/e_j
// + sum_{j>l} e_l/e_j) with numpy's association of each sum (np_sum_f: left to right below eight terms, else the first eight
// pairwise and the rest left to right) -- K - 1 divisions per lane.  As loops of lane-dependent length (hull_param_quarter2)
// the wavefront walks through the longest of every kind, a division at a time, and the two calls were 11.6 k clocks of an
// evaluation's 33 k (profiles/r05_experiments/mindist_quad_phases.txt).  Here every lane forms ALL K quotients of both curves
// -- 2 K independent divisions, unrolled -- and the sums pick their terms by predicate in the order numpy adds them.
// PLANAR: every z of both curves and of both closest points is an exact zero (2-D curves, bezier.py:1294-1308): the z loads,
// the z clause of the row match (0 == 0) and the `+= dz * dz` (s + 0 = s) are dropped -- the same values.
template <int K, bool TWO = true, bool PLANAR = false>      // TWO = false: curve 1 only (c2 / cl2 / t2 unused: the curve <-> polygon search)
__device__ __forceinline__ void hull_param_quarter2_t(const double* c1, const double* c2, const V3& cl1, const V3& cl2,
                                                      double* sh_e, double* sh_q, double& t1, double& t2)
{
    static_assert(K >= 2 && K <= 16, "a 16-lane row per curve pair");
    const int lane = threadIdx.x & 63, l = lane & 15;
    const int li = l < K ? l : 0;                        // (lanes past the curve repeat point 0; nothing of theirs is used)
    const double ax = c1[li], ay = c1[K + li], az = PLANAR ? 0.0 : c1[2 * K + li];
    const double bx = TWO ? c2[li] : 0.0, by = TWO ? c2[K + li] : 0.0, bz = (TWO && !PLANAR) ? c2[2 * K + li] : 0.0;
    const bool hit1 = l < K && ax == cl1.x && ay == cl1.y && (PLANAR || az == cl1.z);
    const bool hit2 = TWO && l < K && bx == cl2.x && by == cl2.y && (PLANAR || bz == cl2.z);
    const unsigned m1 = (unsigned)(__ballot(hit1) >> (lane & 48)) & 0xffffu;
    const unsigned m2 = TWO ? (unsigned)(__ballot(hit2) >> (lane & 48)) & 0xffffu : 1u;
    if (m1) t1 = (double)(__ffs((int)m1) - 1) / (double)(K - 1);
    if (TWO && m2) t2 = (double)(__ffs((int)m2) - 1) / (double)(K - 1);
    if (m1 && m2) return;
    double ea, eb;
    {
        const double dx = cl1.x - ax, dy = cl1.y - ay, dz = cl1.z - az;
        double s = 0.0;
        s += dx * dx; s += dy * dy;
        if constexpr (!PLANAR) s += dz * dz;
        ea = __builtin_sqrt(s);
        eb = 1.0;
        if constexpr (TWO) {
            const double ex = cl2.x - bx, ey = cl2.y - by, ez = cl2.z - bz;
            double u = 0.0;
            u += ex * ex; u += ey * ey;
            if constexpr (!PLANAR) u += ez * ez;
            eb = __builtin_sqrt(u);
        }
    }
    sh_e[l] = ea;
    if constexpr (TWO) sh_e[16 + l] = eb;
    // The 2 K quotients e_l / e_j of a lane are IEEE divisions, which the compiler expands (v_div_scale x 2, v_rcp, two Newton steps
    // on the reciprocal, q0 = n r, rem = n - d q0, q = q0 + rem r through v_div_fmas, v_div_fixup: 11 instructions).  Everything up
    // to the refined reciprocal depends on the DENOMINATOR alone, and when numerator and denominator are ordinary numbers
    // (v_div_scale scales nothing, v_div_fmas is a plain fused multiply-add, v_div_fixup passes its operand) the quotient is
    // fma(fma(-d, n r, n), r, n r) with that reciprocal: the same instructions on the same values, so the same bits.  The lane
    // that owns e_j refines 1 / e_j once (5 instructions) and leaves it beside e_j; every quotient is then 3 instructions
    // instead of 11 (round 6: this block was 727 of the ~2200 instructions of an expansion, and a call of the Jacobian's
    // size is issue bound).  Distances outside [2^-300, 2^300] (v_div_scale leaves operands alone while their exponents are less than
    // 768 apart and neither is near the ends of the range) -- a zero where the closest point IS a control point of a curve
    // whose parameter comes from the sums, an overflow, a NaN -- send the whole wavefront through the divisions as written.
    bool fast = true;
#ifndef OBTG_MD_PLAIN_DIVISIONS
    {
        const bool ok_a = ea >= 0x1p-300 && ea <= 0x1p300, ok_b = !TWO || (eb >= 0x1p-300 && eb <= 0x1p300);
        fast = __ballot(l < K && ((!m1 && !ok_a) || (TWO && !m2 && !ok_b))) == 0ull;
    }
#else
    fast = false;
#endif
    double qa[K], qb[K];
    if (fast) {
        auto refined_rcp = [](double d) {
            double r = __builtin_amdgcn_rcp(d);
            double e = __builtin_fma(-d, r, 1.0);
            r = __builtin_fma(r, e, r);
            e = __builtin_fma(-d, r, 1.0);
            return __builtin_fma(r, e, r);
        };
        sh_q[l] = refined_rcp(ea);
        if constexpr (TWO) sh_q[16 + l] = refined_rcp(eb);
        wave_sync();
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const double da = sh_e[j], ra = sh_q[j];
            const double q0 = ea * ra;
            qa[j] = __builtin_fma(__builtin_fma(-da, q0, ea), ra, q0);
            if constexpr (TWO) {
                const double db = sh_e[16 + j], rb = sh_q[16 + j];
                const double p0 = eb * rb;
                qb[j] = __builtin_fma(__builtin_fma(-db, p0, eb), rb, p0);
            } else qb[j] = 0.0;
        }
        wave_sync();                                    // (the reciprocals have been read: sh_q takes the weights below)
    } else {
        wave_sync();
#pragma unroll
        for (int j = 0; j < K; ++j) { qa[j] = ea / sh_e[j]; qb[j] = TWO ? eb / sh_e[16 + j] : 0.0; }
    }
    // terms j < l (l of them)
    auto sum_lo = [&](const double (&q)[K]) {
        double r = 0.0;
#pragma unroll
        for (int j = 0; j < (K < 8 ? K : 7); ++j) r = j < l ? r + q[j] : r;
        if constexpr (K > 8) {
            double P = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
#pragma unroll
            for (int j = 8; j < K - 1; ++j) P = j < l ? P + q[j] : P;
            r = l >= 8 ? P : r;
        }
        return r;
    };
    // terms j > l (K - 1 - l of them)
    auto sum_hi = [&](const double (&q)[K]) {
        double r = 0.0;
#pragma unroll
        for (int j = 1; j < K; ++j) r = j > l ? r + q[j] : r;
        if constexpr (K > 8) {                            // eight or more terms: lanes l <= K - 9, the first eight start at l + 1
            double x[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                x[k] = q[1 + k];
#pragma unroll
                for (int l0 = 1; l0 <= K - 9; ++l0) x[k] = l == l0 ? q[l0 + 1 + k] : x[k];
            }
            double P = ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
#pragma unroll
            for (int j = 9; j < K; ++j) P = j >= l + 9 ? P + q[j] : P;
            r = K - 1 - l >= 8 ? P : r;
        }
        return r;
    };
    const double a1 = sum_lo(qa), a2 = sum_hi(qa);
    const double Wa = 1 / (1 + a1 + a2);
    sh_q[l] = Wa * (double)l / (double)K;
    if constexpr (TWO) {
        const double b1 = sum_lo(qb), b2 = sum_hi(qb);
        const double Wb = 1 / (1 + b1 + b2);
        sh_q[16 + l] = Wb * (double)l / (double)K;
    }
    wave_sync();
    if (!m1) t1 = np_sum(sh_q, K);
    if (TWO && !m2) t2 = np_sum(sh_q + 16, K);
}

// the curve's parameter alone (curve <-> polygon search)
template <bool PLANAR = false>
__device__ __forceinline__ double hull_param_row(const double* c1, int K, const V3& cl1, double* sh_e, double* sh_q)
{
    double t1 = 0.0, t2 = 0.0;
    switch (K) {        // (wave-uniform)
#define OBTG_CASE(NC_) case NC_: hull_param_quarter2_t<NC_, false, PLANAR>(c1, c1, cl1, cl1, sh_e, sh_q, t1, t2); return t1;
        OBTG_NC_DYN(OBTG_CASE)
#undef OBTG_CASE
        default: hull_param_quarter2(c1, c1, K, cl1, cl1, sh_e, sh_q, t1, t2);
    }
    return t1;
}

template <bool PLANAR = false>
__device__ __forceinline__ void hull_param_rows(const double* c1, const double* c2, int K, const V3& cl1, const V3& cl2,
                                                double* sh_e, double* sh_q, double& t1, double& t2)
{
    switch (K) {        // (wave-uniform)
#define OBTG_CASE(NC_) case NC_: hull_param_quarter2_t<NC_, true, PLANAR>(c1, c2, cl1, cl2, sh_e, sh_q, t1, t2); return;
        OBTG_NC_DYN(OBTG_CASE)
#undef OBTG_CASE
        default: hull_param_quarter2(c1, c2, K, cl1, cl2, sh_e, sh_q, t1, t2);
    }
}

// What a node's visit needs of it, for the four (curve 1, curve 2) pairs the rows of the wavefront name: the gjkNew call,
// the split parameters of its closest points (bezier.py:1313-1351), the end-point bound (_upperbound, bezier.py:1255-1280).
// o1 / o2: offsets of the row's two curves in `lds`; rec: the row's record.  Every lane of a row stores the row's (equal)
// values: no lane-dependent region (see the queue pull of k_min_dist_wave).
template <bool PLANAR = false, int KC = 0>      // KC: the control-point count when the kernel is built for one (0: any count up to 16)
__device__ __forceinline__ void md_eval_rows(const double* lds, int o1, int o2, int K, double eps, int max_iter, int md_cap,
                                             double* rec, double* sh_e, double* sh_q
#ifdef OBTG_MD_TIMING
                                             , unsigned long long* tm      // [0] the lockstep gjkNew, [1] the split parameters, [2] bound + record
#endif
)
{
    Ctx<MemLds> g;
    g.mem = MemLds{ lds };
    g.P1 = Poly{ o1, K, K, 1 };
    g.P2 = Poly{ o2, K, K, 1 };
    g.trace = nullptr; g.trace_cap = 0; g.n_support = 0;
    Result gr;
#ifdef OBTG_MD_TIMING
    const unsigned long long tq0 = __builtin_readcyclecounter();
#endif
    if constexpr (PLANAR) gjk::run_quarter2<MemLds>(g, max_iter, md_cap, gr);
    else gjk::run_quarter<MemLds>(g, max_iter, md_cap, gr);
#ifdef OBTG_MD_TIMING
    if (gr.dist == -1.0) rec[R_CAP] = 0.0;      // (keeps the call ahead of the stamp)
    tm[0] += __builtin_readcyclecounter() - tq0;
    const unsigned long long tq1 = __builtin_readcyclecounter();
#endif
    const bool cap = gr.status == OBTG_ST_MD_CAP || gr.status == OBTG_ST_CYCLE;
    double lb = eps, t1 = 0.5, t2 = 0.5;
    const double* c1 = lds + o1; const double* c2 = lds + o2;
    if (gr.flag > 0 && !cap) {
        lb = gr.dist;
        if constexpr (KC > 0) hull_param_quarter2_t<KC, true, PLANAR>(c1, c2, gr.c1, gr.c2, sh_e, sh_q, t1, t2);
        else hull_param_rows<PLANAR>(c1, c2, K, gr.c1, gr.c2, sh_e, sh_q, t1, t2);
    }
#ifdef OBTG_MD_TIMING
    if (t1 == -1.0) rec[R_CAP] = 0.0;
    tm[1] += __builtin_readcyclecounter() - tq1;
    const unsigned long long tq2 = __builtin_readcyclecounter();
#endif
    double dd[4];
    if constexpr (PLANAR) {          // (norm_seq with dz = 0: s + 0 * 0 = s)
        auto n2 = [](double ax, double ay, double bx, double by) {
            const double dx = ax - bx, dy = ay - by;
            double s = 0.0;
            s += dx * dx; s += dy * dy;
            return __builtin_sqrt(s);
        };
        dd[0] = n2(c1[0], c1[K], c2[0], c2[K]);
        dd[1] = n2(c1[0], c1[K], c2[K - 1], c2[2 * K - 1]);
        dd[2] = n2(c1[K - 1], c1[2 * K - 1], c2[0], c2[K]);
        dd[3] = n2(c1[K - 1], c1[2 * K - 1], c2[K - 1], c2[2 * K - 1]);
    } else {
        dd[0] = norm_seq(c1[0], c1[K], c1[2 * K], c2[0], c2[K], c2[2 * K]);
        dd[1] = norm_seq(c1[0], c1[K], c1[2 * K], c2[K - 1], c2[2 * K - 1], c2[3 * K - 1]);
        dd[2] = norm_seq(c1[K - 1], c1[2 * K - 1], c1[3 * K - 1], c2[0], c2[K], c2[2 * K]);
        dd[3] = norm_seq(c1[K - 1], c1[2 * K - 1], c1[3 * K - 1], c2[K - 1], c2[2 * K - 1], c2[3 * K - 1]);
    }
    int am = 0;
    for (int i = 1; i < 4; ++i) if (dd[i] < dd[am]) am = i;
    for (int i = 0; i < 4; ++i) if (dd[i] != dd[i]) { am = i; break; }
    rec[R_CAP] = cap ? 1.0 : 0.0; rec[R_LB] = lb; rec[R_T1] = t1; rec[R_T2] = t2; rec[R_UB] = dd[am]; rec[R_AM] = (double)am;
#ifdef OBTG_MD_TIMING
    wave_sync();
    if (rec[R_UB] == -1.0) rec[R_CAP] = 0.0;
    tm[2] += __builtin_readcyclecounter() - tq2;
#endif
}

#ifndef OBTG_MD_MIN_WAVES_PLANAR
#define OBTG_MD_MIN_WAVES_PLANAR 3     // workers per SIMD of an issue-bound planar call (K = 11: 162 registers, nothing spilled; four -- 128 registers + 116 B of scratch -- are no faster: 59.8 against 59.6 ms on the Jacobian list)
#endif
// PLANAR: every curve of the call has z == 0 in every control point (the host has looked): the planar gjkNew machine per row.
// W: worker waves per SIMD the registers are held to.  The planar form has two builds: W = 2 for calls that are bound by the chain
// of their longest search (a few pairs per worker: one evaluation's 4560 pairs; K = 11 takes 162 registers either way, but 2048
// workers leave each chain more of its SIMD), W = OBTG_MD_MIN_WAVES_PLANAR for calls bound by the chip's issue rate (the Jacobian's
// 114 000 searches) -- launch_min_dist picks by pairs per worker (OBTG_MD_MANY).  History: profiles/r06_experiments/mindist_steps.txt.
// KC: the control-point count the kernel is built for (0: any up to 16, the count-dependent parts behind wave-uniform switches).
// One kernel for every count took the registers of its largest case -- the split parameters' 2 K quotients per lane at K = 16 --
// and the degree-10 searches (K = 11) spilled for it.
template <bool PLANAR, int W, int KC>
__global__ __launch_bounds__(64, W) void k_min_dist_quad(const MdParams p)
{
    extern __shared__ double md_lds[];
    const int lane = threadIdx.x, q = lane >> 4;
    const int K = KC > 0 ? KC : p.K, BL = md_quad_blob(K), FRM = md_quad_frame(K);
    double* st = p.stack + (size_t)blockIdx.x * p.max_depth * FRM;
    double* sh_e = md_lds + 2 * BL;             // [4][kMdShRow]: [2][16] per row of the wavefront, the rows' banks apart
    double* sh_q = sh_e + 4 * kMdShRow;         // the same
    double* dump = sh_q + 4 * kMdShRow;         // [64]
    double* scs = dump + 64;                    // [min(max_depth, kMdScsLds)][Q_NSCAL] frame scalars; deeper frames: st + d * FRM + BL
  for (;;) {
    const int ticket = atomicAdd(p.queue, lane == 0 ? 1 : 0);      // (see k_min_dist_wave)
    const int slot = __builtin_amdgcn_readfirstlane(ticket);
    if (slot >= p.n_pairs) break;
    const int k = p.order ? p.order[slot] : slot;
    wave_sync();
    double* cur = md_lds;                       // [BL] blob of the frame `cur_depth`
    double* nxt = cur + BL;                     // [BL] blob being built
    // The pair's own curves as the blob of a frame "-1" whose four children are all the root: pieces c1 c1 c2 c2.  The
    // root is then visited like every other node (its record is child 0's), and the kernel has ONE copy of the evaluation.
    const double* ca = p.curves + (size_t)p.pa[k] * 3 * K;
    const double* cb = p.curves + (size_t)p.pb[k] * 3 * K;
    for (int i = lane; i < 3 * K; i += kWave) {
        const double a = ca[i], bq = cb[i];
        nxt[i] = a; nxt[3 * K + i] = a; nxt[6 * K + i] = bq; nxt[9 * K + i] = bq;
    }
    // The frame whose children the walk is going through lives in registers (the values are wave-uniform); `scs` holds the
    // frames below it.  Frame -1: the parameter square itself, split "at (1, 1)", whose child 0 is the root.
    double f_t1l = 0, f_t1h = 1, f_t2l = 0, f_t2h = 1, f_t1 = 1, f_t2 = 1, f_alpha = INFINITY, f_rt1 = -1, f_rt2 = -1;
    int f_next = 0;
    int depth = -1, cur_depth = -1;   // depth: the frame in registers; cur_depth: which frame's blob `cur` holds
    int eval_depth = -1;              // which frame's blob `nxt` is about to become
    int nodes = 0, calls = 0, dmax = 0, status = OBTG_MD_OK;
    double r0 = INFINITY, r1 = -1, r2 = -1;
    bool done = false;
#ifdef OBTG_MD_TIMING      // (variant builds: info[] then carries phase totals in units of 1024 clocks)
    unsigned long long tm_ev[3] = { 0, 0, 0 }, tm_eval = 0, tm_split = 0, tm_walk = 0, tm_store = 0, tm_fetch = 0;
#endif
    while (!done) {
        // ---- the four children of the blob in `nxt`, a row of the wavefront each
        wave_sync();
#ifdef OBTG_MD_TIMING
        unsigned long long tq = __builtin_readcyclecounter();
        md_eval_rows<PLANAR, KC>(md_lds, (int)(nxt - md_lds) + (q >> 1) * 3 * K, (int)(nxt - md_lds) + (2 + (q & 1)) * 3 * K, K, p.eps,
                     p.max_iter, p.md_cap, nxt + 12 * K + q * R_NREC, sh_e + q * kMdShRow, sh_q + q * kMdShRow, tm_ev);
#else
        md_eval_rows<PLANAR, KC>(md_lds, (int)(nxt - md_lds) + (q >> 1) * 3 * K, (int)(nxt - md_lds) + (2 + (q & 1)) * 3 * K, K, p.eps,
                     p.max_iter, p.md_cap, nxt + 12 * K + q * R_NREC, sh_e + q * kMdShRow, sh_q + q * kMdShRow);
#endif
        wave_sync();
#ifdef OBTG_MD_TIMING
        const unsigned long long tq3 = __builtin_readcyclecounter();
#endif
        if (eval_depth >= 0) {
            double* f = st + (size_t)eval_depth * FRM;
            for (int i = lane; i < BL; i += kWave) f[i] = nxt[i];
        }
        { double* tsw = cur; cur = nxt; nxt = tsw; }
        cur_depth = eval_depth;
#ifdef OBTG_MD_TIMING
        if (cur[12 * K] == -1.0) tm_eval += 1;      // (keeps the records ahead of the stamp)
        tm_store += __builtin_readcyclecounter() - tq3;
        tm_eval += __builtin_readcyclecounter() - tq;
        tq = __builtin_readcyclecounter();
#endif
        // ---- the depth-first walk (bezier.py:1283-1408 unrolled onto the frames), until a node has to be expanded -- its
        //      pieces go to `nxt`, the evaluation of its children is the next trip's -- or the search ends.
        //      A value returned to a frame: `if newAlpha < retval[0]: retval = ...` (bezier.py:1383-1406); to frame -1: the answer.
#define OBTG_MD_RETURN() \
    { if (depth < 0) { done = true; break; } \
      if (r0 < f_alpha) { f_alpha = r0; f_rt1 = r1; f_rt2 = r2; } \
      continue; }
        for (;;) {
            if (f_next >= 4) {                   // the four children are done: this frame's value goes to its parent
                r0 = f_alpha; r1 = f_rt1; r2 = f_rt2;
                depth--;
                if (depth < 0) { done = true; break; }
                const bool deep = depth >= kMdScsLds;
                if (deep) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                const double* sc = deep ? st + (size_t)depth * FRM + BL : scs + depth * Q_NSCAL;
                f_t1 = sc[F_T1]; f_t2 = sc[F_T2]; f_t1l = sc[F_T1L]; f_t1h = sc[F_T1H]; f_t2l = sc[F_T2L]; f_t2h = sc[F_T2H];
                f_alpha = sc[F_ALPHA]; f_rt1 = sc[F_RT1]; f_rt2 = sc[F_RT2]; f_next = (int)sc[F_STATE];
                OBTG_MD_RETURN()
            }
            // ---- child f_next of the frame: (c3,c5) (c3,c6) (c4,c5) (c4,c6), a node at depth + 1
            const int ch = f_next++, h1 = ch >> 1, h2 = ch & 1;
            if (depth + 2 > 1000) { r0 = r1 = r2 = -1; OBTG_MD_RETURN() }
            if (nodes >= p.max_nodes) { status = OBTG_MD_NODE_CAP; done = true; break; }
            nodes++;
            if (depth + 2 > dmax) dmax = depth + 2;
            if (cur_depth != depth) {            // the walk came back up: fetch this frame's blob again
#ifdef OBTG_MD_TIMING
                const unsigned long long tf0 = __builtin_readcyclecounter();
#endif
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                const double* f = st + (size_t)depth * FRM;
                for (int i = lane; i < BL; i += kWave) cur[i] = f[i];
                cur_depth = depth;
                wave_sync();
#ifdef OBTG_MD_TIMING
                if (cur[0] == -1.0) tm_fetch += 1;
                tm_fetch += __builtin_readcyclecounter() - tf0;
#endif
            }
            const double* rec = cur + 12 * K + ch * R_NREC;
            calls++;
            if (rec[R_CAP] != 0.0) { status = OBTG_MD_GJK_CAP; done = true; break; }
            const double t1len = f_t1h - f_t1l, t2len = f_t2h - f_t2l;
            const double m1 = f_t1l + f_t1 * t1len, m2 = f_t2l + f_t2 * t2len;
            const double n1l = h1 ? m1 : f_t1l, n1h = h1 ? f_t1h : m1;
            const double n2l = h2 ? m2 : f_t2l, n2h = h2 ? f_t2h : m2;
            const double lb = rec[R_LB];
            double t1 = rec[R_T1], t2 = rec[R_T2];
            const int am = (int)rec[R_AM];
            const double ub = rec[R_UB], t1loc = (am >> 1) ? 1.0 : 0.0, t2loc = (am & 1) ? 1.0 : 0.0;
            double alpha = f_alpha, nT1, nT2;
            if (ub <= alpha) {
                alpha = ub;
                nT1 = (1 - t1loc) * n1l + t1loc * n1h;
                nT2 = (1 - t2loc) * n2l + t2loc * n2h;
            } else { nT1 = -1; nT2 = -1; }
            if (lb >= alpha * (1 - p.eps)) { r0 = alpha; r1 = nT1; r2 = nT2; OBTG_MD_RETURN() }
            if (depth + 2 >= p.max_depth) { status = OBTG_MD_DEPTH_CAP; r0 = alpha; r1 = nT1; r2 = nT2; done = true; break; }
            if (t1 != t1) t1 = 0;
            if (t2 != t2) t2 = 0;
            // ---- expand the child: both pieces of both its curves to `nxt`; this frame goes to `scs`, the child's into the registers
#ifdef OBTG_MD_TIMING
            const unsigned long long ts0 = __builtin_readcyclecounter();
#endif
            if constexpr (KC > 0) {
                if constexpr (6 * KC <= kWave) split_both_t<KC>(cur + h1 * 3 * K, cur + (2 + h2) * 3 * K, K, t1, t2, nxt, 0, 6, dump);
                else {
                    split_both_t<KC>(cur + h1 * 3 * K, cur + (2 + h2) * 3 * K, K, t1, t2, nxt, 0, 3, dump);
                    split_both_t<KC>(cur + h1 * 3 * K, cur + (2 + h2) * 3 * K, K, t1, t2, nxt, 3, 3, dump);
                }
            } else split_both(cur + h1 * 3 * K, cur + (2 + h2) * 3 * K, K, t1, t2, nxt, dump);
#ifdef OBTG_MD_TIMING
            wave_sync();
            if (nxt[0] == -1.0) tm_split += 1;
            tm_split += __builtin_readcyclecounter() - ts0;
#endif
            if (depth >= 0) {
                double* sc = depth >= kMdScsLds ? st + (size_t)depth * FRM + BL : scs + depth * Q_NSCAL;
                sc[F_T1] = f_t1; sc[F_T2] = f_t2; sc[F_T1L] = f_t1l; sc[F_T1H] = f_t1h; sc[F_T2L] = f_t2l; sc[F_T2H] = f_t2h;
                sc[F_ALPHA] = f_alpha; sc[F_RT1] = f_rt1; sc[F_RT2] = f_rt2; sc[F_STATE] = (double)f_next;
            }
            f_t1 = t1; f_t2 = t2; f_t1l = n1l; f_t1h = n1h; f_t2l = n2l; f_t2h = n2h;
            f_alpha = alpha; f_rt1 = nT1; f_rt2 = nT2; f_next = 0;
            depth++;
            eval_depth = depth;
            break;
        }
#undef OBTG_MD_RETURN
#ifdef OBTG_MD_TIMING
        tm_walk += __builtin_readcyclecounter() - tq;      // (includes the split)
#endif
    }
#ifdef OBTG_MD_TIMING
    // gjk_calls <- the lockstep gjkNew loops, depth <- a trip's evaluation in all (gjkNew included), status <- split | walk << 16
#if OBTG_MD_TIMING == 2     // split parameters, bound + record, blob store, blob re-fetch
    calls = (int)(tm_ev[1] >> 10); dmax = (int)(tm_ev[2] >> 10); status = (int)(tm_store >> 10) | ((int)(tm_fetch >> 10) << 16);
#else
    calls = (int)(tm_ev[0] >> 10); dmax = (int)(tm_eval >> 10); status = (int)(tm_split >> 10) | ((int)(tm_walk >> 10) << 16);
#endif
#endif
    // (every lane, the same values to the same addresses)
    p.res[3 * k] = r0; p.res[3 * k + 1] = r1; p.res[3 * k + 2] = r2;
    if (p.info) { p.info[4 * k] = nodes; p.info[4 * k + 1] = calls; p.info[4 * k + 2] = dmax; p.info[4 * k + 3] = status; }
  }
}

// -------------------------------------------------------------------------------------
//  Robust curve <-> curve minimum distance: an opt-in replacement for
//  _minDist that does not inherit gjkNew's non-minimal distances and unbounded loops (SURVEY.md 8(a)
//  G2/G3: 458 of 1499 reference distances are not minimal; some pairs never return).
//  Breadth-first branch & bound on the parameter square, one pair per wavefront, one node per lane:
//    node      = [t1, t1 + 2^-l] x [t2, t2 + 2^-l]; its two sub-curves come from the originals by two
//                de Casteljau splits per coordinate (nothing but (t1, t2, l) is stored);
//    upper     = distances between the four end-point pairs of the two sub-curves (points ON the
//                curves): alpha = wave minimum;
//    lower     = gap of the two control polygons projected on d = (mid-point difference): a valid bound
//                on the distance of the convex hulls, hence of the sub-curves, that closes
//                quadratically with the node size near a regular minimum;
//    a node survives while lower < alpha (1 - eps) and is then cut into its four children.
//  When the frontier empties, alpha is within a relative eps of the true minimum, or below eps times the
//  largest coordinate (the curves touch) (status OK); the node
//  budget / frontier capacity / level cap end the search with the best alpha so far and a status.
// -------------------------------------------------------------------------------------
struct MdrParams {
    const double* __restrict__ curves;   // [n_curves][3][K]
    const int* __restrict__ pa;
    const int* __restrict__ pb;
    int n_pairs, K, max_nodes, max_level, cap;
    double eps;
    double* frontier;                     // [n_pairs][2][cap][3]  (t1, t2, level)
    double* __restrict__ res;             // [n_pairs][3]  (dist, t1, t2)
    int* __restrict__ info;               // [n_pairs][4]  (nodes, levels, max frontier, status)
};

// sub-curve of one coordinate row on [a, a + w]: left part of a split at b = a + w, then the right
// part of that at a / b; row in/out through `x` (K doubles, lane-private LDS)
__device__ __forceinline__ void subcurve_row(double* x, int K, double a, double w)
{
    const double b = a + w;
    if (b < 1.0) {                                   // keep the left part [0, b]
        for (int sz = K; sz > 1; --sz)
            for (int i = K - 1; i >= K - sz + 1; --i) x[i] = (1 - b) * x[i - 1] + b * x[i];
    }
    if (a > 0.0) {                                   // of that, the right part [a / b, 1]
        const double u = a / b;
        for (int sz = K; sz > 1; --sz)
            for (int i = 0; i < sz - 1; ++i) x[i] = (1 - u) * x[i] + u * x[i + 1];
    }
}

// the same with the row in registers (K a compile-time constant): the loops of subcurve_row, element for element
template <int KC>
__device__ __forceinline__ void subcurve_row_reg(double (&x)[KC], double a, double w)
{
    const double b = a + w;
    if (b < 1.0) {
#pragma unroll
        for (int sz = KC; sz > 1; --sz)
#pragma unroll
            for (int i = KC - 1; i >= KC - sz + 1; --i) x[i] = (1 - b) * x[i - 1] + b * x[i];
    }
    if (a > 0.0) {
        const double u = a / b;
#pragma unroll
        for (int sz = KC; sz > 1; --sz)
#pragma unroll
            for (int i = 0; i < sz - 1; ++i) x[i] = (1 - u) * x[i] + u * x[i + 1];
    }
}

// KC > 0: the control-point count is a compile-time constant and a node's six sub-curve rows live in REGISTERS (round 5).
// The generic form (KC = 0) keeps every lane's rows in LDS: a sub-curve is K (K - 1) dependent multiply-adds, each an LDS
// round trip, twelve rows per node (six per pass) -- ~95 us per 64-node step of a wave that has its SIMD to itself, and the
// longest pair of the C5-sized sweep (13 945 nodes, 27 levels) WAS the launch: 20.8 ms alone, 20.8 ms with the other 4559
// pairs beside it (tools/robust_stats_probe.py).  In registers the same loops are register arithmetic with the elements of a
// level independent of one another, and pass B reuses the rows pass A formed.  The operations per element and their order
// are those of the LDS form (this unit is compiled with -ffp-contract=off): identical results, node and level counts
// (test_min_dist_robust_register_form_is_the_lds_form; OBTG_MDR_GENERIC=1 selects the LDS form).
template <int KC>
__global__ __launch_bounds__(64) void k_min_dist_robust(const MdrParams p)
{
    extern __shared__ double mr_lds[];
    __shared__ int s_count;
    const int k = blockIdx.x, lane = threadIdx.x, K = KC > 0 ? KC : p.K;
    const int pitch = (3 * K) | 1;
    double* orig = mr_lds;                         // [2][3][K]
    double* work = orig + 6 * K + lane * pitch;    // per lane: row scratch [K], projections [2][K]
    const double* ca = p.curves + (size_t)p.pa[k] * 3 * K;
    const double* cb = p.curves + (size_t)p.pb[k] * 3 * K;
    double scale = 0.0;
    for (int i = lane; i < 3 * K; i += kWave) {
        const double a = ca[i], bq = cb[i];
        orig[i] = a; orig[3 * K + i] = bq;
        scale = fmax(scale, fmax(__builtin_fabs(a), __builtin_fabs(bq)));
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) scale = fmax(scale, __shfl_xor(scale, m));
    const double abs_tol = p.eps * scale;          // "the curves touch": a relative criterion cannot close on distance 0
    double* fa = p.frontier + (size_t)k * 2 * p.cap * 3;
    double* fb = fa + (size_t)p.cap * 3;
    if (lane == 0) { fa[0] = 0.0; fa[1] = 0.0; fa[2] = 0.0; }
    __syncthreads();
    int n_a = 1, nodes = 0, levels = 0, front_max = 1, status = OBTG_MD_OK;
    double alpha = INFINITY, bt1 = -1, bt2 = -1;
    while (n_a > 0) {
        if (lane == 0) s_count = 0;
        __syncthreads();
        for (int base = 0; base < n_a; base += kWave) {
            const bool valid = base + lane < n_a;
            const int ni = valid ? base + lane : base;
            const double t1 = fa[3 * ni], t2 = fa[3 * ni + 1];
            const int lev = (int)fa[3 * ni + 2];
            const double w = __builtin_ldexp(1.0, -lev);
            // pass A: end points and a middle control point of both sub-curves
            double e[2][2][3], mid[2][3];
            double sub[2][3][KC > 0 ? KC : 1];          // KC > 0: the node's six sub-curve rows, kept for pass B
            if constexpr (KC > 0) {
#pragma unroll
                for (int cv = 0; cv < 2; ++cv)
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
#pragma unroll
                        for (int i = 0; i < KC; ++i) sub[cv][q][i] = orig[(cv * 3 + q) * KC + i];
                        subcurve_row_reg<KC>(sub[cv][q], cv ? t2 : t1, w);
                        e[cv][0][q] = sub[cv][q][0]; e[cv][1][q] = sub[cv][q][KC - 1];
                        mid[cv][q] = 0.5 * (sub[cv][q][0] + sub[cv][q][KC - 1]);
                    }
            } else {
                for (int cv = 0; cv < 2; ++cv)
                    for (int q = 0; q < 3; ++q) {
                        for (int i = 0; i < K; ++i) work[i] = orig[(cv * 3 + q) * K + i];
                        subcurve_row(work, K, cv ? t2 : t1, w);
                        e[cv][0][q] = work[0]; e[cv][1][q] = work[K - 1];
                        mid[cv][q] = 0.5 * (work[0] + work[K - 1]);
                    }
            }
            double best = INFINITY, b1 = -1, b2 = -1;
            for (int i1 = 0; i1 < 2; ++i1)
                for (int i2 = 0; i2 < 2; ++i2) {
                    const double dx = e[0][i1][0] - e[1][i2][0], dy = e[0][i1][1] - e[1][i2][1], dz = e[0][i1][2] - e[1][i2][2];
                    const double dd = __builtin_sqrt(dx * dx + dy * dy + dz * dz);
                    if (dd < best) { best = dd; b1 = t1 + i1 * w; b2 = t2 + i2 * w; }
                }
            if (!valid) best = INFINITY;
            // wave minimum of the upper bounds (lowest lane among equals)
            double wb = best;
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) wb = fmin(wb, __shfl_xor(wb, m));
            if (wb < alpha) {
                const unsigned long long who = __ballot(best == wb);
                const int src = __ffsll((long long)who) - 1;
                alpha = wb; bt1 = __shfl(b1, src); bt2 = __shfl(b2, src);
            }
            // pass B: projections of both control polygons on d = mid1 - mid2
            double d[3] = { mid[0][0] - mid[1][0], mid[0][1] - mid[1][1], mid[0][2] - mid[1][2] };
            const double dn = __builtin_sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            double lb = 0.0;
            if (dn > 0.0) {
                double lo1 = INFINITY, hi2 = -INFINITY;
                if constexpr (KC > 0) {
                    const double dq0 = d[0] / dn, dq1 = d[1] / dn, dq2 = d[2] / dn;
#pragma unroll
                    for (int i = 0; i < KC; ++i) {            // (the LDS form's pr[] += dq * row, q = 0, 1, 2, from 0.0)
                        double s1 = 0.0, s2 = 0.0;
                        s1 += dq0 * sub[0][0][i]; s1 += dq1 * sub[0][1][i]; s1 += dq2 * sub[0][2][i];
                        s2 += dq0 * sub[1][0][i]; s2 += dq1 * sub[1][1][i]; s2 += dq2 * sub[1][2][i];
                        lo1 = fmin(lo1, s1); hi2 = fmax(hi2, s2);
                    }
                } else {
                    double* pr = work + K;                  // [2][K]
                    for (int i = 0; i < 2 * K; ++i) pr[i] = 0.0;
                    for (int cv = 0; cv < 2; ++cv)
                        for (int q = 0; q < 3; ++q) {
                            for (int i = 0; i < K; ++i) work[i] = orig[(cv * 3 + q) * K + i];
                            subcurve_row(work, K, cv ? t2 : t1, w);
                            const double dq = d[q] / dn;
                            for (int i = 0; i < K; ++i) pr[cv * K + i] += dq * work[i];
                        }
                    for (int i = 0; i < K; ++i) { lo1 = fmin(lo1, pr[i]); hi2 = fmax(hi2, pr[K + i]); }
                }
                lb = fmax(0.0, (lo1 - hi2) * (1.0 - 1e-12));
            }
            const bool keep = valid && lb < alpha * (1 - p.eps) && alpha > abs_tol;
            if (keep) {
                if (lev + 1 > p.max_level) status = OBTG_MD_DEPTH_CAP;
                else {
                    const int pos = atomicAdd(&s_count, 4);
                    if (pos + 4 <= p.cap) {
                        const double h = 0.5 * w;
                        for (int c4 = 0; c4 < 4; ++c4) {
                            fb[3 * (pos + c4)] = t1 + (c4 >> 1) * h;
                            fb[3 * (pos + c4) + 1] = t2 + (c4 & 1) * h;
                            fb[3 * (pos + c4) + 2] = (double)(lev + 1);
                        }
                    } else status = OBTG_MD_NODE_CAP;
                }
            }
        }
        nodes += n_a;
        levels++;
        __syncthreads();
        // statuses are per lane: any lane hitting a cap ends the search with the best alpha so far
        const unsigned long long capped = __ballot(status != OBTG_MD_OK);
        if (capped) { status = __shfl(status, __ffsll((long long)capped) - 1); break; }
        n_a = min(s_count, p.cap);
        if (n_a > front_max) front_max = n_a;
        if (nodes + n_a > p.max_nodes && n_a > 0) { status = OBTG_MD_NODE_CAP; break; }
        double* tsw = fa; fa = fb; fb = tsw;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __syncthreads();
    }
    if (lane == 0) {
        p.res[3 * k] = alpha; p.res[3 * k + 1] = bt1; p.res[3 * k + 2] = bt2;
        if (p.info) { p.info[4 * k] = nodes; p.info[4 * k + 1] = levels; p.info[4 * k + 2] = front_max; p.info[4 * k + 3] = status; }
    }
}


// -------------------------------------------------------------------------------------
//  True hull distance (gjk_true.h; NOT the reference's gjkNew): one pair per lane, point sets in global memory
//  (SoA per polygon as for k_gjk_pairs).  Per pair: flag (1 separated / 0 intersecting), closest points, distance
//  = the hull distance within a relative eps, `lower` = the proven lower bound at exit, iterations, status.
// -------------------------------------------------------------------------------------
struct GjkTrueParams {
    const double* __restrict__ soa;
    const int* __restrict__ off;
    const int* __restrict__ pa;
    const int* __restrict__ pb;
    int n_pairs, max_iter;
    double eps;
    int* __restrict__ flag;
    double* __restrict__ p1;
    double* __restrict__ p2;
    double* __restrict__ dist;
    double* __restrict__ lower;
    int* iters;
    int* status;
};

struct SoaSet {          // one polygon of the SoA table: x[K] y[K] z[K]
    const double* b;
    int K;
    __device__ __forceinline__ tgjk::P3 operator()(int i) const { return tgjk::P3{ b[i], b[K + i], b[2 * K + i] }; }
};

template <class S1, class S2>
struct SupportScan {
    S1 s1; S2 s2; int K1, K2;
    __device__ __forceinline__ void operator()(const tgjk::P3& d, int& i1, int& i2) const
    {
        i1 = 0; i2 = 0;
        double m1 = tgjk::dot(s1(0), d), m2 = -tgjk::dot(s2(0), d);
        for (int i = 1; i < K1; ++i) { const double c = tgjk::dot(s1(i), d); if (c > m1) { m1 = c; i1 = i; } }
        for (int i = 1; i < K2; ++i) { const double c = -tgjk::dot(s2(i), d); if (c > m2) { m2 = c; i2 = i; } }
    }
};

__global__ __launch_bounds__(256) void k_gjk_true_pairs(const GjkTrueParams p)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= p.n_pairs) return;
    const int a = p.pa[k], b = p.pb[k];
    const int oa = p.off[a], ob = p.off[b];
    const SoaSet s1{ p.soa + 3 * oa, p.off[a + 1] - oa }, s2{ p.soa + 3 * ob, p.off[b + 1] - ob };
    double scale = 0.0;
    for (int i = 0; i < 3 * s1.K; ++i) scale = fmax(scale, __builtin_fabs(s1.b[i]));
    for (int i = 0; i < 3 * s2.K; ++i) scale = fmax(scale, __builtin_fabs(s2.b[i]));
    const SupportScan<SoaSet, SoaSet> sup{ s1, s2, s1.K, s2.K };
    const tgjk::Result r = tgjk::true_distance(sup, s1, s2, p.eps, p.eps * scale, p.max_iter);
    p.flag[k] = r.flag;
    p.p1[3 * k] = r.c1.x; p.p1[3 * k + 1] = r.c1.y; p.p1[3 * k + 2] = r.c1.z;
    p.p2[3 * k] = r.c2.x; p.p2[3 * k + 1] = r.c2.y; p.p2[3 * k + 2] = r.c2.z;
    p.dist[k] = r.dist;
    if (p.lower) p.lower[k] = r.lower;
    if (p.iters) p.iters[k] = r.iters;
    if (p.status) p.status[k] = r.status;
}

// -------------------------------------------------------------------------------------
//  Robust curve <-> polygon minimum distance (NOT the reference's `_minDist2Poly`, which
//  inherits gjkNew's non-minimal distances as "lower bounds" and its unbounded loops, bezier.py:1411-1496).
//  Breadth-first branch & bound on the curve parameter, one pair per wavefront, one node per lane.  A node is a
//  sub-curve [t, t + 2^-level]; with the true hull distance of gjk_true.h both bounds are valid for the curve itself:
//    upper = distance from the sub-curve's two end points (points ON the curve) to the polygon's hull,
//    lower = the certified lower bound of the distance between the sub-curve's control hull and the polygon's hull.
//  Survivors (lower < alpha (1 - eps)) are halved into a ping-pong frontier in global memory.  Ends when the
//  frontier is empty or alpha <= eps x (largest coordinate) (the curve touches the polygon); caps return the best
//  upper bound with a status.
// -------------------------------------------------------------------------------------
struct Md2rParams {
    const double* __restrict__ curves;   // [n_curves][3][K]
    const double* __restrict__ soa;      // polygons, SoA
    const int* __restrict__ off;
    const int* __restrict__ pc;
    const int* __restrict__ pp;
    int n_pairs, K, max_nodes, max_level, cap;
    double eps;
    double* frontier;                     // [n_pairs][2][cap][2]  (t, level)
    double* __restrict__ res;             // [n_pairs][5]  (dist, t, closest point on the polygon[3])
    int* __restrict__ info;               // [n_pairs][4]  (nodes, levels, max frontier, status)
};

struct LdsRows {         // a lane's sub-curve: rows x[K] y[K] z[K] in LDS
    const double* b;
    int K;
    __device__ __forceinline__ tgjk::P3 operator()(int i) const { return tgjk::P3{ b[i], b[K + i], b[2 * K + i] }; }
};

struct OnePoint {
    tgjk::P3 q;
    __device__ __forceinline__ tgjk::P3 operator()(int) const { return q; }
};

// KC > 0: the node's three sub-curve rows are formed in registers (subcurve_row_reg: see k_min_dist_robust) and then put into
// the lane's LDS rows, where the two true-distance searches read them by support index; KC = 0: formed in the LDS rows.
template <int KC>
__global__ __launch_bounds__(64) void k_min_dist2poly_robust(const Md2rParams p)
{
    extern __shared__ double mr_lds[];
    __shared__ int s_count;
    const int k = blockIdx.x, lane = threadIdx.x, K = KC > 0 ? KC : p.K;
    const int pitch = (3 * K) | 1;
    const int po = p.off[p.pp[k]], Kp = p.off[p.pp[k] + 1] - po;
    double* orig = mr_lds;                         // [3][K]
    double* poly = orig + 3 * K;                   // [3][Kp]
    double* work = poly + 3 * Kp + lane * pitch;   // per lane: the node's sub-curve [3][K]
    const double* cc = p.curves + (size_t)p.pc[k] * 3 * K;
    double scale = 0.0;
    for (int i = lane; i < 3 * K; i += kWave) { const double a = cc[i]; orig[i] = a; scale = fmax(scale, __builtin_fabs(a)); }
    for (int i = lane; i < 3 * Kp; i += kWave) { const double a = p.soa[3 * po + i]; poly[i] = a; scale = fmax(scale, __builtin_fabs(a)); }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) scale = fmax(scale, __shfl_xor(scale, m));
    const double abs_tol = p.eps * scale;
    double* fa = p.frontier + (size_t)k * 2 * p.cap * 2;
    double* fb = fa + (size_t)p.cap * 2;
    if (lane == 0) { fa[0] = 0.0; fa[1] = 0.0; }
    __syncthreads();
    const LdsRows pset{ poly, Kp };
    int n_a = 1, nodes = 0, levels = 0, front_max = 1, status = OBTG_MD_OK;
    double alpha = INFINITY, bt = -1, bx = 0, by = 0, bz = 0;
    while (n_a > 0) {
        if (lane == 0) s_count = 0;
        __syncthreads();
        for (int base = 0; base < n_a; base += kWave) {
            const bool valid = base + lane < n_a;
            const int ni = valid ? base + lane : base;
            const double t = fa[2 * ni];
            const int lev = (int)fa[2 * ni + 1];
            const double w = __builtin_ldexp(1.0, -lev);
            if constexpr (KC > 0) {
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    double x[KC];
#pragma unroll
                    for (int i = 0; i < KC; ++i) x[i] = orig[q * KC + i];
                    subcurve_row_reg<KC>(x, t, w);
#pragma unroll
                    for (int i = 0; i < KC; ++i) work[q * KC + i] = x[i];
                }
            } else {
                for (int q = 0; q < 3; ++q) {
                    for (int i = 0; i < K; ++i) work[q * K + i] = orig[q * K + i];
                    subcurve_row(work + q * K, K, t, w);
                }
            }
            // upper bounds: the sub-curve's end points against the polygon's hull
            double best = INFINITY, bt_l = -1;
            tgjk::P3 bc{ 0, 0, 0 };
            for (int e = 0; e < 2; ++e) {
                const int ie = e ? K - 1 : 0;
                const OnePoint one{ tgjk::P3{ work[ie], work[K + ie], work[2 * K + ie] } };
                const SupportScan<OnePoint, LdsRows> sup{ one, pset, 1, Kp };
                const tgjk::Result r = tgjk::true_distance(sup, one, pset, 1e-13, abs_tol, 64);
                if (r.dist < best) { best = r.dist; bt_l = t + e * w; bc = r.c2; }
            }
            if (!valid) best = INFINITY;
            double wb = best;
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) wb = fmin(wb, __shfl_xor(wb, m));
            if (wb < alpha) {
                const unsigned long long who = __ballot(best == wb);
                const int src = __ffsll((long long)who) - 1;
                alpha = wb; bt = __shfl(bt_l, src);
                bx = __shfl(bc.x, src); by = __shfl(bc.y, src); bz = __shfl(bc.z, src);
            }
            // lower bound: control hull of the sub-curve against the polygon's hull
            const LdsRows cset{ work, K };
            const SupportScan<LdsRows, LdsRows> sup2{ cset, pset, K, Kp };
            const tgjk::Result rl = tgjk::true_distance(sup2, cset, pset, 1e-6, abs_tol, 64);
            const double lb = rl.flag ? rl.lower * (1.0 - 1e-12) : 0.0;
            const bool keep = valid && lb < alpha * (1 - p.eps) && alpha > abs_tol;
            if (keep) {
                if (lev + 1 > p.max_level) status = OBTG_MD_DEPTH_CAP;
                else {
                    const int pos = atomicAdd(&s_count, 2);
                    if (pos + 2 <= p.cap) {
                        const double h = 0.5 * w;
                        fb[2 * pos] = t; fb[2 * pos + 1] = (double)(lev + 1);
                        fb[2 * pos + 2] = t + h; fb[2 * pos + 3] = (double)(lev + 1);
                    } else status = OBTG_MD_NODE_CAP;
                }
            }
        }
        nodes += n_a;
        levels++;
        __syncthreads();
        const unsigned long long capped = __ballot(status != OBTG_MD_OK);
        if (capped) { status = __shfl(status, __ffsll((long long)capped) - 1); break; }
        n_a = min(s_count, p.cap);
        if (n_a > front_max) front_max = n_a;
        if (nodes + n_a > p.max_nodes && n_a > 0) { status = OBTG_MD_NODE_CAP; break; }
        double* tsw = fa; fa = fb; fb = tsw;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __syncthreads();
    }
    if (lane == 0) {
        double* o = p.res + 5 * (size_t)k;
        o[0] = alpha; o[1] = bt; o[2] = bx; o[3] = by; o[4] = bz;
        if (p.info) { p.info[4 * k] = nodes; p.info[4 * k + 1] = levels; p.info[4 * k + 2] = front_max; p.info[4 * k + 3] = status; }
    }
}

// frame layout for the polygon form: c1[3K] then scalars
enum { G_T1 = 0, G_T1L, G_T1H, G_ALPHA, G_RT1, G_PX, G_PY, G_PZ, G_STATE, G_NSCAL };
// a child's record / the blob of a frame in k_min_dist2poly_quad (further down)
enum { P_CAP = 0, P_POS, P_LB, P_T1, P_UB, P_AM, P_CX, P_CY, P_CZ, P_NREC };
__host__ __device__ constexpr int md2_quad_blob(int K) { return 6 * K + 2 * P_NREC; }
__host__ __device__ constexpr int md2_quad_frame(int K) { return md2_quad_blob(K) + G_NSCAL; }   // a frame in the global stack: blob, then (below level kMdScsLds) its scalars

struct Md2Params {
    const double* __restrict__ curves;   // [n_curves][3][K]
    const double* __restrict__ soa;      // polygons, SoA
    const int* __restrict__ off;
    const int* __restrict__ pc;
    const int* __restrict__ pp;
    int n_pairs, K, max_iter, md_cap, max_depth, max_nodes;
    double eps, eps3;                     // eps3 = eps**3 as Python forms it (bezier.py:1468: libm pow, on the host)
    double* stack;
    double* __restrict__ res;             // [n_pairs][5]
    int* __restrict__ info;
    int quad_frame;                       // k_min_dist2poly_quad: doubles per frame of `stack` (what min_dist2poly_stack_doubles sized it by)
};

constexpr int kPolyBias = 1 << 24;
struct MemTwo {   // indices below kPolyBias address the stack frame, the rest the polygon table
    const double* f;
    const double* __restrict__ g;
    __device__ __forceinline__ double operator()(int idx) const
    {
        return idx < kPolyBias ? f[idx] : g[idx - kPolyBias];
    }
};

__global__ __launch_bounds__(64) void k_min_dist2poly(const Md2Params p)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= p.n_pairs) return;
    const int K = p.K, FR = 3 * K + G_NSCAL;
    double* st = p.stack + (size_t)k * p.max_depth * FR;
    const double* ca = p.curves + (size_t)p.pc[k] * 3 * K;
    const int po = p.off[p.pp[k]], PK = p.off[p.pp[k] + 1] - po;
    for (int i = 0; i < 3 * K; ++i) st[i] = ca[i];
    {
        double* sc = st + 3 * K;
        sc[G_T1L] = 0; sc[G_T1H] = 1; sc[G_ALPHA] = INFINITY; sc[G_STATE] = 0;
    }
    int depth = 0, nodes = 0, calls = 0, dmax = 0, status = OBTG_MD_OK;
    double r0 = INFINITY, r1 = -1, rx = -1, ry = -1, rz = -1;
    bool returning = false;
    for (;;) {
        double* f = st + (size_t)depth * FR;
        double* sc = f + 3 * K;
        int state = (int)sc[G_STATE];
        if (!returning && state == 0) {
            if (depth + 1 > 1000) { r0 = r1 = rx = -1; ry = rz = -1; returning = true; depth--; if (depth < 0) break; continue; }
            if (nodes >= p.max_nodes) { status = OBTG_MD_NODE_CAP; break; }
            nodes++;
            if (depth + 1 > dmax) dmax = depth + 1;
            Ctx<MemTwo> g;
            g.mem = MemTwo{ f, p.soa };
            g.P1 = Poly{ 0, K, K, 1 };
            g.P2 = Poly{ kPolyBias + 3 * po, PK, PK, 1 };
            g.trace = nullptr; g.trace_cap = 0; g.n_support = 0;
            Result gr;
            gjk::run(g, p.max_iter, p.md_cap, gr);
            calls++;
            if (gr.status == OBTG_ST_MD_CAP || gr.status == OBTG_ST_CYCLE) { status = OBTG_MD_GJK_CAP; break; }
            double lb, t1, nT1, alpha = sc[G_ALPHA];
            double cx, cy, cz;
            if (gr.flag > 0) {
                lb = gr.dist;
                t1 = hull_param(f, K, gr.c1);
                cx = gr.c2.x; cy = gr.c2.y; cz = gr.c2.z;
                // _upperboundPoly (bezier.py:1535-1547)
                const double d0 = norm_seq(f[0], f[K], f[2 * K], cx, cy, cz);
                const double d1 = norm_seq(f[K - 1], f[2 * K - 1], f[3 * K - 1], cx, cy, cz);
                int am = (d1 < d0) ? 1 : 0;
                if (d0 != d0) am = 0; else if (d1 != d1) am = 1;
                const double ub = am ? d1 : d0, t1loc = am ? 1.0 : 0.0;
                if (ub <= alpha) { alpha = ub; nT1 = (1 - t1loc) * sc[G_T1L] + t1loc * sc[G_T1H]; }
                else nT1 = -1;
            } else {
                t1 = 0.5; nT1 = -1; cx = cy = cz = -1; lb = p.eps3;
            }
            if (lb >= alpha * (1 - p.eps)) {
                r0 = alpha; r1 = nT1; rx = cx; ry = cy; rz = cz; returning = true; depth--;
                if (depth < 0) break;
                continue;
            }
            if (depth + 1 >= p.max_depth) { status = OBTG_MD_DEPTH_CAP; r0 = alpha; r1 = nT1; rx = cx; ry = cy; rz = cz; break; }
            if (t1 != t1) t1 = 0;
            sc[G_T1] = t1; sc[G_ALPHA] = alpha; sc[G_RT1] = nT1; sc[G_PX] = cx; sc[G_PY] = cy; sc[G_PZ] = cz;
            state = 1; sc[G_STATE] = 1;
        }
        if (returning) {
            if (r0 < sc[G_ALPHA]) { sc[G_ALPHA] = r0; sc[G_RT1] = r1; sc[G_PX] = rx; sc[G_PY] = ry; sc[G_PZ] = rz; }
            returning = false;
            state = (int)sc[G_STATE];
        }
        if (state >= 3) {
            r0 = sc[G_ALPHA]; r1 = sc[G_RT1]; rx = sc[G_PX]; ry = sc[G_PY]; rz = sc[G_PZ];
            returning = true; depth--;
            if (depth < 0) break;
            continue;
        }
        {
            const int h1 = state - 1;
            const double t1 = sc[G_T1];
            double* nf = f + FR;
            for (int c = 0; c < 3; ++c) split_row(f + c * K, K, t1, h1, nf + c * K);
            double* ns = nf + 3 * K;
            const double t1len = sc[G_T1H] - sc[G_T1L];
            const double m1 = sc[G_T1L] + t1 * t1len;
            ns[G_T1L] = h1 ? m1 : sc[G_T1L]; ns[G_T1H] = h1 ? sc[G_T1H] : m1;
            ns[G_ALPHA] = sc[G_ALPHA]; ns[G_STATE] = 0;
            sc[G_STATE] = state + 1;
            depth++;
        }
    }
    p.res[5 * k] = r0; p.res[5 * k + 1] = r1; p.res[5 * k + 2] = rx; p.res[5 * k + 3] = ry; p.res[5 * k + 4] = rz;
    if (p.info) { p.info[4 * k] = nodes; p.info[4 * k + 1] = calls; p.info[4 * k + 2] = dmax; p.info[4 * k + 3] = status; }
}

// =====================================================================================
//  launchers
// =====================================================================================
int launch_gjk_pairs(obtg_ctx* c, const double* d_soa, const int* d_off, const int* d_pa,
                     const int* d_pb, int n_pairs, int max_iter, int md_cap, int* d_flag,
                     double* d_p1, double* d_p2, double* d_dist, short* d_trace, int trace_cap,
                     int* d_nsup, int* d_status, bool planar)
{
    if (n_pairs <= 0) return OBTG_OK;
    int chunk = 1024;
    while (chunk > 256 && (n_pairs + chunk - 1) / chunk < 1024) chunk >>= 1;
    GjkPairsParams p{ d_soa, d_off, d_pa, d_pb, n_pairs, max_iter, md_cap, chunk, d_flag, d_p1, d_p2, d_dist,
                      d_trace, trace_cap, d_nsup, d_status };
    const dim3 grid((n_pairs + chunk - 1) / chunk);
    ScopedKernelTimer t(c, OBTG_K_GJK);
    if (planar) hipLaunchKernelGGL(k_gjk_pairs<true>, grid, dim3(256), 0, c->stream, p);
    else hipLaunchKernelGGL(k_gjk_pairs<false>, grid, dim3(256), 0, c->stream, p);
    OBTG_HIP(c, hipGetLastError());
    return OBTG_OK;
}

// Tile-major chunking of the hull pair list for rows that do not fit LDS (MODE 2 above).
// Pairs are bucketed by (a / 8, b / 64); a chunk is one bucket (<= 512 pairs, <= 72 objects when the
// list is the usual all-pairs sweep; arbitrary lists are handled by cutting buckets at kMaxPairs /
// kMaxObjs).  Cached in the context until the pair list or the polygons change.
// Tiles are TA x 64 blocks of the pair matrix.  A workgroup's lanes refill from its tile only, so the taller the tile
// the fewer lanes idle at its end -- as long as the four workgroups a CU can hold (register bound of the tiled
// kernels) still fit its 160 KB of LDS.  C4 (16 points): TA = 8 / 12 / 13 / 14 -> 17.96 / 14.25 / 13.80 /
// 16.11 ms per sweep (14 costs a workgroup per CU).
static int tile_height(int nc)
{
    const int vpq = nc | 1, occ = 4;
    const size_t budget = (size_t)160 * 1024 / occ - 1280;          // 1040 bytes of static LDS per workgroup
    int ta = 8;
    while (ta < 32 && planar_lds_bytes<2>(ta + 1 + 64, vpq, (ta + 1) * 64) <= budget) ++ta;
    if (const char* e = getenv("OBTG_TILE_A")) ta = std::max(4, std::min(32, atoi(e)));     // (experiments)
    return ta;
}

static int build_tiles(obtg_ctx* c, int /*vp*/)
{
    if (c->tile_valid) return OBTG_OK;
    const int TA = tile_height(c->deg + 1), TB = 64, kMaxPairs = TA * TB, kMaxObjs = TA + TB;
    const int np = c->n_hull_pairs, nobj = c->n_veh + c->n_poly;
    if (nobj > 65535) return OBTG_ERR_UNSUPPORTED;
    const std::vector<int>& pa = c->h_hp_a;
    const std::vector<int>& pb = c->h_hp_b;
    if ((int)pa.size() != np) return OBTG_ERR_UNSUPPORTED;
    const int nbb = (nobj + TB - 1) / TB;
    std::vector<int> order(np);
    for (int i = 0; i < np; ++i) order[i] = i;
    auto key = [&](int i) { return (long)(pa[i] / TA) * nbb + pb[i] / TB; };
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return key(x) < key(y); });
    std::vector<int> chunk_off{ 0 }, cobj_off{ 0 }, cobjs, chunk_ij;
    std::vector<unsigned> pslots(np);
    std::vector<int> slot_of(nobj, -1), touched;
    int max_objs = 0, max_pairs = 0, start = 0;
    long last_key = -1;
    bool tile_split = false;        // a tile cut into several chunks (duplicate pairs): no chunk holds all of its vehicles
    auto close_chunk = [&](int end) {
        // the first chunk of a tile writes the tile's temporal-separation rows in the one-launch pair sweep
        const int i0 = order[start];
        const long k0 = key(i0);
        const int ti0 = pa[i0] / TA * TA, tj0 = pb[i0] / TB * TB;
        if (k0 == last_key) tile_split = true;
        const bool first = k0 != last_key && ti0 < c->n_veh - 1 && tj0 < c->n_veh;
        chunk_ij.push_back(first ? ti0 : -1);
        chunk_ij.push_back(tj0);
        last_key = k0;
        for (int o : touched) slot_of[o] = -1;
        max_objs = std::max(max_objs, (int)touched.size());
        max_pairs = std::max(max_pairs, end - start);
        cobjs.insert(cobjs.end(), touched.begin(), touched.end());
        cobj_off.push_back((int)cobjs.size());
        chunk_off.push_back(end);
        touched.clear();
        start = end;
    };
    for (int s2 = 0; s2 < np; ++s2) {
        const int i = order[s2];
        int need = (slot_of[pa[i]] < 0) + (slot_of[pb[i]] < 0 && pb[i] != pa[i]);
        const bool new_bucket = s2 > start && key(i) != key(order[s2 - 1]);
        if (s2 > start && (new_bucket || s2 - start >= kMaxPairs || (int)touched.size() + need > kMaxObjs))
            close_chunk(s2);
        for (int o : { pa[i], pb[i] })
            if (slot_of[o] < 0) { slot_of[o] = (int)touched.size(); touched.push_back(o); }
        pslots[s2] = (unsigned)slot_of[pa[i]] | ((unsigned)slot_of[pb[i]] << 16);
    }
    if (np > start) close_chunk(np);
    auto up = [&](DevBuf& d, const void* src, size_t bytes) -> int {
        int rc = d.reserve(bytes);
        if (rc) return rc;
        if (hipMemcpyAsync(d.p, src, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) return OBTG_ERR_DEVICE;
        return OBTG_OK;
    };
    int rc;
    if ((rc = up(c->d_tile_chunk_off, chunk_off.data(), chunk_off.size() * sizeof(int)))) return rc;
    if ((rc = up(c->d_tile_order, order.data(), order.size() * sizeof(int)))) return rc;
    if ((rc = up(c->d_tile_pslots, pslots.data(), pslots.size() * sizeof(unsigned)))) return rc;
    if ((rc = up(c->d_tile_cobj_off, cobj_off.data(), cobj_off.size() * sizeof(int)))) return rc;
    if ((rc = up(c->d_tile_cobjs, cobjs.data(), cobjs.size() * sizeof(int)))) return rc;
    if ((rc = up(c->d_tile_ij, chunk_ij.data(), chunk_ij.size() * sizeof(int)))) return rc;
    {   // does the list hold every vehicle pair (i < j)?  Then every tile's vehicles are staged by the tile's chunk.
        const long nv = c->n_veh, want = nv * (nv - 1) / 2;
        std::vector<char> seen((size_t)std::max(want, 1L), 0);
        long have = 0;
        for (int i = 0; i < np; ++i) {
            const long a = pa[i], bb = pb[i];
            if (a < bb && bb < nv) {
                const long idx = a * nv - a * (a + 1) / 2 - a - 1 + bb;
                if (!seen[(size_t)idx]) { seen[(size_t)idx] = 1; ++have; }
            }
        }
        c->tile_ts_ok = want > 0 && have == want && TA <= 32 && !tile_split;
    }
    c->tile_a = TA;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return OBTG_ERR_DEVICE;   // sources are locals
    c->tile_n_chunks = (int)chunk_off.size() - 1;
    c->tile_max_objs = max_objs; c->tile_max_pairs = max_pairs;
    c->tile_valid = true;
    return OBTG_OK;
}

// Shape of a MODE 0 sweep's grid: workgroups per row, pairs per chunk (and chunks a workgroup takes one after the other:
// `passes`, 1 unless forced).  Lanes refill from their workgroup's chunk, so a chunk must hold several pairs per lane
// (in list order 316 pairs per 256 lanes ran 0.281 ms where 1264 pairs ran 0.215 ms); with the history order ~860-pair
// chunks do as well as 1264 for the plain sweep, while the one-launch pair sweep (a separation share and two stagings
// fewer per row) prefers 1280.  Small batches trade chunk size for enough workgroups to fill the chip; a row whose
// objects leave less LDS than the target chunk needs gets more, smaller chunks.
// Round 3 tried choosing (workgroups, passes) by "rounds of the chip x duration of a workgroup" -- the workgroup timeline
// shows a launch as rounds of ~57 us workgroups -- and measured it wrong: C3 as ONE workgroup per row in two passes (one
// round at five workgroups per CU) is level with two per row (three rounds at four), and C5's plain sweep as one workgroup
// per row in five passes runs 0.269 ms against 0.240 - 0.244 ms as six or seven per row (OBTG_SWEEP_WGS scan,
// profiles/r03_experiments): when every workgroup starts at once they all stage, sort, run gjkNew and store at the
// same time, while many staggered workgroups keep the CUs' units mixed.  The launches are bound by the chip's aggregate
// issue rate, not by rounds.  OBTG_SWEEP_WGS / OBTG_SWEEP_PASSES force a shape for experiments.
// OBTG_TIMELINE=<file>: the 20th (OBTG_TIMELINE_AT) instrumented launch of the process leaves its workgroup timeline
// there (text: block, start and end in 10 ns ticks from the first start, HW_ID, XCC_ID; tools/timeline_report.py)
struct TimelineDump {
    obtg_ctx* c;
    unsigned grid;
    unsigned long long* dev = nullptr;
    int rc = OBTG_OK;
    TimelineDump(obtg_ctx* c_, unsigned grid_, unsigned long long*& slot) : c(c_), grid(grid_)
    {
        static const char* path = getenv("OBTG_TIMELINE");
        static const int at = getenv("OBTG_TIMELINE_AT") ? atoi(getenv("OBTG_TIMELINE_AT")) : 20;
        static int count = 0;
        if (!(path && path[0] && ++count == at)) return;
        if ((rc = c->ws_misc[6].reserve((size_t)grid * 4 * sizeof(unsigned long long)))) return;
        if (hipMemsetAsync(c->ws_misc[6].p, 0, (size_t)grid * 4 * sizeof(unsigned long long), c->stream) != hipSuccess) {
            rc = OBTG_ERR_DEVICE; return;
        }
        dev = c->ws_misc[6].as<unsigned long long>();
        slot = dev;
    }
    int finish(const char* header)
    {
        if (!dev) return OBTG_OK;
        std::vector<unsigned long long> h((size_t)grid * 4);
        OBTG_HIP(c, hipMemcpyAsync(h.data(), dev, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
        OBTG_HIP(c, hipStreamSynchronize(c->stream));
        if (FILE* f = fopen(getenv("OBTG_TIMELINE"), "w")) {
            unsigned long long t0 = ~0ull;
            for (unsigned i = 0; i < grid; ++i) if (h[4 * i] && h[4 * i] < t0) t0 = h[4 * i];
            fprintf(f, "%s\n", header);
            for (unsigned i = 0; i < grid; ++i)
                fprintf(f, "%u %llu %llu %llu %llu\n", i, h[4 * i] ? h[4 * i] - t0 : 0ull, h[4 * i + 1] ? h[4 * i + 1] - t0 : 0ull,
                        h[4 * i + 2] & 0xffffffffull, h[4 * i + 2] >> 32);
            fclose(f);
        }
        return OBTG_OK;
    }
};

// launch with the timer's events riding on the dispatch itself (kernel start / stop timestamps), or plainly
template <class K, class P>
static void launch_timed(ScopedKernelTimer& t, K kern, dim3 grid, dim3 block, size_t lds, hipStream_t stream, const P& params)
{
    if (t.ext && t.a && t.b) hipExtLaunchKernelGGL(kern, grid, block, (unsigned)lds, stream, t.a, t.b, 0, params);
    else hipLaunchKernelGGL(kern, grid, block, lds, stream, params);
}

// (experiment, profiles/r06_experiments: coarser bins keep more of a tile row's pairs next to each other in the sorted order --
// lanes of a wave then share object A in LDS -- at the price of mixing trip counts within a wave)
static int sweep_hist_shift()
{
    static const int v = getenv("OBTG_HIST_SHIFT") ? std::max(0, std::min(7, atoi(getenv("OBTG_HIST_SHIFT")))) : 0;
    return v;
}

static int sweep_refill_min()
{
    static const int v = getenv("OBTG_REFILL_MIN") ? std::max(1, std::min(64, atoi(getenv("OBTG_REFILL_MIN")))) : 32;
    return v;
}

struct SweepShape {
    int wgs = 1, passes = 1, chunk = 0, per_cu = 1;
    size_t lds = 0;
};
static SweepShape sweep_shape(const obtg_ctx* c, int B, int nc, int waves_per_simd, int chunk_target)
{
    const int np = c->n_hull_pairs, n_obj = c->n_veh + c->n_poly + c->n_obs, vpq = nc | 1;   // (point obstacles: staged by the pair sweep)
    const size_t fixed = planar_lds_bytes<0>(n_obj, vpq, 0);
    SweepShape sh;
    // workgroups per CU: what the kernel's registers allow, fewer while the row's objects leave no room for 256 pairs
    int per_cu = nc <= 11 ? waves_per_simd : 1;
    size_t budget = 0;
    for (; per_cu >= 1; --per_cu) {
        budget = (size_t)160 * 1024 / per_cu - 1280;
        if (per_cu == 1) budget = 64 * 1024;                       // one workgroup per CU: the launch limit
        if (fixed + 14 * 256 <= budget) break;
    }
    if (per_cu < 1) per_cu = 1;
    const int chunk_max = fixed + 14 * 256 <= budget ? (int)std::min<size_t>((budget - fixed) / 14, 65535) : 256;
    int W = std::max(1, (np + chunk_target - 1) / chunk_target);
    // Small batches (a rank's share of a row-sharded iteration, bench.py --mode rows): as many workgroups per row as ONE
    // round of resident workgroups has room for beside the row's dynamics group -- a second round costs a whole
    // workgroup's latency again (B = 145 at C3: 6 per row 0.040 ms, 8 per row 0.052, the old rule's 16 per row 0.068;
    // B = 289: 3 per row 0.061, 8 per row 0.080) -- and three per row while the launch is two or three rounds
    // (B = 577: 0.104 against 0.109 / 0.116 for 4 / 2); tools/r04_small_batch_scan.sh, profiles/r04_experiments/.
    {
        const long slots = (long)per_cu * std::max(1, c->n_cus);
        if ((long)B * (W + 1) < slots) {
            const int fill = (int)((slots - B + B / 2) / std::max(1, B));         // round((slots - B) / B)
            W = std::max(W, std::min(fill, std::max(1, np / 128)));
        } else if ((long)B * (W + 1) < 3 * slots) W = std::max(W, std::min(3, std::max(1, np / 128)));
    }
    while ((np + W - 1) / W > chunk_max && (np + W - 1) / W > 256) ++W;           // the row's objects leave less LDS
    int Q = 1;
    if (const char* e = getenv("OBTG_SWEEP_WGS")) W = std::max(1, atoi(e));
    if (const char* e = getenv("OBTG_SWEEP_PASSES")) Q = std::max(1, atoi(e));
    const int per_wg = (np + W - 1) / W;
    while ((per_wg + Q - 1) / Q > chunk_max && (per_wg + Q - 1) / Q > 64) ++Q;   // (a forced shape may need passes to fit)
    sh.passes = Q;
    sh.chunk = (per_wg + Q - 1) / Q;
    sh.wgs = (np + sh.passes * sh.chunk - 1) / (sh.passes * sh.chunk);
    sh.per_cu = per_cu;
    sh.lds = planar_lds_bytes<0>(n_obj, vpq, sh.chunk);
    return sh;
}

int launch_gjk_swarm(obtg_ctx* c, const double* dY, int B, int max_iter, int md_cap, int* d_flag,
                     double* d_p1, double* d_p2, double* d_dist, int* d_nsup, int* d_status, SweepFold* fold)
{
    if (B <= 0 || c->n_hull_pairs <= 0) return OBTG_OK;
    GjkSwarmParams p{};
    p.Y = dY; p.poly = c->d_poly_pts.as<double>(); p.poly_off = c->d_poly_off.as<int>();
    p.pa = c->d_hp_a.as<int>(); p.pb = c->d_hp_b.as<int>();
    p.n_veh = c->n_veh; p.dim = c->dim; p.nc = c->deg + 1; p.n_poly = c->n_poly;
    p.n_poly_pts = c->n_poly_pts; p.n_pairs = c->n_hull_pairs;
    const int vlen = c->dim * (c->deg + 1);
    p.vp = (vlen % 2 == 0) ? vlen + 1 : vlen;
    // workgroups per row.  Lanes refill from their workgroup's chunk, so a chunk must hold several
    // pairs per lane (measured at C3 in list order: 316 pairs per 256 lanes = 0.281 ms, 1264 pairs =
    // 0.215 ms); with the history order 864-pair chunks do as well as 1264 and leave LDS for a fifth
    // workgroup per CU.  Small batches trade chunk size for enough workgroups to fill the chip.
    const int wgs = std::max(1, (c->n_hull_pairs + kSweepChunk - 1) / kSweepChunk);     // the general kernels' chunking
    p.chunk = (c->n_hull_pairs + wgs - 1) / wgs;
    p.wgs_per_row = (c->n_hull_pairs + p.chunk - 1) / p.chunk;
    p.max_iter = max_iter; p.md_cap = md_cap;
    p.refill_min = sweep_refill_min(); p.hist_shift = sweep_hist_shift();
    p.flag = d_flag; p.p1 = d_p1; p.p2 = d_p2; p.dist = d_dist; p.nsup = d_nsup; p.status = d_status;
    if (c->fd.Y0) {
        if (c->fd_dedup) return kNeedBatch;      // the de-duplication mask compares rows in memory
        p.Y = c->fd.Y0; p.fd = 1 + c->fd.row0; p.fd_fixed = c->fd.fixed; p.fd_h = c->fd.h;
    }
    size_t lds = sizeof(double) * ((size_t)c->n_veh * p.vp + 3 * (size_t)c->n_poly_pts);
    const bool planar = c->dim == 2 && c->polys_planar;
    if (planar && c->max_poly_K <= c->deg + 1 && c->deg + 1 <= 127) {
        const int nc = c->deg + 1;
        const int vp2 = nc | 1;                    // object pitch in 16-byte points (PlanarShape<NC>::VPQ)
        const SweepShape shape = sweep_shape(c, B, nc, kSweepWavesPerSimd, kSweepChunk);
        p.chunk = shape.chunk; p.wgs_per_row = shape.wgs; p.passes = shape.passes;
        const size_t lds2 = shape.lds;
        void (*kp)(const GjkSwarmParams) = nullptr;
        void (*kf)(const GjkSwarmParams) = nullptr;
        void (*kt)(const GjkSwarmParams) = nullptr;
#define OBTG_GJK_CASE(NC_) \
    case NC_: kp = k_gjk_swarm_planar<NC_, 0>; kf = k_gjk_swarm_planar<NC_, 1>; kt = k_gjk_swarm_planar<NC_, 2>; break;
        switch (nc) {
            OBTG_NC_SEP(OBTG_GJK_CASE)
            default: break;
        }
#undef OBTG_GJK_CASE
        constexpr size_t kTileAbove = 48 * 1024, kSweepMax = 64 * 1024;
        if (kt && lds2 > kTileAbove) {
            // large rows: tile-major chunks, each staging only the objects it touches
            int rc = build_tiles(c, vp2);
            if (rc == OBTG_OK) {
                GjkSwarmParams q = p;
                q.chunk_off = c->d_tile_chunk_off.as<int>(); q.order = c->d_tile_order.as<int>();
                q.pslots = c->d_tile_pslots.as<unsigned>(); q.cobj_off = c->d_tile_cobj_off.as<int>();
                q.cobjs = c->d_tile_cobjs.as<int>(); q.max_objs = c->tile_max_objs;
                q.chunk = c->tile_max_pairs; q.wgs_per_row = c->tile_n_chunks; q.passes = 1;
                const size_t ldst = planar_lds_bytes<2>(q.max_objs, vp2, q.chunk);
                if (ldst <= 64 * 1024) {
                    const size_t npairs = (size_t)c->n_hull_pairs;
                    obtg::DevBuf& hist_out = c->d_gjk_len[c->gjk_len_cur ^ 1];
                    if (c->gjk_history) { if (int rc2 = hist_out.reserve((size_t)B * npairs)) return rc2; }
                    q.B = B;
                    q.len_in = (c->gjk_history && c->gjk_len_rows > 0) ? c->d_gjk_len[c->gjk_len_cur].as<unsigned char>() : nullptr;
                    q.len_in_stride = c->gjk_len_rows == B ? (int)npairs : 0;
                    q.len_out = c->gjk_history ? hist_out.as<unsigned char>() : nullptr;
                    if (c->gjk_history) { c->gjk_len_cur ^= 1; c->gjk_len_rows = B; }
                    if (ldst > 48 * 1024)
                        OBTG_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kt),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldst));
                    ScopedKernelTimer t(c, OBTG_K_GJK);
                    hipLaunchKernelGGL(kt, dim3((unsigned)((size_t)B * q.wgs_per_row)), dim3(256), ldst, c->stream, q);
                    OBTG_HIP(c, hipGetLastError());
                    return OBTG_OK;
                }
            } else if (rc != OBTG_ERR_UNSUPPORTED) return rc;
        }
        if (kp && lds2 <= kSweepMax) {   // larger rows: the general kernel does better than 1-2 workgroups per CU
            if (lds2 > 48 * 1024)
                OBTG_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kp),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
            ScopedKernelTimer t(c, OBTG_K_GJK);
            if (c->fd_dedup && B > 1 && kf) {
                // Finite-difference de-duplication: row 0 in full, its results broadcast to every
                // row, then one workgroup per row re-evaluates the pairs whose hulls differ from row 0.
                int rc = c->ws_misc[7].reserve((size_t)B * c->n_veh);
                if (rc) return rc;
                unsigned char* chg = c->ws_misc[7].as<unsigned char>();
                const int tot = B * c->n_veh;
                hipLaunchKernelGGL(k_changed_objects, dim3((tot + 255) / 256), dim3(256), 0, c->stream, dY, B,
                                   c->n_veh, c->dim * (c->deg + 1), chg);
                {   // row 0 alone: chunk it as a one-row batch (many small workgroups)
                    GjkSwarmParams r0 = p;
                    int w0 = (c->n_hull_pairs + 1279) / 1280;
                    while (w0 < 2048 && (c->n_hull_pairs + w0 - 1) / w0 > 256) w0 <<= 1;
                    r0.chunk = (c->n_hull_pairs + w0 - 1) / w0;
                    r0.wgs_per_row = (c->n_hull_pairs + r0.chunk - 1) / r0.chunk;
                    r0.B = 1; r0.passes = 1;
                    hipLaunchKernelGGL(kp, dim3((unsigned)(8 * r0.wgs_per_row)), dim3(OBTG_SWEEP_THREADS),
                                       planar_lds_bytes<0>(c->n_veh + c->n_poly, vp2, r0.chunk), c->stream, r0);
                }
                const size_t np = (size_t)c->n_hull_pairs;
                const dim3 cb(256);
                const unsigned gy = (unsigned)((B - 1 + 15) / 16);
                const dim3 g1((unsigned)((np + 255) / 256), gy), g3((unsigned)((3 * np + 255) / 256), gy);
                hipLaunchKernelGGL(k_bcast_row0<int>, g1, cb, 0, c->stream, d_flag, np, B);
                hipLaunchKernelGGL(k_bcast_row0<double>, g3, cb, 0, c->stream, d_p1, 3 * np, B);
                hipLaunchKernelGGL(k_bcast_row0<double>, g3, cb, 0, c->stream, d_p2, 3 * np, B);
                hipLaunchKernelGGL(k_bcast_row0<double>, g1, cb, 0, c->stream, d_dist, np, B);
                if (d_nsup) hipLaunchKernelGGL(k_bcast_row0<int>, g1, cb, 0, c->stream, d_nsup, np, B);
                if (d_status) hipLaunchKernelGGL(k_bcast_row0<int>, g1, cb, 0, c->stream, d_status, np, B);
                GjkSwarmParams q = p;
                q.chg = chg;
                q.chunk = 1024; q.passes = 1;
                const size_t ldsf = planar_lds_bytes<1>(c->n_veh + c->n_poly, vp2, q.chunk);
                hipLaunchKernelGGL(kf, dim3((unsigned)(B - 1)), dim3(256), ldsf, c->stream, q);
            } else {
                // trip-count history: this sweep orders its pairs by the counts the previous one left
                // behind (same batch shape: row by row; otherwise every row follows the old row 0)
                const size_t np = (size_t)c->n_hull_pairs;
                obtg::DevBuf& hist_out = c->d_gjk_len[c->gjk_len_cur ^ 1];
                if (int rc = hist_out.reserve((size_t)B * np)) return rc;
                p.B = B;
                p.len_in = (c->gjk_history && c->gjk_len_rows > 0) ? c->d_gjk_len[c->gjk_len_cur].as<unsigned char>() : nullptr;
                p.len_in_stride = c->gjk_len_rows == B ? (int)np : 0;
                p.len_out = c->gjk_history ? hist_out.as<unsigned char>() : nullptr;
                const unsigned grid = (unsigned)(((size_t)B + 7) / 8 * 8 * p.wgs_per_row);
                hipLaunchKernelGGL(kp, dim3(grid), dim3(OBTG_SWEEP_THREADS), lds2, c->stream, p);
                c->gjk_len_cur ^= 1;
                c->gjk_len_rows = B;
            }
            OBTG_HIP(c, hipGetLastError());
            return OBTG_OK;
        }
    }
    if (!planar && !c->fd_dedup && c->max_poly_K <= c->deg + 1) {
        // 3-D (or non-planar polygons): the fixed-point-count 3-D sweep when it is instantiated and fits
        const int nc = c->deg + 1, n_obj = c->n_veh + c->n_poly;
        void (*k3)(const GjkSwarmParams) = nullptr;
        switch (nc) {
#define OBTG_CASE(NC_) case NC_: k3 = k_gjk_swarm_3d<NC_>; break;
            OBTG_NC_DYN(OBTG_CASE)
#undef OBTG_CASE
            default: break;
        }
        // Chunks: a pair of the 3-D machine can take 50 scans (cycling pairs run until the detector fires), and a
        // workgroup lasts as long as its longest pair whatever its size -- so few large workgroups (one round of
        // the chip) beat many small ones; split rows only for very small batches.
        // ... up to what the chip holds at once (160 VGPRs: three workgroups per CU, 768).  C2_file, 433 rows: two
        // workgroups per row are 866 -- a second round for the last 98, 0.243 ms; one per row is one round, 0.177 ms.
        {
            long resident = 256 * 3;
            if (k3) {
                int per_cu = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(k3), 256,
                                                                 sweep3d_lds_bytes(n_obj, nc, 256)) == hipSuccess && per_cu > 0)
                    resident = (long)per_cu * c->n_cus;
                else (void)hipGetLastError();
            }
            int wgs3 = (c->n_hull_pairs + kSweepChunk - 1) / kSweepChunk;
            while ((long)B * wgs3 * 2 <= resident && (c->n_hull_pairs + wgs3 - 1) / wgs3 > 256) wgs3 <<= 1;
            p.chunk = (c->n_hull_pairs + wgs3 - 1) / wgs3;
            p.wgs_per_row = (c->n_hull_pairs + p.chunk - 1) / p.chunk;
        }
        const size_t lds3 = sweep3d_lds_bytes(n_obj, nc, p.chunk);
        if (k3 && lds3 <= 64 * 1024 && p.chunk <= 65535) {
            if (lds3 > 48 * 1024)
                OBTG_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k3),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
            const size_t np = (size_t)c->n_hull_pairs;
            obtg::DevBuf& hist_out = c->d_gjk_len[c->gjk_len_cur ^ 1];
            if (int rc2 = hist_out.reserve((size_t)B * np)) return rc2;
            p.B = B;
            p.len_in = (c->gjk_history && c->gjk_len_rows > 0) ? c->d_gjk_len[c->gjk_len_cur].as<unsigned char>() : nullptr;
            p.len_in_stride = c->gjk_len_rows == B ? (int)np : 0;
            p.len_out = c->gjk_history ? hist_out.as<unsigned char>() : nullptr;
            const unsigned grid = (unsigned)(((size_t)B + 7) / 8 * 8 * p.wgs_per_row);
            // one launch for the batch: fold the row's temporal-separation block (and speed rows) into the sweep
            bool folded = false;
            if (fold && fold->d_out_sep && c->dim == 3 && c->R == 0 && c->n_obs == 0 && c->n_pairs > 0 && !c->fd_dedup) {
                void (*kf3)(const GjkSwarmParams, const NsParams, const NsParams, int) = nullptr;
                switch (nc) {
#define OBTG_CASE(NC_) case NC_: kf3 = k_pair_sweep_3d<NC_>; break;
                    OBTG_NC_DYN(OBTG_CASE)
#undef OBTG_CASE
                    default: break;
                }
                int rc2 = ensure_tables(c);
                if (rc2) return rc2;
                const int W = p.wgs_per_row, vlen3 = 3 * nc, vp3 = (vlen3 % 2 == 0) ? vlen3 + 1 : vlen3;
                const int L = 2 * c->deg + 1, tpf = (L % 2 == 0) ? L + 1 : L, tr = 16;
                NsParams ts{}, sp{};
                ts.Y = p.Y; ts.obs = nullptr; ts.pairs = c->d_pairs.as<int2>(); ts.W2 = c->d_w2.as<double>();
                ts.Tt = c->d_Tt.as<double>(); ts.Td = c->d_Td.as<double>(); ts.Tf = c->d_Tf.as<double>(); ts.out = fold->d_out_sep;
                ts.n_veh = c->n_veh; ts.n_obj = c->n_obj; ts.R = 0; ts.item_begin = 0; ts.item_count = c->n_pairs;
                const int groups_t = (c->n_pairs + kWave - 1) / kWave;
                ts.groups_per_wg = (groups_t + W - 1) / W; ts.wgs_per_row = W; ts.waves = 4;
                ts.stage_all = 1; ts.stage_slots = c->n_obj; ts.tile_rows = tr; ts.tiling = 0; ts.tiles = nullptr;
                ts.sign = 1.0; ts.offset = -square_as_python(fold->max_sep);
                ts.fd = p.fd; ts.fd_fixed = p.fd_fixed; ts.fd_h = p.fd_h;
                size_t ns_bytes = sizeof(double) * ((size_t)ts.stage_slots * vp3 + (size_t)4 * tr * tpf);
                if (fold->d_out_speed && fold->d_tf && !c->speed2.d_out) {      // (a second speed bound: the dynamics launch writes both)
                    sp = ts;
                    sp.pairs = nullptr; sp.tf = fold->d_tf; sp.out = fold->d_out_speed; sp.n_obj = c->n_veh;
                    sp.item_count = c->n_veh;
                    const int groups_s = (c->n_veh + kWave - 1) / kWave;
                    sp.groups_per_wg = (groups_s + W - 1) / W;
                    sp.stage_all = 0; sp.stage_slots = std::min(c->n_veh, kWave * sp.groups_per_wg);
                    const double b2 = square_as_python(fold->speed_bound);
                    sp.sign = fold->speed_is_max ? -1.0 : 1.0; sp.offset = fold->speed_is_max ? b2 : -b2;
                    ns_bytes = std::max(ns_bytes, sizeof(double) * ((size_t)sp.stage_slots * vp3 + (size_t)4 * tr * tpf));
                }
                const size_t lds_a = (lds3 + 15) / 16 * 16;
                if (kf3 && lds_a + ns_bytes <= 64 * 1024) {
                    if (lds_a + ns_bytes > 48 * 1024)
                        OBTG_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kf3),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_a + ns_bytes)));
                    ScopedKernelTimer t(c, OBTG_K_PAIR_SWEEP);
                    hipLaunchKernelGGL(kf3, dim3(grid), dim3(256), lds_a + ns_bytes, c->stream, p, ts, sp, (int)(lds_a / sizeof(double)));
                    folded = true;
                    fold->did_sep = true;
                    fold->did_speed = sp.out != nullptr;
                }
            }
            if (!folded) {
                ScopedKernelTimer t(c, OBTG_K_GJK);
                hipLaunchKernelGGL(k3, dim3(grid), dim3(256), lds3, c->stream, p);
            }
            c->gjk_len_cur ^= 1;
            c->gjk_len_rows = B;
            OBTG_HIP(c, hipGetLastError());
            return OBTG_OK;
        }
    }
    if (lds > 160 * 1024 - 64) return OBTG_ERR_UNSUPPORTED;      // the general kernel stages the whole row (the tiled sweep above does not)
    p.chunk = (c->n_hull_pairs + wgs - 1) / wgs;                  // (the fixed-count branches above may have reshaped the grid)
    p.wgs_per_row = (c->n_hull_pairs + p.chunk - 1) / p.chunk; p.passes = 1;
    auto kern = planar ? k_gjk_swarm<true> : k_gjk_swarm<false>;
    if (lds > 48 * 1024)
        OBTG_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ScopedKernelTimer t(c, OBTG_K_GJK);
    hipLaunchKernelGGL(kern, dim3((unsigned)((size_t)B * p.wgs_per_row)), dim3(256), lds, c->stream, p);
    OBTG_HIP(c, hipGetLastError());
    return OBTG_OK;
}

// The pair sweeps of a batch as one launch: the planar GJK sweep whose workgroups also write their
// row's temporal-separation block (GjkSwarmParams::ts).  Shapes outside that kernel: two launches.
// does obtg_pair_sweep_dev run as ONE launch for this context (large batch)?  Mirrors launch_pair_sweep's decision.
// Rows per pass of the separation block's transposition tile (4 waves x rows x odd(2n+1) doubles), which borrows the LDS
// behind the staged objects; small rows get a larger allocation so that 16 rows fit.  0: no room.
static int pair_sweep_tile_rows(const obtg_ctx* c, int nc, size_t& lds)
{
    const int n_obj = c->n_veh + c->n_poly + c->n_obs, vpq = nc | 1;
    const int L = 2 * c->deg + 1, tpf = (L % 2 == 0) ? L + 1 : L;
    const size_t objects = (size_t)16 * n_obj * vpq;
    auto tile = [&](int rows) { return (size_t)4 * rows * tpf * sizeof(double); };
    int tr = 64;                       // the largest multiple of 8 that fits
    while (tr >= 8 && objects + tile(tr) > lds) tr -= 8;
    // up to 32 rows per pass (two passes per 64-pair group) while the workgroup keeps its place in the CU's LDS
    const size_t budget = (size_t)160 * 1024 / kPairSweepWavesPerSimd - 1280;
    for (int want = 32; want >= 8 && tr < want; want -= 8)
        if (objects + tile(want) <= budget) { lds = std::max(lds, objects + tile(want)); tr = want; break; }
    return tr >= 8 ? tr : 0;
}

bool pair_sweep_is_one_launch(const obtg_ctx* c)
{
    const int nc = c->deg + 1;
    if (!nc_in_dyn(nc)) return false;
    if (!(c->dim == 2 && c->polys_planar && c->max_poly_K <= nc && c->n_hull_pairs > 0 && !c->fd_dedup && c->R == 0 &&
          c->n_pairs > 0)) return false;
    size_t lds = sweep_shape(c, 1 << 20, nc, kPairSweepWavesPerSimd, kPairSweepChunk).lds;
    return pair_sweep_tile_rows(c, nc, lds) > 0 && lds <= 48 * 1024;
}

int launch_pair_sweep(obtg_ctx* c, const double* dY, int B, double max_sep, double* d_out_sep, int max_iter,
                      int md_cap, int* d_flag, double* d_p1, double* d_p2, double* d_dist, int* d_nsup, int* d_status,
                      SweepFold* speed)
{
    // with a virtual finite-difference batch (c->fd) only the one-launch kernel applies: it forms the rows while
    // staging them; OBTG_ERR_UNSUPPORTED tells the caller to materialise the batch instead
    if (B <= 0) return OBTG_OK;
    const int nc = c->deg + 1;
    void (*kern)(const GjkSwarmParams) = nullptr;
    switch (nc) {
#define OBTG_CASE(NC_) case NC_: kern = k_pair_sweep<NC_>; break;
        OBTG_NC_DYN(OBTG_CASE)
#undef OBTG_CASE
        default: break;
    }
    bool fused = kern && c->dim == 2 && c->polys_planar && c->max_poly_K <= nc && c->n_hull_pairs > 0 &&
                 !c->fd_dedup && c->R == 0 && c->n_pairs > 0;
    GjkSwarmParams p{};
    size_t lds = 0;
    if (fused) {
        int rc = ensure_tables(c);
        if (rc) return rc;
        p.Y = dY; p.poly = c->d_poly_pts.as<double>(); p.poly_off = c->d_poly_off.as<int>();
        p.pa = c->d_hp_a.as<int>(); p.pb = c->d_hp_b.as<int>();
        p.n_veh = c->n_veh; p.dim = c->dim; p.nc = nc; p.n_poly = c->n_poly;
        p.n_poly_pts = c->n_poly_pts; p.n_pairs = c->n_hull_pairs;
        const SweepShape shape = sweep_shape(c, B, nc, kPairSweepWavesPerSimd, kPairSweepChunk);
        p.chunk = shape.chunk; p.wgs_per_row = shape.wgs; p.passes = shape.passes;
        p.max_iter = max_iter; p.md_cap = md_cap;
        p.refill_min = sweep_refill_min(); p.hist_shift = sweep_hist_shift();
        p.flag = d_flag; p.p1 = d_p1; p.p2 = d_p2; p.dist = d_dist; p.nsup = d_nsup; p.status = d_status;
        lds = shape.lds;
        const int tr = pair_sweep_tile_rows(c, nc, lds);      // the transposition tile borrows the LDS behind the objects
        fused = tr > 0 && lds <= 48 * 1024;
        p.ts.pairs = c->d_pairs.as<int2>(); p.ts.W2 = c->d_w2.as<double>(); p.ts.out = d_out_sep;
        p.ts.n_pairs = c->n_pairs; p.ts.sign = 1.0; p.ts.offset = 0.0 - square_as_python(max_sep);
        p.ts_tile_rows = tr;
        if (c->n_obs > 0) {            // point obstacles: constant curves behind the hull objects, for the separation rows only
            p.obs = c->d_obs.as<double>(); p.n_obs = c->n_obs;
            p.ts.n_veh = c->n_veh; p.ts.obs_shift = c->n_poly;
        }
        if (c->fd.Y0) { p.Y = c->fd.Y0; p.fd = 1 + c->fd.row0; p.fd_fixed = c->fd.fixed; p.fd_h = c->fd.h; }
    }
    if (!fused && kern && c->dim == 2 && c->polys_planar && c->max_poly_K <= nc && c->n_hull_pairs > 0 && !c->fd_dedup &&
        c->R == 0 && c->n_obs == 0 && c->n_pairs > 0 && lds > 48 * 1024) {
        // large rows: the tiled sweep, its chunks writing their tile's separation rows
        void (*kt)(const GjkSwarmParams) = nullptr;
        switch (nc) {
#define OBTG_CASE(NC_) case NC_: kt = k_pair_sweep_tiled<NC_>; break;
            OBTG_NC_DYN(OBTG_CASE)
#undef OBTG_CASE
            default: break;
        }
        const int vpq = nc | 1;
        int rc = build_tiles(c, vpq);
        if (rc != OBTG_OK && rc != OBTG_ERR_UNSUPPORTED) return rc;
        if (rc == OBTG_OK && kt && c->tile_ts_ok) {
            GjkSwarmParams q = p;
            q.chunk_off = c->d_tile_chunk_off.as<int>(); q.order = c->d_tile_order.as<int>();
            q.pslots = c->d_tile_pslots.as<unsigned>(); q.cobj_off = c->d_tile_cobj_off.as<int>();
            q.cobjs = c->d_tile_cobjs.as<int>(); q.max_objs = c->tile_max_objs;
            q.chunk_ij = c->d_tile_ij.as<int2>(); q.tile_a = c->tile_a;
            q.chunk = c->tile_max_pairs; q.wgs_per_row = c->tile_n_chunks;
            const size_t ldst = planar_lds_bytes<2>(q.max_objs, vpq, q.chunk);
            const int L = 2 * c->deg + 1, tpf = (L % 2 == 0) ? L + 1 : L;
            const size_t behind = ldst - (size_t)16 * q.max_objs * vpq;
            int tr = 64;
            while (tr >= 8 && (size_t)4 * tr * tpf * sizeof(double) > behind) tr -= 8;
            if (tr >= 8 && ldst <= 64 * 1024) {
                q.ts_tile_rows = tr;
                const size_t npairs = (size_t)c->n_hull_pairs;
                obtg::DevBuf& hist_out = c->d_gjk_len[c->gjk_len_cur ^ 1];
                if (c->gjk_history) { if (int rc2 = hist_out.reserve((size_t)B * npairs)) return rc2; }
                q.B = B;
                q.len_in = (c->gjk_history && c->gjk_len_rows > 0) ? c->d_gjk_len[c->gjk_len_cur].as<unsigned char>() : nullptr;
                q.len_in_stride = c->gjk_len_rows == B ? (int)npairs : 0;
                q.len_out = c->gjk_history ? hist_out.as<unsigned char>() : nullptr;
                if (c->gjk_history) { c->gjk_len_cur ^= 1; c->gjk_len_rows = B; }
                unsigned grid_t = (unsigned)((size_t)B * q.wgs_per_row);
                size_t lds_t = ldst;
                static const bool fold_dyn_t = !(getenv("OBTG_FOLD_DYNAMICS") && getenv("OBTG_FOLD_DYNAMICS")[0] == '0');
                if (fold_dyn_t && speed && speed->d_out_ang && speed->d_out_speed && speed->d_tf && c->d_ang_w22n.p != nullptr) {
                    const int L4 = 4 * c->deg + 1;
                    const size_t lds_dyn = sizeof(double) * ((size_t)kWave * L4 + (size_t)(kWave / 2) * L);
                    if (std::max(ldst, lds_dyn) <= (size_t)160 * 1024 / 4 - 1280) {       // still four workgroups per CU
                        AngParams& d = q.dyn;
                        d.Y = q.Y; d.tf = speed->d_tf; d.out = speed->d_out_ang; d.out_speed = speed->d_out_speed;
                        d.n_veh = c->n_veh; d.total = B * c->n_veh;
                        d.w2 = square_as_python(speed->max_rate);
                        const double b2 = square_as_python(speed->speed_bound);
                        d.sp_sign = speed->speed_is_max ? -1.0 : 1.0; d.sp_offset = speed->speed_is_max ? b2 : -b2;
                        if (c->speed2.d_out) {
                            const double c2 = square_as_python(c->speed2.bound);
                            d.out_speed2 = c->speed2.d_out;
                            d.sp2_sign = c->speed2.is_max ? -1.0 : 1.0; d.sp2_offset = c->speed2.is_max ? c2 : -c2;
                        }
                        d.W2n = c->d_ang_w2n.as<double>(); d.W22n = c->d_ang_w22n.as<double>(); d.Wn = c->d_ang_wn.as<double>();
                        d.fd = q.fd; d.fd_fixed = q.fd_fixed; d.fd_h = q.fd_h;
                        q.dyn_first_block = (int)grid_t;
                        grid_t += (unsigned)((d.total + kWave - 1) / kWave);
                        lds_t = std::max(ldst, lds_dyn);
                        speed->did_dynamics = true;
                    }
                }
                if (lds_t > 48 * 1024)
                    OBTG_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kt),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t));
                ScopedKernelTimer tm(c, OBTG_K_PAIR_SWEEP, true);
                launch_timed(tm, kt, dim3(grid_t), dim3(256), lds_t, c->stream, q);
                OBTG_HIP(c, hipGetLastError());
                return OBTG_OK;
            }
        }
    }
    if (!fused) {
        // the 3-D sweep takes the separation block (and speed rows) into its launch where it can; otherwise two
        // launches, each forming the virtual batch's rows itself or asking for the batch (kNeedBatch)
        SweepFold f;
        if (speed) f = *speed;
        f.max_sep = max_sep; f.d_out_sep = d_out_sep; f.did_sep = f.did_speed = false;
        int rc = launch_gjk_swarm(c, dY, B, max_iter, md_cap, d_flag, d_p1, d_p2, d_dist, d_nsup, d_status, &f);
        if (rc) return rc;
        if (speed) speed->did_speed = f.did_speed;
        if (f.did_sep) return OBTG_OK;
        if (speed && c->R > 0) {       // DEG_ELEV > 0: the separation rows and the dynamics rows share a launch
            rc = launch_sep_dynamics_elev(c, dY, B, max_sep, d_out_sep, *speed);
            if (rc == OBTG_OK) { speed->did_dynamics = true; return OBTG_OK; }
            if (rc != OBTG_ERR_UNSUPPORTED) return rc;
        }
        return launch_temporal_sep(c, dY, B, max_sep, 0, c->n_pairs, false, d_out_sep);
    }
    const size_t np = (size_t)c->n_hull_pairs;
    obtg::DevBuf& hist_out = c->d_gjk_len[c->gjk_len_cur ^ 1];
    if (int rc = hist_out.reserve((size_t)B * np)) return rc;
    p.B = B;
    p.len_in = (c->gjk_history && c->gjk_len_rows > 0) ? c->d_gjk_len[c->gjk_len_cur].as<unsigned char>() : nullptr;
    p.len_in_stride = c->gjk_len_rows == B ? (int)np : 0;
    p.len_out = c->gjk_history ? hist_out.as<unsigned char>() : nullptr;
    unsigned grid = (unsigned)(((size_t)B + 7) / 8 * 8 * p.wgs_per_row);
    // the speed / angular-rate groups of the batch as the grid's last workgroups (they fill the slots the sweep's tail
    // leaves empty): when the caller wants them and their 26 KB of LDS fit the sweep's allocation class
    static const bool fold_dyn = !(getenv("OBTG_FOLD_DYNAMICS") && getenv("OBTG_FOLD_DYNAMICS")[0] == '0');
    if (fold_dyn && speed && speed->d_out_ang && speed->d_out_speed && speed->d_tf && c->d_ang_w22n.p != nullptr) {
        const int L4 = 4 * c->deg + 1, L2 = 2 * c->deg + 1;
        const size_t lds_dyn = sizeof(double) * ((size_t)kWave * L4 + (size_t)(kWave / 2) * L2);
        const size_t budget = (size_t)160 * 1024 / kPairSweepWavesPerSimd - 1280;
        if (nc <= 11 && std::max(lds, lds_dyn) <= budget) {
            AngParams& d = p.dyn;
            d.Y = p.Y; d.tf = speed->d_tf; d.out = speed->d_out_ang; d.out_speed = speed->d_out_speed;
            d.n_veh = c->n_veh; d.total = B * c->n_veh;
            d.w2 = square_as_python(speed->max_rate);
            const double b2 = square_as_python(speed->speed_bound);
            d.sp_sign = speed->speed_is_max ? -1.0 : 1.0; d.sp_offset = speed->speed_is_max ? b2 : -b2;
            if (c->speed2.d_out) {
                const double c2 = square_as_python(c->speed2.bound);
                d.out_speed2 = c->speed2.d_out;
                d.sp2_sign = c->speed2.is_max ? -1.0 : 1.0; d.sp2_offset = c->speed2.is_max ? c2 : -c2;
            }
            d.W2n = c->d_ang_w2n.as<double>(); d.W22n = c->d_ang_w22n.as<double>(); d.Wn = c->d_ang_wn.as<double>();
            d.fd = p.fd; d.fd_fixed = p.fd_fixed; d.fd_h = p.fd_h;
            p.dyn_first_block = (int)grid;
            grid += (unsigned)((d.total + kWave - 1) / kWave);
            lds = std::max(lds, lds_dyn);
            speed->did_dynamics = true;
        }
    }
    TimelineDump tl(c, grid, p.timeline);
    if (tl.rc) return tl.rc;
    {
        ScopedKernelTimer tm(c, OBTG_K_PAIR_SWEEP, true);
        launch_timed(tm, kern, dim3(grid), dim3(OBTG_SWEEP_THREADS), lds, c->stream, p);
    }
    {
        char hdr[256];
        snprintf(hdr, sizeof hdr, "# grid %u sweep_blocks %d wgs_per_row %d passes %d chunk %d B %d", grid,
                 p.dyn.out ? p.dyn_first_block : (int)grid, p.wgs_per_row, p.passes, p.chunk, B);
        if (int rc = tl.finish(hdr)) return rc;
    }
    c->gjk_len_cur ^= 1;
    c->gjk_len_rows = B;
    OBTG_HIP(c, hipGetLastError());
    return OBTG_OK;
}

// obtg_constraint_sweep_fd_structured_dev: the whole step of an FD view as ONE launch that evaluates row 0 in full and,
// per perturbed row, only what its vehicle touches (k_step_fd_structured).  OBTG_ERR_UNSUPPORTED for shapes outside the
// one-launch planar sweep (the caller uses the brute-force sweep, whose results are the same).
int launch_step_fd_structured(obtg_ctx* c, int B, double max_sep, double* d_out_sep, int max_iter, int md_cap, int* d_flag,
                              double* d_p1, double* d_p2, double* d_dist, int* d_nsup, int* d_status, SweepFold* speed)
{
    if (B <= 0) return OBTG_OK;
    if (!c->fd.Y0) return OBTG_ERR_ARG;
    if (c->ang_exact && c->R > 0) return OBTG_ERR_UNSUPPORTED;   // (obtg_ctx_set_ang_rate_order(2): the double-double pass follows the brute-force launches)
    const int nc = c->deg + 1;
    void (*kern)(const StructuredParams) = nullptr;
    void (*kern_mid)(const StructuredParams) = nullptr;      // two workgroups per CU: rows beyond 40 KB of LDS (C4: 79 KB)
    void (*kern_big)(const StructuredParams) = nullptr;      // one: beyond 80 KB
    const bool elev = c->R > 0;
    switch (nc) {
        case 4: kern = elev ? k_step_fd_structured<4, true> : k_step_fd_structured<4, false>; break;
        case 6: kern = elev ? k_step_fd_structured<6, true> : k_step_fd_structured<6, false>; break;
        case 8: kern = elev ? k_step_fd_structured<8, true> : k_step_fd_structured<8, false>; break;
        case 9: kern = elev ? k_step_fd_structured<9, true> : k_step_fd_structured<9, false>; break;
        case 11: kern = elev ? k_step_fd_structured<11, true> : k_step_fd_structured<11, false>;
                 if (!elev) { kern_mid = k_step_fd_structured<11, false, 2>; kern_big = k_step_fd_structured<11, false, 1>; }
                 break;
        case 16: if (!elev) { kern = k_step_fd_structured<16, false>; kern_mid = k_step_fd_structured<16, false, 2>;
                              kern_big = k_step_fd_structured<16, false, 1>; }
                 break;
        default: break;
    }
    const bool ok = kern && c->dim == 2 && c->polys_planar && c->max_poly_K <= nc && c->n_hull_pairs > 0 && !c->fd_dedup &&
                    c->n_pairs > 0 && speed && speed->d_out_ang && speed->d_out_speed &&
                    speed->d_tf && d_out_sep && c->n_veh + c->n_obs < 65535 && 2 * c->deg + c->R + 1 <= 512;      // (object ids as 16-bit halves of a word)
    if (!ok) return OBTG_ERR_UNSUPPORTED;
    int rc = ensure_tables(c);
    if (rc) return rc;
    if (c->d_ang_w22n.p == nullptr) return OBTG_ERR_UNSUPPORTED;
    if (elev && (c->ang_elevate_first || c->d_ang_T4.p == nullptr || c->d_ang_cv2.p == nullptr || c->d_Tf.p == nullptr))
        return OBTG_ERR_UNSUPPORTED;
    StructuredParams sp{};
    GjkSwarmParams& p = sp.g;
    p.Y = c->fd.Y0; p.fd = 1 + c->fd.row0; p.fd_fixed = c->fd.fixed; p.fd_h = c->fd.h;
    p.poly = c->d_poly_pts.as<double>(); p.poly_off = c->d_poly_off.as<int>();
    p.pa = c->d_hp_a.as<int>(); p.pb = c->d_hp_b.as<int>();
    p.n_veh = c->n_veh; p.dim = c->dim; p.nc = nc; p.n_poly = c->n_poly;
    p.n_poly_pts = c->n_poly_pts; p.n_pairs = c->n_hull_pairs;
    p.max_iter = max_iter; p.md_cap = md_cap; p.refill_min = sweep_refill_min(); p.hist_shift = sweep_hist_shift();
    p.flag = d_flag; p.p1 = d_p1; p.p2 = d_p2; p.dist = d_dist; p.nsup = d_nsup; p.status = d_status;
    p.B = B; p.chunk = 256; p.wgs_per_row = 1; p.passes = 1;
    p.ts.pairs = c->d_pairs.as<int2>(); p.ts.W2 = c->d_w2.as<double>(); p.ts.out = d_out_sep;
    p.ts.n_pairs = c->n_pairs; p.ts.sign = 1.0; p.ts.offset = 0.0 - square_as_python(max_sep);
    p.ts.Td = c->d_Td.as<double>(); p.ts.Tf = c->d_Tf.as<double>(); p.ts.R = c->R;
    if (c->n_obs > 0) {                // point obstacles (optimization.py:86-98): objects of the separation pair table only
        p.obs = c->d_obs.as<double>(); p.n_obs = c->n_obs;
        p.ts.n_veh = c->n_veh; p.ts.obs_shift = c->n_poly; p.ts.n_tobj = c->n_veh + c->n_obs;
    }
    sp.cv4 = c->d_ang_T4.as<double>(); sp.cv2 = c->d_ang_cv2.as<double>(); sp.R = c->R;
    {
        AngParams& d = p.dyn;
        d.Y = p.Y; d.tf = speed->d_tf; d.out = speed->d_out_ang; d.out_speed = speed->d_out_speed;
        d.n_veh = c->n_veh; d.total = B * c->n_veh;
        d.w2 = square_as_python(speed->max_rate);
        const double b2 = square_as_python(speed->speed_bound);
        d.sp_sign = speed->speed_is_max ? -1.0 : 1.0; d.sp_offset = speed->speed_is_max ? b2 : -b2;
        if (c->speed2.d_out) {
            const double c2 = square_as_python(c->speed2.bound);
            d.out_speed2 = c->speed2.d_out;
            d.sp2_sign = c->speed2.is_max ? -1.0 : 1.0; d.sp2_offset = c->speed2.is_max ? c2 : -c2;
        }
        d.W2n = c->d_ang_w2n.as<double>(); d.W22n = c->d_ang_w22n.as<double>(); d.Wn = c->d_ang_wn.as<double>();
        d.fd = p.fd; d.fd_fixed = p.fd_fixed; d.fd_h = p.fd_h;
    }
    const int n_obj = c->n_veh + c->n_poly, vpq = nc | 1, L = 2 * c->deg + 1, LR = L + c->R;
    // S: one workgroup per (64-pair group, row range); about two thousand workgroups of streams
    sp.n_sep_groups = (c->n_pairs + kWave - 1) / kWave;
    // (a large step -- C4: 58 GB of separation rows -- gets more, so that a stream stays near 4 MB)
    const double sep_stream_bytes = 8.0 * kWave * LR * (double)B * sp.n_sep_groups;
    // (small batches -- a rank's share of a row-sharded iteration -- get fewer, longer streams: every S workgroup repeats row
    // 0's products for its group and every G workgroup its gjkNew chunk, and 2048 + 1024 of them are three rounds of resident
    // workgroups for a few rows each.  C3, tools/r04_struct_small_batch_scan.sh: B = 145 0.044 -> 0.027 ms with 512 / 384
    // workgroups, B = 289 0.055 -> 0.040 with 1024 / 512; from B = 577 on the old numbers are the best)
    const double s_floor = std::min(2048.0, std::max(256.0, 3.5 * B));
    int s_target = (int)std::min(32768.0, std::max(s_floor, sep_stream_bytes / (4 << 20)));
    if (const char* e = getenv("OBTG_STRUCT_SEP_WGS")) s_target = std::max(1, atoi(e));      // (experiments)
    int s_ranges = std::max(1, std::min(B, s_target / std::max(1, sp.n_sep_groups)));
    sp.sep_rows_per = (B + s_ranges - 1) / s_ranges;
    s_ranges = (B + sp.sep_rows_per - 1) / sp.sep_rows_per;
    // G: chunks of ~80 hull pairs (one short gjkNew phase per workgroup), row ranges for ~512 workgroups
    sp.gjk_chunk_pairs = 16 * (size_t)n_obj * vpq > 40 * 1024 ? 64 : 80;       // (large rows: the chunk's results behind 70 KB of hulls stay under 80 KB)
    if (const char* e = getenv("OBTG_STRUCT_GJK_CHUNK")) sp.gjk_chunk_pairs = std::max(16, atoi(e));
    const int g_target = getenv("OBTG_STRUCT_GJK_WGS") ? std::max(1, atoi(getenv("OBTG_STRUCT_GJK_WGS")))
                                                       : (int)std::min(1024.0, std::max(192.0, 2.6 * B));
    sp.gjk_chunks = (c->n_hull_pairs + sp.gjk_chunk_pairs - 1) / sp.gjk_chunk_pairs;
    const int g_target_b = (int)std::min(32768.0, std::max((double)g_target, 68.0 * c->n_hull_pairs * (double)B / (4 << 20)));
    int g_ranges = std::max(1, std::min(B, g_target_b / std::max(1, sp.gjk_chunks)));
    sp.gjk_rows_per = (B + g_ranges - 1) / g_ranges;
    g_ranges = (B + sp.gjk_rows_per - 1) / sp.gjk_rows_per;
    sp.fix_chunk = 256;
    p.vp_off = c->d_vp_off.as<int>(); p.vp_idx = c->d_vp_idx.as<int>();
    sp.n_kind[0] = sp.n_sep_groups * s_ranges;
    sp.first_pert = c->fd.row0 > 0 ? 0 : 1;            // (a range that starts later: its local row 0 is a perturbed row too)
    sp.n_kind[1] = B - sp.first_pert;
    sp.n_kind[2] = sp.gjk_chunks * g_ranges;
    // D: ~256 streams of row 0's groups (at most 64 rows each, at least 4 when there are that many: at C5 a stream of 19
    // rows keeps its workgroup for 430 us of a 650 us launch, one of 5 rows for 190), the advanced vehicles 64 to a
    // group, and one workgroup per 8 rows that looks for rows with their own tf
    sp.dyn_groups_per_row = (c->n_veh + kWave - 1) / kWave;
    sp.dyn_rows_per = std::min(64, std::max((B + 255) / 256, std::min(4, B)));
    // (DEG_ELEV > 0: stream workgroups of 16 vehicles that collect their rows in LDS -- while that area fits beside the tables)
    sp.dyn_split = elev && !(getenv("OBTG_STRUCT_DYN_SPLIT") && getenv("OBTG_STRUCT_DYN_SPLIT")[0] == '0') &&
                   sizeof(double) * (dyn_elev_lds_doubles(c->deg, c->R) + dyn_elev_stage_doubles(c->deg, c->R)) <= 72 * (size_t)1024;
    if (sp.dyn_split) sp.dyn_rows_per = std::min(64, 4 * sp.dyn_rows_per);      // (16 vehicles per stream workgroup instead of 64: the same bytes)
    if (const char* e = getenv("OBTG_STRUCT_DYN_ROWS")) sp.dyn_rows_per = std::max(1, std::min(64, atoi(e)));
    sp.dyn_streams = (sp.dyn_split ? 4 : 1) * sp.dyn_groups_per_row * ((B + sp.dyn_rows_per - 1) / sp.dyn_rows_per);
    sp.dyn_fix_groups = (B - sp.first_pert + kWave - 1) / kWave;
    const int dyn_x = sp.dyn_groups_per_row * ((B - 1 + 7) / 8);
    sp.n_kind[3] = sp.dyn_streams + sp.dyn_fix_groups + dyn_x;
    unsigned grid = 0;
    {
        // shares of every 16 block ids by expected work (workgroups x duration on the C3 timeline: S 12.5, F 15.2, G 20.2, D 10.7 us)
        // (what decides is where each kind's ids run out: the kinds should end at about the same block id, so the D
        // weights are those of its neighbours, not its 190 us streams: 7:3:3:3 instead of 7:4:4:1 costs C5 10 %)
        const double cost_flat[4] = { 12.5, 15.2, 20.2, 16.0 }, cost_elev[4] = { 40.0, 36.0, 38.0, 40.0 };
        const double* cost = elev ? cost_elev : cost_flat;     // (DEG_ELEV > 0: the streams are 2n+R+1 columns wide, the dynamics groups k_dynamics_elev's)
        double w[4], tot = 0.0;
        for (int k = 0; k < 4; ++k) { w[k] = sp.n_kind[k] * cost[k]; tot += w[k]; }
        tot -= w[3];
        w[3] = (sp.dyn_streams + sp.dyn_fix_groups) * cost[3];      // (the workgroups that look for rows with their own tf come last and cost nothing)
        tot += w[3];
        int sum = 0;
        for (int k = 0; k < 4; ++k) { sp.per16[k] = sp.n_kind[k] > 0 ? std::max(1, (int)(16.0 * w[k] / tot + 0.5)) : 0; sum += sp.per16[k]; }
        if (const char* e = getenv("OBTG_STRUCT_PER16")) {      // (experiments: "S,F,G,D" summing to 16)
            int v[4] = { 0, 0, 0, 0 };
            if (sscanf(e, "%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3]) == 4 && v[0] + v[1] + v[2] + v[3] == 16) {
                for (int k = 0; k < 4; ++k) sp.per16[k] = v[k];
                sum = 16;
            }
        }
        while (sum != 16) {             // give to / take from the kind with the largest share
            int kmax = 0;
            for (int k = 1; k < 4; ++k) if (sp.per16[k] > sp.per16[kmax]) kmax = k;
            sp.per16[kmax] += sum < 16 ? 1 : -1;
            sum += sum < 16 ? 1 : -1;
        }
        // spread each kind's slots evenly over the 16
        struct Slot { double pos; int kind; };
        std::vector<Slot> slots;
        for (int k = 0; k < 4; ++k)
            for (int j = 0; j < sp.per16[k]; ++j) slots.push_back({ (j + 0.5) / sp.per16[k] + 1e-3 * k, k });
        std::sort(slots.begin(), slots.end(), [](const Slot& a, const Slot& b) { return a.pos < b.pos; });
        int seen[4] = { 0, 0, 0, 0 };
        int groups = 0;
        for (int i = 0; i < 16; ++i) { sp.pat[i] = (unsigned char)slots[i].kind; sp.rank[i] = (unsigned char)seen[slots[i].kind]++; }
        for (int k = 0; k < 4; ++k)
            if (sp.per16[k] > 0) groups = std::max(groups, (sp.n_kind[k] + sp.per16[k] - 1) / sp.per16[k]);
        grid = (unsigned)groups * 16u;
    }
    p.dyn_first_block = 0;
    const int n_sobj = c->n_veh + c->n_obs;                  // what an S workgroup stages: vehicles and point obstacles
    size_t lds_s = std::max((size_t)16 * n_sobj * vpq, sizeof(double) * kWave * L);      // (flat: the tile overlays the staged row)
    size_t lds_d_elev = 0;
    if (elev) {
        // staged row + the group's coefficient image [64][PA] + one 16-row output tile of up to 128 columns
        const int KS = (L + 3) / 4, PA = 4 * KS + 2;
        lds_s = (size_t)16 * n_sobj * vpq + sizeof(double) * (kWave * PA + 16 * (size_t)std::min(LR, 128));
        lds_d_elev = sizeof(double) * (dyn_elev_lds_doubles(c->deg, c->R) + (sp.dyn_split ? dyn_elev_stage_doubles(c->deg, c->R) : 0));
    }
    const size_t lds_g = (planar_lds_bytes<0>(n_obj, vpq, sp.gjk_chunk_pairs) + 15) / 16 * 16 + (size_t)sp.gjk_chunk_pairs * 68 + 16;
    size_t lds_f = std::max(planar_lds_bytes<1>(n_obj, vpq, sp.fix_chunk), (size_t)16 * (n_obj + c->n_obs) * vpq);
    if (elev) {                        // the fix-up rows' image [32][PA] and tile behind the staged objects
        const int KS = (L + 3) / 4, PA = 4 * KS + 2;
        lds_f = std::max(lds_f, (size_t)16 * (n_obj + c->n_obs) * vpq + sizeof(double) * (32 * PA + 16 * (size_t)std::min(LR, 128)));
    }
    const size_t lds_d = sizeof(double) * ((size_t)kWave * (4 * c->deg + 1) + (size_t)kWave * L);
    const size_t lds = std::max(std::max(lds_s, lds_g), std::max(lds_f, elev ? lds_d_elev : lds_d));
    if (lds > (elev ? 76 : 40) * (size_t)1024) {
        const size_t with_static = lds + 1536;               // (the kernel's own __shared__ variables)
        if (kern_mid && with_static <= 80 * (size_t)1024) kern = kern_mid;
        else if (kern_big && with_static <= 158 * (size_t)1024) kern = kern_big;
        else return OBTG_ERR_UNSUPPORTED;
    }
    if (lds > 48 * 1024)
        OBTG_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    TimelineDump tl(c, grid, p.timeline);
    if (tl.rc) return tl.rc;
    {
        ScopedKernelTimer tm(c, OBTG_K_PAIR_SWEEP, true);
        launch_timed(tm, kern, dim3(grid), dim3(256), lds, c->stream, sp);
    }
    OBTG_HIP(c, hipGetLastError());
    char hdr[256];
    snprintf(hdr, sizeof hdr, "# grid %u structured n_S %d n_F %d n_G %d n_D %d per16 %d %d %d %d pat %d%d%d%d%d%d%d%d%d%d%d%d%d%d%d%d B %d",
             grid, sp.n_kind[0], sp.n_kind[1], sp.n_kind[2], sp.n_kind[3], sp.per16[0], sp.per16[1], sp.per16[2], sp.per16[3],
             sp.pat[0], sp.pat[1], sp.pat[2], sp.pat[3], sp.pat[4], sp.pat[5], sp.pat[6], sp.pat[7], sp.pat[8], sp.pat[9], sp.pat[10],
             sp.pat[11], sp.pat[12], sp.pat[13], sp.pat[14], sp.pat[15], B);
    return tl.finish(hdr);
}

size_t min_dist2poly_stack_doubles(int K, int max_depth)      // per pair: the larger of the frame forms
{
    const int fr = 3 * K + G_NSCAL, bl = K <= kMdQuadMaxK ? md2_quad_frame(K) : 0;
    return (size_t)max_depth * (fr > bl ? fr : bl);
}

// worker waves of the wave-per-pair searches: waves per SIMD x 4 SIMDs x CUs, never more than pairs (OBTG_MD_WAVES_PER_SIMD)
int min_dist_workers(const obtg_ctx* c, int n_pairs, size_t lds_per_wave, int waves_per_simd)
{
    static const int env = getenv("OBTG_MD_WAVES_PER_SIMD") ? atoi(getenv("OBTG_MD_WAVES_PER_SIMD")) : 0;
    int per_simd = env > 0 ? env : waves_per_simd;
    const int by_lds = (int)((size_t)160 * 1024 / (lds_per_wave ? lds_per_wave : 1)) / 4;   // a CU's LDS over its four SIMDs
    if (per_simd > by_lds) per_simd = by_lds > 0 ? by_lds : 1;
    const long w = (long)per_simd * 4 * (c->n_cus > 0 ? c->n_cus : 256);
    return (int)(w < n_pairs ? w : n_pairs);
}

// which forms of the curve <-> curve search the LDS budget allows for (K, max_depth), and what each asks per worker wave
struct MdPlan {
    size_t lds_w, lds_q;
    bool wave_ok, quad_ok;
};
static MdPlan md_plan(int K, int max_depth)
{
    MdPlan pl;
    pl.lds_w = sizeof(double) * ((size_t)12 * K + 6 * kMdMaxK + (size_t)max_depth * F_NSCAL);
    pl.lds_q = sizeof(double) * ((size_t)2 * md_quad_blob(K) + 8 * kMdShRow + 64 + (size_t)(max_depth < kMdScsLds ? max_depth : kMdScsLds) * Q_NSCAL);
    pl.wave_ok = pl.lds_w <= 48 * 1024;
    pl.quad_ok = K <= kMdQuadMaxK && pl.lds_q <= 48 * 1024;
    return pl;
}

// doubles of frame stack obtg_min_dist has to provide: a stack per WORKER wave for the forms that run as workers on the queue
// (either may be chosen at launch), a stack per pair only when neither fits and the one-lane form runs
size_t min_dist_stack_doubles(const obtg_ctx* c, int K, int max_depth, int n_pairs)
{
    const MdPlan pl = md_plan(K, max_depth);
    size_t need = 0;
    if (pl.quad_ok) need = (size_t)min_dist_workers(c, n_pairs, pl.lds_q, OBTG_MD_MIN_WAVES_PLANAR) * max_depth * md_quad_frame(K);
    if (pl.wave_ok) {
        const size_t w = (size_t)min_dist_workers(c, n_pairs, pl.lds_w, OBTG_MD_MIN_WAVES) * max_depth * (6 * K + F_NSCAL);
        if (w > need) need = w;
    }
    if (!pl.quad_ok && !pl.wave_ok) need = (size_t)n_pairs * max_depth * (6 * K + F_NSCAL);
    return need;
}

int launch_min_dist(obtg_ctx* c, const double* d_curves, int K, const int* d_pa, const int* d_pb,
                    int n_pairs, double eps, int max_iter, int md_cap, int max_depth, int max_nodes,
                    double* d_stack, double* d_res, int* d_info, const int* d_order, int* d_queue, bool planar)
{
    if (n_pairs <= 0) return OBTG_OK;
    if (K < 2 || K > kMdMaxK || max_depth < 1) return OBTG_ERR_UNSUPPORTED;
    MdParams p{ d_curves, d_pa, d_pb, n_pairs, K, max_iter, md_cap, max_depth, max_nodes, eps, d_stack, d_res, d_info,
                d_order, d_queue };
    ScopedKernelTimer t(c, OBTG_K_MIN_DIST);
    const MdPlan pl = md_plan(K, max_depth);
    const size_t lds_w = pl.lds_w, lds_q = pl.lds_q;
    const char* env_form = getenv("OBTG_MD_FORM");          // "wave": a wavefront per gjkNew call (read per launch: the A/B test flips it)
    const bool quad = pl.quad_ok && d_queue && !(env_form && !strcmp(env_form, "wave"));
    if (quad) {                            // a 16-lane row per child: four gjkNew calls of a node's children in lockstep
        OBTG_HIP(c, hipMemsetAsync(d_queue, 0, sizeof(int), c->stream));
        // (both read per launch: the tests flip them in-process)
        const char* env_planar = getenv("OBTG_MD_PLANAR");                       // "0": the 3-D machine on planar curves too (A/B runs)
        const bool no_planar = env_planar && env_planar[0] == '0';
        const char* env_many = getenv("OBTG_MD_MANY");                           // pairs per worker from which a call counts as issue bound
        const int many_env = env_many ? atoi(env_many) : 0;
        const int w3 = min_dist_workers(c, n_pairs, lds_q, OBTG_MD_MIN_WAVES_PLANAR), w2 = min_dist_workers(c, n_pairs, lds_q, OBTG_MD_MIN_WAVES);
        const bool many = (long)n_pairs >= (long)(many_env > 0 ? many_env : 8) * w3;
        const int form = (planar && !no_planar) ? (many ? 2 : 1) : 0;      // 0: the 3-D machine, 1: planar, chain bound, 2: planar, issue bound
        void (*kern)(const MdParams) = form == 2 ? k_min_dist_quad<true, OBTG_MD_MIN_WAVES_PLANAR, 0>
                                                 : (form == 1 ? k_min_dist_quad<true, OBTG_MD_MIN_WAVES, 0> : k_min_dist_quad<false, OBTG_MD_MIN_WAVES, 0>);
        switch (K) {        // the counts with a build of their own
#define OBTG_CASE(NC_) \
        case NC_: \
            kern = form == 2 ? k_min_dist_quad<true, OBTG_MD_MIN_WAVES_PLANAR, NC_> \
                             : (form == 1 ? k_min_dist_quad<true, OBTG_MD_MIN_WAVES, NC_> : k_min_dist_quad<false, OBTG_MD_MIN_WAVES, NC_>); \
            break;
            OBTG_NC_DYN(OBTG_CASE)
#undef OBTG_CASE
            default: break;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)(form == 2 ? w3 : w2)), dim3(kWave), lds_q, c->stream, p);
    } else if (pl.wave_ok && d_queue) {    // one pair per wavefront at a time, the waves as workers on a queue
        OBTG_HIP(c, hipMemsetAsync(d_queue, 0, sizeof(int), c->stream));
        hipLaunchKernelGGL(k_min_dist_wave, dim3((unsigned)min_dist_workers(c, n_pairs, lds_w, OBTG_MD_MIN_WAVES)), dim3(kWave), lds_w, c->stream, p);
    } else
        hipLaunchKernelGGL(k_min_dist, dim3((n_pairs + 63) / 64), dim3(64), 0, c->stream, p);
    OBTG_HIP(c, hipGetLastError());
    return OBTG_OK;
}

// _minDist2Poly with one (curve, polygon) pair per wavefront: as k_min_dist_wave, with the polygon (<= 32
// vertices) copied to LDS once and only the curve in the frames.
__global__ __launch_bounds__(64) void k_min_dist2poly_wave(const Md2Params p)
{
    extern __shared__ double m2_lds[];
    const int k = blockIdx.x, lane = threadIdx.x, li = lane & 31;
    const int K = p.K, FR = 3 * K + G_NSCAL;
    const int rl = lane / K, il = lane - rl * K;          // (row, point) of the lane in the level-parallel splits
    const int po = p.off[p.pp[k]], PK = p.off[p.pp[k] + 1] - po;
    double* cur = m2_lds;                       // [3K] curve of the node being evaluated
    double* nxt = cur + 3 * K;                  // [3K] the child being built
    double* pol = nxt + 3 * K;                  // [3][32] polygon, SoA
    double* sh_e = pol + 3 * kMdMaxK;           // [kMdMaxK] (+ the three scratch rows of a split, with sh_q)
    double* sh_q = sh_e + kMdMaxK;              // [kMdMaxK] + 2 rows
    double* scs = sh_q + 3 * kMdMaxK;           // [max_depth][G_NSCAL]
    double* st = p.stack + (size_t)k * p.max_depth * FR;
    const double* ca = p.curves + (size_t)p.pc[k] * 3 * K;
    for (int i = lane; i < 3 * K; i += kWave) { const double a = ca[i]; st[i] = a; cur[i] = a; }
    for (int i = lane; i < 3 * PK; i += kWave) pol[(i / PK) * kMdMaxK + (i % PK)] = p.soa[3 * po + i];
    if (lane == 0) { scs[G_T1L] = 0; scs[G_T1H] = 1; scs[G_ALPHA] = INFINITY; scs[G_STATE] = 0; }
    __syncthreads();
    int depth = 0, cur_depth = 0, nodes = 0, calls = 0, dmax = 0, status = OBTG_MD_OK;
    double r0 = INFINITY, r1 = -1, rx = -1, ry = -1, rz = -1;
    bool returning = false;
    for (;;) {
        double* f = st + (size_t)depth * FR;
        double* sc = scs + depth * G_NSCAL;
        int state = (int)sc[G_STATE];
        if (!returning && state == 0) {
            if (depth + 1 > 1000) { r0 = r1 = rx = -1; ry = rz = -1; returning = true; depth--; if (depth < 0) break; continue; }
            if (nodes >= p.max_nodes) { status = OBTG_MD_NODE_CAP; break; }
            nodes++;
            if (depth + 1 > dmax) dmax = depth + 1;
            Ctx<MemLds> g;
            g.mem = MemLds{ m2_lds };
            g.P1 = Poly{ (int)(cur - m2_lds), K, K, 1 };
            g.P2 = Poly{ (int)(pol - m2_lds), kMdMaxK, PK, 1 };
            g.trace = nullptr; g.trace_cap = 0; g.n_support = 0;
            Result gr;
            gjk::run<MemLds, false, true>(g, p.max_iter, p.md_cap, gr);
            calls++;
            if (gr.status == OBTG_ST_MD_CAP || gr.status == OBTG_ST_CYCLE) { status = OBTG_MD_GJK_CAP; break; }
            double lb, t1, nT1, alpha = sc[G_ALPHA];
            double cx, cy, cz;
            if (gr.flag > 0) {
                lb = gr.dist;
                const double tp = hull_param_wave(cur, K, gr.c1, sh_e, sh_q, lane < 32 ? li : kMdMaxK);
                t1 = __shfl(tp, 0);
                cx = gr.c2.x; cy = gr.c2.y; cz = gr.c2.z;
                const double d0 = norm_seq(cur[0], cur[K], cur[2 * K], cx, cy, cz);
                const double d1 = norm_seq(cur[K - 1], cur[2 * K - 1], cur[3 * K - 1], cx, cy, cz);
                int am = (d1 < d0) ? 1 : 0;
                if (d0 != d0) am = 0; else if (d1 != d1) am = 1;
                const double ub = am ? d1 : d0, t1loc = am ? 1.0 : 0.0;
                if (ub <= alpha) { alpha = ub; nT1 = (1 - t1loc) * sc[G_T1L] + t1loc * sc[G_T1H]; }
                else nT1 = -1;
            } else {
                t1 = 0.5; nT1 = -1; cx = cy = cz = -1; lb = p.eps3;
            }
            if (lb >= alpha * (1 - p.eps)) {
                r0 = alpha; r1 = nT1; rx = cx; ry = cy; rz = cz; returning = true; depth--;
                if (depth < 0) break;
                continue;
            }
            if (depth + 1 >= p.max_depth) { status = OBTG_MD_DEPTH_CAP; r0 = alpha; r1 = nT1; rx = cx; ry = cy; rz = cz; break; }
            if (t1 != t1) t1 = 0;
            wave_sync();
            if (lane == 0) {
                sc[G_T1] = t1; sc[G_ALPHA] = alpha; sc[G_RT1] = nT1; sc[G_PX] = cx; sc[G_PY] = cy; sc[G_PZ] = cz; sc[G_STATE] = 1;
            }
            wave_sync();
            state = 1;
        }
        if (returning) {
            if (r0 < sc[G_ALPHA]) {
                wave_sync();
                if (lane == 0) { sc[G_ALPHA] = r0; sc[G_RT1] = r1; sc[G_PX] = rx; sc[G_PY] = ry; sc[G_PZ] = rz; }
                wave_sync();
            }
            returning = false;
            state = (int)sc[G_STATE];
        }
        if (state >= 3) {
            r0 = sc[G_ALPHA]; r1 = sc[G_RT1]; rx = sc[G_PX]; ry = sc[G_PY]; rz = sc[G_PZ];
            returning = true; depth--;
            if (depth < 0) break;
            continue;
        }
        {
            if (cur_depth != depth) {
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                for (int i = lane; i < 3 * K; i += kWave) cur[i] = f[i];
                cur_depth = depth;
                wave_sync();
            }
            const int h1 = state - 1;
            const double t1 = sc[G_T1];
            double* nf = f + FR;
            if (3 * K <= kWave) split_rows3_wave(cur, K, t1, h1, nxt, rl, il, sh_e);  // level-parallel (see k_min_dist_wave); sh_e + sh_q: 64 idle doubles
            else if (lane < 3) split_row_lds(cur + lane * K, K, t1, h1, nxt + lane * K, sh_e + lane * kMdMaxK);
            wave_sync();
            for (int i = lane; i < 3 * K; i += kWave) nf[i] = nxt[i];
            const double t1len = sc[G_T1H] - sc[G_T1L];
            const double m1 = sc[G_T1L] + t1 * t1len;
            const double a_in = sc[G_ALPHA];
            const double n1l = h1 ? m1 : sc[G_T1L], n1h = h1 ? sc[G_T1H] : m1;
            wave_sync();
            if (lane == 0) {
                double* ns = sc + G_NSCAL;
                ns[G_T1L] = n1l; ns[G_T1H] = n1h; ns[G_ALPHA] = a_in; ns[G_STATE] = 0;
                sc[G_STATE] = state + 1;
            }
            double* tsw = cur; cur = nxt; nxt = tsw;
            depth++;
            cur_depth = depth;
            wave_sync();
        }
    }
    if (lane == 0) {
        p.res[5 * k] = r0; p.res[5 * k + 1] = r1; p.res[5 * k + 2] = rx; p.res[5 * k + 3] = ry; p.res[5 * k + 4] = rz;
        if (p.info) { p.info[4 * k] = nodes; p.info[4 * k + 1] = calls; p.info[4 * k + 2] = dmax; p.info[4 * k + 3] = status; }
    }
}

// ---- _minDist2Poly, both children at a time (round 5): k_min_dist_quad's arrangement for bezier.py:1411-1496, whose nodes have TWO
// children (the curve's pieces against the same polygon), both always visited, the cut taken inside the child -- so the children's
// gjkNew calls, split parameters and end-point bounds are worked out when the parent is split, a 16-lane row of the wavefront each
// (curve and polygon of at most 16 points; rows 2 and 3 repeat rows 0 and 1), in lockstep.  Why it matters here: a launch of the
// wave form lasts as long as the pair whose inner gjkNew never converges (md_cap rounds of minimumDistance: 3 such pairs among the
// 4096 of bench.py --mode mindist took 5.3 ms), and a lockstep trip of 16-lane rows is a third of a wavefront-wide one.
// A frame's blob: the curve's two pieces (left, right), then the two children's records.

__device__ __forceinline__ void split_one_both(const double* c, int K, double t, double* blob, double* dump)
{
    switch (K) {        // (wave-uniform; 3 K <= 48 lanes)
#define OBTG_CASE(NC_) case NC_: if (NC_ <= kMdQuadMaxK) { split_both_t<NC_>(c, c, K, t, t, blob, 0, 3, dump); return; } break;
        OBTG_NC_DYN(OBTG_CASE)
#undef OBTG_CASE
        default: break;
    }
    split_both_t<0>(c, c, K, t, t, blob, 0, 3, dump);
}

// the record of the (curve piece, polygon) pair a row of the wavefront names: gjkNew, the curve's split parameter (bezier.py:
// 1436-1452), _upperboundPoly (the curve's end points against the polygon's closest point)
template <bool PLANAR = false, int KC = 0>
__device__ __forceinline__ void md2_eval_rows(const double* lds, int oc, int K, int op, int PK, int max_iter, int md_cap,
                                              double* rec, double* sh_e, double* sh_q)
{
    Ctx<MemLds> g;
    g.mem = MemLds{ lds };
    g.P1 = Poly{ oc, K, K, 1 };
    g.P2 = Poly{ op, 16, PK, 1 };
    g.trace = nullptr; g.trace_cap = 0; g.n_support = 0;
    Result gr;
    if constexpr (PLANAR) gjk::run_quarter2<MemLds>(g, max_iter, md_cap, gr);
    else gjk::run_quarter<MemLds>(g, max_iter, md_cap, gr);
    const bool cap = gr.status == OBTG_ST_MD_CAP || gr.status == OBTG_ST_CYCLE;
    const bool pos = gr.flag > 0 && !cap;
    const double* c1 = lds + oc;
    double t1 = 0.5, ub = INFINITY;
    int am = 0;
    if (pos) {
        if constexpr (KC > 0) { double t2u = 0.0; hull_param_quarter2_t<KC, false, PLANAR>(c1, c1, gr.c1, gr.c1, sh_e, sh_q, t1, t2u); }
        else t1 = hull_param_row<PLANAR>(c1, K, gr.c1, sh_e, sh_q);
        // (PLANAR: the curve's z row and the closest point's z are exact zeros; norm_seq's `+= dz * dz` adds 0)
        const double d0 = norm_seq(c1[0], c1[K], PLANAR ? 0.0 : c1[2 * K], gr.c2.x, gr.c2.y, PLANAR ? 0.0 : gr.c2.z);
        const double d1 = norm_seq(c1[K - 1], c1[2 * K - 1], PLANAR ? 0.0 : c1[3 * K - 1], gr.c2.x, gr.c2.y, PLANAR ? 0.0 : gr.c2.z);
        am = (d1 < d0) ? 1 : 0;
        if (d0 != d0) am = 0; else if (d1 != d1) am = 1;
        ub = am ? d1 : d0;
    }
    rec[P_CAP] = cap ? 1.0 : 0.0; rec[P_POS] = pos ? 1.0 : 0.0; rec[P_LB] = gr.dist; rec[P_T1] = t1; rec[P_UB] = ub;
    rec[P_AM] = (double)am; rec[P_CX] = gr.c2.x; rec[P_CY] = gr.c2.y; rec[P_CZ] = gr.c2.z;
}

#ifndef OBTG_MD2_MIN_WAVES_PLANAR
#define OBTG_MD2_MIN_WAVES_PLANAR 3     // 4096 curve-polygon pairs: 2 / 3 / 4 workers per SIMD 2.82 / 2.70 / 2.85 ms -- the launch is its deepest search's chain
#endif
// PLANAR: the curves and the polygons of the call all have z == 0 (the host has looked): the planar gjkNew machine per row
template <bool PLANAR, int KC>        // KC: the curve's control-point count when the kernel is built for one (see k_min_dist_quad)
__global__ __launch_bounds__(64, PLANAR ? OBTG_MD2_MIN_WAVES_PLANAR : OBTG_MD_MIN_WAVES) void k_min_dist2poly_quad(const Md2Params p)
{
    extern __shared__ double m2q_lds[];
    const int k = blockIdx.x, lane = threadIdx.x, q = (lane >> 4) & 1;        // rows 2, 3 repeat rows 0, 1
    const int K = KC > 0 ? KC : p.K, BL = md2_quad_blob(K), FRM = p.quad_frame;   // FRM: doubles per frame of this call's stack (>= md2_quad_frame(K))
    const int po = p.off[p.pp[k]], PK = p.off[p.pp[k] + 1] - po;
    double* st = p.stack + (size_t)k * p.max_depth * FRM;
    double* cur = m2q_lds;                      // [BL] blob of the frame `cur_depth`
    double* nxt = cur + BL;                     // [BL] blob being built
    double* pol = nxt + BL;                     // [3][16] polygon, SoA
    double* sh_e = pol + 48;                    // [4][kMdShRow]
    double* sh_q = sh_e + 4 * kMdShRow;
    double* dump = sh_q + 4 * kMdShRow;         // [64]
    double* scs = dump + 64;                    // [min(max_depth, kMdScsLds)][G_NSCAL] frame scalars; deeper frames: st + d * FRM + BL
    const double* ca = p.curves + (size_t)p.pc[k] * 3 * K;
    for (int i = lane; i < 3 * K; i += kWave) { const double a = ca[i]; nxt[i] = a; nxt[3 * K + i] = a; }      // frame "-1": pieces c, c
    for (int i = lane; i < 3 * PK; i += kWave) pol[(i / PK) * 16 + (i % PK)] = p.soa[3 * po + i];
    // the frame whose children the walk is going through, in registers (see k_min_dist_quad); frame -1: [0, 1] split "at 1"
    double f_t1l = 0, f_t1h = 1, f_t1 = 1, f_alpha = INFINITY, f_rt1 = -1, f_px = -1, f_py = -1, f_pz = -1;
    int f_next = 0;
    int depth = -1, cur_depth = -1, eval_depth = -1;
    int nodes = 0, calls = 0, dmax = 0, status = OBTG_MD_OK;
    double r0 = INFINITY, r1 = -1, rx = -1, ry = -1, rz = -1;
    bool done = false;
    while (!done) {
        wave_sync();
        md2_eval_rows<PLANAR, KC>(m2q_lds, (int)(nxt - m2q_lds) + q * 3 * K, K, (int)(pol - m2q_lds), PK, p.max_iter, p.md_cap,
                      nxt + 6 * K + q * P_NREC, sh_e + (lane >> 4) * kMdShRow, sh_q + (lane >> 4) * kMdShRow);
        wave_sync();
        if (eval_depth >= 0) {
            double* f = st + (size_t)eval_depth * FRM;
            for (int i = lane; i < BL; i += kWave) f[i] = nxt[i];
        }
        { double* tsw = cur; cur = nxt; nxt = tsw; }
        cur_depth = eval_depth;
#define OBTG_MD2_RETURN() \
    { if (depth < 0) { done = true; break; } \
      if (r0 < f_alpha) { f_alpha = r0; f_rt1 = r1; f_px = rx; f_py = ry; f_pz = rz; } \
      continue; }
        for (;;) {
            if (f_next >= 2) {                   // both children done: this frame's value goes to its parent
                r0 = f_alpha; r1 = f_rt1; rx = f_px; ry = f_py; rz = f_pz;
                depth--;
                if (depth < 0) { done = true; break; }
                const bool deep = depth >= kMdScsLds;
                if (deep) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                const double* sc = deep ? st + (size_t)depth * FRM + BL : scs + depth * G_NSCAL;
                f_t1 = sc[G_T1]; f_t1l = sc[G_T1L]; f_t1h = sc[G_T1H]; f_alpha = sc[G_ALPHA]; f_rt1 = sc[G_RT1];
                f_px = sc[G_PX]; f_py = sc[G_PY]; f_pz = sc[G_PZ]; f_next = (int)sc[G_STATE];
                OBTG_MD2_RETURN()
            }
            const int h1 = f_next++;             // child: the left (0) / right (1) piece, a node at depth + 1
            if (depth + 2 > 1000) { r0 = r1 = rx = -1; ry = rz = -1; OBTG_MD2_RETURN() }
            if (nodes >= p.max_nodes) { status = OBTG_MD_NODE_CAP; done = true; break; }
            nodes++;
            if (depth + 2 > dmax) dmax = depth + 2;
            if (cur_depth != depth) {            // the walk came back up: fetch this frame's blob again
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                const double* f = st + (size_t)depth * FRM;
                for (int i = lane; i < BL; i += kWave) cur[i] = f[i];
                cur_depth = depth;
                wave_sync();
            }
            const double* rec = cur + 6 * K + h1 * P_NREC;
            calls++;
            if (rec[P_CAP] != 0.0) { status = OBTG_MD_GJK_CAP; done = true; break; }
            const double t1len = f_t1h - f_t1l;
            const double m1 = f_t1l + f_t1 * t1len;
            const double n1l = h1 ? m1 : f_t1l, n1h = h1 ? f_t1h : m1;
            double lb, t1, nT1, alpha = f_alpha, cx, cy, cz;
            if (rec[P_POS] != 0.0) {
                lb = rec[P_LB]; t1 = rec[P_T1];
                cx = rec[P_CX]; cy = rec[P_CY]; cz = rec[P_CZ];
                const double ub = rec[P_UB], t1loc = rec[P_AM] != 0.0 ? 1.0 : 0.0;
                if (ub <= alpha) { alpha = ub; nT1 = (1 - t1loc) * n1l + t1loc * n1h; }
                else nT1 = -1;
            } else {
                t1 = 0.5; nT1 = -1; cx = cy = cz = -1; lb = p.eps3;
            }
            if (lb >= alpha * (1 - p.eps)) { r0 = alpha; r1 = nT1; rx = cx; ry = cy; rz = cz; OBTG_MD2_RETURN() }
            if (depth + 2 >= p.max_depth) { status = OBTG_MD_DEPTH_CAP; r0 = alpha; r1 = nT1; rx = cx; ry = cy; rz = cz; done = true; break; }
            if (t1 != t1) t1 = 0;
            // expand the child: both its pieces to `nxt`; this frame goes to `scs`, the child's into the registers
            if constexpr (KC > 0) split_both_t<KC>(cur + h1 * 3 * K, cur + h1 * 3 * K, K, t1, t1, nxt, 0, 3, dump);
            else split_one_both(cur + h1 * 3 * K, K, t1, nxt, dump);
            if (depth >= 0) {
                double* sc = depth >= kMdScsLds ? st + (size_t)depth * FRM + BL : scs + depth * G_NSCAL;
                sc[G_T1] = f_t1; sc[G_T1L] = f_t1l; sc[G_T1H] = f_t1h; sc[G_ALPHA] = f_alpha; sc[G_RT1] = f_rt1;
                sc[G_PX] = f_px; sc[G_PY] = f_py; sc[G_PZ] = f_pz; sc[G_STATE] = (double)f_next;
            }
            f_t1 = t1; f_t1l = n1l; f_t1h = n1h; f_alpha = alpha; f_rt1 = nT1; f_px = cx; f_py = cy; f_pz = cz; f_next = 0;
            depth++;
            eval_depth = depth;
            break;
        }
#undef OBTG_MD2_RETURN
    }
    // (every lane, the same values to the same addresses)
    p.res[5 * k] = r0; p.res[5 * k + 1] = r1; p.res[5 * k + 2] = rx; p.res[5 * k + 3] = ry; p.res[5 * k + 4] = rz;
    if (p.info) { p.info[4 * k] = nodes; p.info[4 * k + 1] = calls; p.info[4 * k + 2] = dmax; p.info[4 * k + 3] = status; }
}

int launch_min_dist_robust(obtg_ctx* c, const double* d_curves, int K, const int* d_pa, const int* d_pb, int n_pairs,
                           double eps, int max_nodes, int max_level, int cap, double* d_frontier, double* d_res, int* d_info)
{
    if (n_pairs <= 0) return OBTG_OK;
    if (K < 2 || K > kMdMaxK || cap < 4 || max_level < 1 || max_level > 50) return OBTG_ERR_UNSUPPORTED;
    MdrParams p{ d_curves, d_pa, d_pb, n_pairs, K, max_nodes, max_level, cap, eps, d_frontier, d_res, d_info };
    size_t lds = sizeof(double) * ((size_t)6 * K + (size_t)kWave * ((3 * K) | 1));
    // the specialised control-point counts keep a node's rows in registers (and need only the curves in LDS)
    const char* env_generic = getenv("OBTG_MDR_GENERIC");                  // (read per launch: the A/B test flips it in-process)
    const bool generic = env_generic && env_generic[0] == '1';
    void (*kern)(const MdrParams) = k_min_dist_robust<0>;
    if (!generic) {
        switch (K) {
#define OBTG_CASE(NC_) case NC_: kern = k_min_dist_robust<NC_>; lds = sizeof(double) * 6 * NC_; break;
            OBTG_NC_DYN(OBTG_CASE)
#undef OBTG_CASE
            default: break;
        }
    }
    ScopedKernelTimer t(c, OBTG_K_MIN_DIST);
    hipLaunchKernelGGL(kern, dim3((unsigned)n_pairs), dim3(kWave), lds, c->stream, p);
    OBTG_HIP(c, hipGetLastError());
    return OBTG_OK;
}

int launch_gjk_true_pairs(obtg_ctx* c, const double* d_soa, const int* d_off, const int* d_pa, const int* d_pb, int n_pairs,
                          double eps, int max_iter, int* d_flag, double* d_p1, double* d_p2, double* d_dist, double* d_lower,
                          int* d_iters, int* d_status)
{
    if (n_pairs <= 0) return OBTG_OK;
    GjkTrueParams p{ d_soa, d_off, d_pa, d_pb, n_pairs, max_iter, eps, d_flag, d_p1, d_p2, d_dist, d_lower, d_iters, d_status };
    ScopedKernelTimer t(c, OBTG_K_GJK);
    hipLaunchKernelGGL(k_gjk_true_pairs, dim3((n_pairs + 255) / 256), dim3(256), 0, c->stream, p);
    OBTG_HIP(c, hipGetLastError());
    return OBTG_OK;
}

int launch_min_dist2poly_robust(obtg_ctx* c, const double* d_curves, int K, const double* d_soa, const int* d_off,
                                const int* d_pc, const int* d_pp, int n_pairs, double eps, int max_nodes, int max_level,
                                int cap, int max_poly_K, double* d_frontier, double* d_res, int* d_info)
{
    if (n_pairs <= 0) return OBTG_OK;
    if (K < 2 || K > kMdMaxK || max_poly_K > 4 * kMdMaxK || cap < 2 || max_level < 1 || max_level > 50) return OBTG_ERR_UNSUPPORTED;
    Md2rParams p{ d_curves, d_soa, d_off, d_pc, d_pp, n_pairs, K, max_nodes, max_level, cap, eps, d_frontier, d_res, d_info };
    const size_t lds = sizeof(double) * ((size_t)3 * K + (size_t)3 * max_poly_K + (size_t)kWave * ((3 * K) | 1));
    if (lds > 64 * 1024) return OBTG_ERR_UNSUPPORTED;
    const char* env_generic = getenv("OBTG_MDR_GENERIC");
    void (*kern)(const Md2rParams) = k_min_dist2poly_robust<0>;
    if (!(env_generic && env_generic[0] == '1')) {
        switch (K) {
#define OBTG_CASE(NC_) case NC_: kern = k_min_dist2poly_robust<NC_>; break;
            OBTG_NC_DYN(OBTG_CASE)
#undef OBTG_CASE
            default: break;
        }
    }
    if (lds > 48 * 1024)
        OBTG_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ScopedKernelTimer t(c, OBTG_K_MIN_DIST);
    hipLaunchKernelGGL(kern, dim3((unsigned)n_pairs), dim3(kWave), lds, c->stream, p);
    OBTG_HIP(c, hipGetLastError());
    return OBTG_OK;
}

int launch_min_dist2poly(obtg_ctx* c, const double* d_curves, int K, const double* d_soa,
                         const int* d_off, const int* d_pc, const int* d_pp, int n_pairs, double eps,
                         int max_iter, int md_cap, int max_depth, int max_nodes, double* d_stack,
                         double* d_res, int* d_info, int max_poly_K, bool planar)
{
    if (n_pairs <= 0) return OBTG_OK;
    if (K < 2 || K > kMdMaxK || max_depth < 1) return OBTG_ERR_UNSUPPORTED;
    Md2Params p{ d_curves, d_soa, d_off, d_pc, d_pp, n_pairs, K, max_iter, md_cap, max_depth, max_nodes, eps, cube_as_python(eps),
                 d_stack, d_res, d_info, (int)(min_dist2poly_stack_doubles(K, max_depth) / (size_t)max_depth) };
    ScopedKernelTimer t(c, OBTG_K_MIN_DIST);
    const size_t lds_w = sizeof(double) * ((size_t)6 * K + 8 * kMdMaxK + (size_t)max_depth * G_NSCAL);
    const size_t lds_q = sizeof(double) * ((size_t)2 * md2_quad_blob(K) + 48 + 8 * kMdShRow + 64 +
                                           (size_t)(max_depth < kMdScsLds ? max_depth : kMdScsLds) * G_NSCAL);
    const char* env_form = getenv("OBTG_MD_FORM");          // "wave": a wavefront per gjkNew call (read per launch: the A/B test flips it)
    const char* env_planar = getenv("OBTG_MD_PLANAR");                           // (A/B runs and tests; read per launch)
    const bool no_planar = env_planar && env_planar[0] == '0';
    if (K <= kMdQuadMaxK && max_poly_K <= 16 && lds_q <= 48 * 1024 && !(env_form && !strcmp(env_form, "wave"))) {
        // both children side by side
        const bool pl2 = planar && !no_planar;
        void (*kern)(const Md2Params) = pl2 ? k_min_dist2poly_quad<true, 0> : k_min_dist2poly_quad<false, 0>;
        switch (K) {        // the counts with a build of their own
#define OBTG_CASE(NC_) case NC_: kern = pl2 ? k_min_dist2poly_quad<true, NC_> : k_min_dist2poly_quad<false, NC_>; break;
            OBTG_NC_DYN(OBTG_CASE)
#undef OBTG_CASE
            default: break;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)n_pairs), dim3(kWave), lds_q, c->stream, p);
    }
    else if (lds_w <= 48 * 1024 && max_poly_K <= kMdMaxK)      // one pair per wavefront
        hipLaunchKernelGGL(k_min_dist2poly_wave, dim3((unsigned)n_pairs), dim3(kWave), lds_w, c->stream, p);
    else
        hipLaunchKernelGGL(k_min_dist2poly, dim3((n_pairs + 63) / 64), dim3(64), 0, c->stream, p);
    OBTG_HIP(c, hipGetLastError());
    return OBTG_OK;
}

}  // namespace obtg
