// Device-side restatement of gjkNew / minimumDistance (gjk/gjk.py:230-360, 493-681) as an
// explicit per-lane state machine.  Included only by gjk_kernels.hip, which is compiled with
// -ffp-contract=off: every multiply-add below is unfused unless written as __builtin_fma.
//
// Arithmetic forms that decide branches (Appendix A of SURVEY.md):
//   dot3  = a0*b0 + a1*b1 + a2*b2, left to right      (gjk.py:174-194 `dot`, used by support,
//                                                      weightedOriginToLine)
//   dotb  = fma(a2,b2, fma(a1,b1, a0*b0))             (`ndarray.dot` / np.linalg.norm on 3-vectors:
//                                                      OpenBLAS ddot in the fixture environment)
//   cross = two rounded products and one subtraction per component (np.cross)
// Support selection: strict '>' scanning from index 0, so the lowest index wins ties
// (gjk.py:87-114).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/obtg.h"

// `a**2` of gjk.py:460 as the reference's libm forms it (see libm_pow2.h)
#define OBTG_P2_TABLE static __device__ const
#define OBTG_P2_FUNC __device__ __forceinline__
#include "libm_pow2.h"

namespace obtg {
namespace gjk {

struct V3 { double x, y, z; };

struct Vert {
    V3 v;      // Minkowski point p1 - p2
    int i1, i2;  // support indices into poly1 / poly2 (the reference stores the points)
};

// which dict keys exist (gjk.py:505-526 dispatches on them)
enum : int { kA = 1, kB = 2, kC = 4, kD = 8, kDpts = 16, kColl = 32 };

struct Simplex {
    Vert A, B, C, D;
    int keys;
};

// A point set: coordinate c of point k is mem[base + c*cs + k]; hasz == 0 means z == 0.
struct Poly {
    int base, cs, K, hasz;
};

struct MemGlobal {
    const double* __restrict__ g;
    __device__ __forceinline__ double operator()(int idx) const { return g[idx]; }
};

struct MemLds {
    const double* l;   // points into the kernel's extern __shared__ array
    __device__ __forceinline__ double operator()(int idx) const { return l[idx]; }
};

__device__ __forceinline__ double dot3(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ double dotb(const V3& a, const V3& b)
{
    return __builtin_fma(a.z, b.z, __builtin_fma(a.y, b.y, a.x * b.x));
}
__device__ __forceinline__ V3 cross(const V3& a, const V3& b)
{
    V3 c;
    c.x = a.y * b.z - a.z * b.y;
    c.y = a.z * b.x - a.x * b.z;
    c.z = a.x * b.y - a.y * b.x;
    return c;
}
__device__ __forceinline__ V3 sub(const V3& a, const V3& b) { return V3{ a.x - b.x, a.y - b.y, a.z - b.z }; }
__device__ __forceinline__ V3 neg(const V3& a) { return V3{ -a.x, -a.y, -a.z }; }
__device__ __forceinline__ bool eq(const V3& a, const V3& b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
__device__ __forceinline__ double normb(const V3& a) { return __builtin_sqrt(dotb(a, a)); }

template <class Mem>
__device__ __forceinline__ V3 point(const Mem& mem, const Poly& P, int k)
{
    V3 r;
    r.x = mem(P.base + k);
    r.y = mem(P.base + P.cs + k);
    r.z = P.hasz ? mem(P.base + 2 * P.cs + k) : 0.0;
    return r;
}

// The support scan (gjk.py:87-114 `support` with `dot`, gjk.py:174-194): argmax of
// p.x*d.x + p.y*d.y + p.z*d.z with strict '>' from index 0.
//
// PLANAR: every point of both sets has z == 0 (2-D curves padded as bezier.py:1294-1308 does).
// Then p.z*d.z is +-0 and adding it never changes the value of the two-term sum that the
// comparison sees (x + (+-0) == x for every x != 0, and +-0 compare equal), so the scan may
// drop the z term and the z load without changing any index.  Search directions stay planar
// too: the only out-of-plane direction, +-ABC (gjk.py:620-633), needs ABC.A0 != 0, which is
// impossible when every z is 0.
template <class Mem, bool PLANAR>
__device__ __forceinline__ double sdot(const Mem& mem, const Poly& P, int i, const V3& d)
{
    const double x = mem(P.base + i), y = mem(P.base + P.cs + i);
    if (PLANAR) return x * d.x + y * d.y;
    const double z = P.hasz ? mem(P.base + 2 * P.cs + i) : 0.0;
    return x * d.x + y * d.y + z * d.z;
}

template <class Mem>
struct Ctx {
    Mem mem;
    Poly P1, P2;
    V3 own1, own2;     // support_pts_quarter only: the lane's own point of each set (run_quarter loads them once per call)
    short* trace;      // nullable, [trace_cap][2]
    int trace_cap;
    int n_support;
};

// gjk.py:493-501 supportPts: both scans share one loop so that their loads overlap
template <class Mem, bool PLANAR = false>
__device__ __forceinline__ void support_pts(Ctx<Mem>& g, const V3& dir, Vert& out)
{
    const V3 nd = neg(dir);
    int i1 = 0, i2 = 0;
    double m1 = sdot<Mem, PLANAR>(g.mem, g.P1, 0, dir);
    double m2 = sdot<Mem, PLANAR>(g.mem, g.P2, 0, nd);
    const int K1 = g.P1.K, K2 = g.P2.K, Kmin = K1 < K2 ? K1 : K2;
    int i = 1;
    for (; i < Kmin; ++i) {
        const double c1 = sdot<Mem, PLANAR>(g.mem, g.P1, i, dir);
        const double c2 = sdot<Mem, PLANAR>(g.mem, g.P2, i, nd);
        if (c1 > m1) { m1 = c1; i1 = i; }
        if (c2 > m2) { m2 = c2; i2 = i; }
    }
    for (int j = i; j < K1; ++j) {
        const double c1 = sdot<Mem, PLANAR>(g.mem, g.P1, j, dir);
        if (c1 > m1) { m1 = c1; i1 = j; }
    }
    for (int j = i; j < K2; ++j) {
        const double c2 = sdot<Mem, PLANAR>(g.mem, g.P2, j, nd);
        if (c2 > m2) { m2 = c2; i2 = j; }
    }
    out.i1 = i1; out.i2 = i2;
    out.v = sub(point(g.mem, g.P1, i1), point(g.mem, g.P2, i2));
    if (g.trace && g.n_support < g.trace_cap) {
        g.trace[2 * g.n_support] = (short)i1;
        g.trace[2 * g.n_support + 1] = (short)i2;
    }
    g.n_support++;
}

// supportPts with the scan spread over the wavefront: every lane holds the same direction (the state
// machine runs redundantly in all 64 lanes, with wave-uniform control flow), lanes 0..K1-1 take the
// points of poly1 and lanes 32..32+K2-1 those of poly2 (K <= 32), and the two half-waves reduce to
// "largest value, lowest index among equals" -- which is what the serial scan with its strict '>'
// from index 0 returns (gjk.py:87-114).  The per-point products are the serial scan's own, so the
// indices and the Minkowski vertex are identical to support_pts'.
template <class Mem>
__device__ __forceinline__ void support_pts_wave(Ctx<Mem>& g, const V3& dir, Vert& out)
{
    const int lane = threadIdx.x & 63, half = lane >> 5, li = lane & 31;
    const Poly& P = half ? g.P2 : g.P1;
    const V3 d = half ? neg(dir) : dir;
    const bool have = li < P.K;
    double v = have ? sdot<Mem, false>(g.mem, P, li, d) : -__builtin_inf();
    const double v0 = __shfl(v, half << 5);                // value of point 0 of this half
    if (v != v) v = -__builtin_inf();                      // `cur > maxd` is false for NaN: never selected
    int idx = have ? li : 0x7fffffff;
    // "largest value, lowest index" over the 32 lanes of the half.  The combination is commutative and associative, so any
    // exchange pattern gives the scan's answer; four of the five rounds stay inside a 16-lane row and are DPP moves (quad
    // permutes, then the half-row and the row mirrored: each lane meets the lanes it has not met yet), only the last one
    // crosses rows through the LDS crossbar.  (Round 5: five ds_bpermute rounds of three transfers each were ~0.9 k clocks of
    // a scan, and a pair's search is a serial chain of scans.)
    auto meet = [&](double ov, int oi) {
        const bool take = ov > v || (ov == v && oi < idx);
        v = take ? ov : v;
        idx = take ? oi : idx;
    };
#define OBTG_DPP_MEET(CTRL) \
    { const int lo_ = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false); \
      const int hi_ = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false); \
      const int oi_ = __builtin_amdgcn_update_dpp(0, idx, CTRL, 0xf, 0xf, false); \
      meet(__hiloint2double(hi_, lo_), oi_); }
    OBTG_DPP_MEET(0xB1)      // quad_perm:[1,0,3,2]
    OBTG_DPP_MEET(0x4E)      // quad_perm:[2,3,0,1]
    OBTG_DPP_MEET(0x141)     // row_half_mirror
    OBTG_DPP_MEET(0x140)     // row_mirror
#undef OBTG_DPP_MEET
    meet(__shfl_xor(v, 16), __shfl_xor(idx, 16));
    if (v0 != v0) idx = 0;                                 // maxd starts as NaN: nothing is ever greater
    const int i1 = __shfl(idx, 0), i2 = __shfl(idx, 32);
    out.i1 = i1; out.i2 = i2;
    out.v = sub(point(g.mem, g.P1, i1), point(g.mem, g.P2, i2));
    if (g.trace && g.n_support < g.trace_cap && lane == 0) {
        g.trace[2 * g.n_support] = (short)i1;
        g.trace[2 * g.n_support + 1] = (short)i2;
    }
    g.n_support++;
}

// The largest of a value per lane over the 16 lanes of a row, in every lane of the row: two DPP moves and one v_max_f64 per
// exchange (quad permutes, half row mirrored, row mirrored).  v_max_f64 is IEEE maxNum: a quiet NaN operand loses against a
// number (the products below are arithmetic results, so their NaNs are quiet); all NaN gives NaN.  (The instruction is written
// out: through fmax() the compiler first canonicalises the moved operand -- a second v_max_f64 per exchange -- and through
// update_dpp it zeroes the destination registers first: 6 instructions per exchange where 3 do.)
__device__ __forceinline__ double row_max(double v)
{
#define OBTG_DPP_MAX(CTRL) \
    { const int lo_ = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true); \
      const int hi_ = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true); \
      const double o_ = __hiloint2double(hi_, lo_); \
      asm("v_max_f64 %0, %1, %2" : "=v"(v) : "v"(v), "v"(o_)); }
    OBTG_DPP_MAX(0xB1)
    OBTG_DPP_MAX(0x4E)
    OBTG_DPP_MAX(0x141)
    OBTG_DPP_MAX(0x140)
#undef OBTG_DPP_MAX
    return v;
}

// "largest value, lowest index" of a row (the serial scan with its strict '>' from index 0, gjk.py:87-114) as: the row's
// largest value, then the lowest lane of the row that holds it -- a ballot and a find-first-set instead of carrying the index
// through the exchanges.  +0 and -0 compare equal here as they do under '>'; lanes past the set hold -inf and sit above the
// set's lanes, so they never win a tie.  NaN: `cur > maxd` is false for a NaN `cur` -- never selected: v_max_f64 drops it and
// `v == m` is false for it -- and for a NaN `maxd`, which the scan starts with when point 0's value is one -- nothing is ever
// greater: the caller's `nan0` (the row's lane 0 holds point 0), which also covers the all-NaN row (m is NaN, no lane equal).
__device__ __forceinline__ int row_argmax(double v, bool nan0)
{
    const int lane = threadIdx.x & 63;
    const double m = row_max(v);
    const unsigned bits = (unsigned)(__ballot(v == m) >> (lane & 48)) & 0xffffu;
    return nan0 ? 0 : __ffs((int)bits) - 1;
}

// supportPts for FOUR calls at a time, one per 16-lane row of the wavefront (K <= 16 points per set): the lanes of a row
// hold that row's call (its own two point sets, direction and simplex), lane l of the row takes point l of BOTH sets, and
// the two "largest value, lowest index" reductions run side by side over the row -- quad permutes, the half row mirrored,
// the row mirrored: all DPP moves, after which every lane of the row holds both answers (no cross-row transfer at all).
// Same products, same tie rule as the serial scan (see support_pts_wave).
template <class Mem>
__device__ __forceinline__ void support_pts_quarter(Ctx<Mem>& g, const V3& dir, Vert& out)
{
    const int lane = threadIdx.x & 63, l = lane & 15;
    const V3 nd = neg(dir);
    const bool have1 = l < g.P1.K, have2 = l < g.P2.K;
    // (sdot's expression on the lane's own points, which do not change during a call)
    double v1 = have1 ? g.own1.x * dir.x + g.own1.y * dir.y + g.own1.z * dir.z : -__builtin_inf();
    double v2 = have2 ? g.own2.x * nd.x + g.own2.y * nd.y + g.own2.z * nd.z : -__builtin_inf();
    // point 0's value NaN: maxd starts as NaN and nothing is ever greater (the row's lane 0 holds point 0)
    const unsigned long long nan1 = __ballot(v1 != v1), nan2 = __ballot(v2 != v2);
    const int row0 = lane & 48;
    const int i1 = row_argmax(v1, (nan1 >> row0) & 1), i2 = row_argmax(v2, (nan2 >> row0) & 1);
    out.i1 = i1; out.i2 = i2;
    out.v = sub(point(g.mem, g.P1, i1), point(g.mem, g.P2, i2));
    g.n_support++;
}

// gjk.py:397-437 weightedOriginToLine
__device__ __forceinline__ double origin_to_line(const V3& A, const V3& B, double& dist)
{
    if (eq(A, B)) { dist = __builtin_sqrt(dot3(A, A)); return 0.0; }
    const V3 v = sub(B, A);
    double t = -dot3(v, A) / dot3(v, v);
    if (t > 1) t = 1; else if (t < 0) t = 0;
    V3 cp;
    cp.x = (1 - t) * A.x + t * B.x;
    cp.y = (1 - t) * A.y + t * B.y;
    cp.z = (1 - t) * A.z + t * B.z;
    dist = __builtin_sqrt(dot3(cp, cp));
    return t;
}

// gjk.py:565-642 simplex3pt WITHOUT its trailing supportPts (every branch ends in one, the
// caller issues it once so that the scan is convergent code for the whole wave)
__device__ __forceinline__ void simplex3_update(Simplex& s, V3& dir)
{
    const V3 A0 = neg(s.A.v), AB = sub(s.B.v, s.A.v), AC = sub(s.C.v, s.A.v);
    const V3 ABC = cross(AB, AC);
    if (dotb(cross(ABC, AC), A0) > 0) {
        if (dotb(AC, A0) > 0) {
            dir = cross(cross(AC, A0), AC);
            s.B = s.A;
        } else if (dotb(AB, A0) > 0) {
            dir = cross(cross(AB, A0), AB);
            s.C = s.A;
        } else {
            dir = s.A.v;              // +A (gjk.py:595)
            s.keys = 0;
        }
    } else if (dotb(cross(AB, ABC), A0) > 0) {
        if (dotb(AB, A0) > 0) {
            dir = cross(cross(AB, A0), AB);
            s.C = s.A;
        } else {
            dir = neg(s.A.v);
            s.keys = 0;
        }
    } else {
        const double h = dotb(ABC, A0);
        if (h == 0) {
            s.keys |= kColl;
            dir = V3{ 0.0, 0.0, 0.0 };
        } else if (h > 0) {
            dir = ABC;
            s.D = s.C; s.C = s.B; s.B = s.A;
            s.keys |= kD | kDpts;
        } else {
            dir = neg(ABC);
            s.D = s.B; s.B = s.A;
            s.keys |= kD | kDpts;
        }
    }
}

// gjk.py:505-561, 646-681 doSimplex minus the supportPts call; returns whether the case
// fetches a new support vertex (everything except the 4-point collision exit does)
__device__ __forceinline__ bool simplex_update(Simplex& s, V3& dir)
{
    if (!(s.keys & kA)) return true;
    if (!(s.keys & kB)) {
        s.B = s.A; s.keys |= kB;
        dir = neg(dir);
        return true;
    }
    if (!(s.keys & kC)) {
        double dist;
        const double t = origin_to_line(s.A.v, s.B.v, dist);
        dir.x = -((1 - t) * s.A.v.x + t * s.B.v.x);
        dir.y = -((1 - t) * s.A.v.y + t * s.B.v.y);
        dir.z = -((1 - t) * s.A.v.z + t * s.B.v.z);
        s.C = s.A; s.keys |= kC;
        return true;
    }
    if (s.keys & kD) {
        // four points: pick the face the origin is in front of (gjk.py:646-681), then the three-point case below --
        // ONE copy of it for both kinds of lanes (a wave usually holds both)
        const V3 A0 = neg(s.A.v), AB = sub(s.B.v, s.A.v), AC = sub(s.C.v, s.A.v), AD = sub(s.D.v, s.A.v);
        const V3 ABC = cross(AB, AC), ACD = cross(AC, AD), ADB = cross(AD, AB);
        if (dotb(ABC, A0) > 0) {
            s.keys &= ~kD;                       // pop('D') only; 'Dpts' stays (gjk.py:660)
        } else if (dotb(ACD, A0) > 0) {
            s.B = s.C; s.C = s.D; s.keys &= ~(kD | kDpts);
        } else if (dotb(ADB, A0) > 0) {
            s.C = s.B; s.B = s.D; s.keys &= ~(kD | kDpts);
        } else {
            s.keys |= kColl;
            dir = V3{ 0.0, 0.0, 0.0 };
            return false;
        }
    }
    simplex3_update(s, dir);
    return true;
}

// WAVE: 0 = a call per lane, 1 = one call per wavefront (support_pts_wave), 2 = one call per 16-lane row (support_pts_quarter)
template <class Mem, bool PLANAR = false, int WAVE = 0>
__device__ __forceinline__ void do_simplex(Ctx<Mem>& g, Simplex& s, V3& dir)
{
    if (simplex_update(s, dir)) {
        if (WAVE == 2) support_pts_quarter<Mem>(g, dir, s.A);
        else if (WAVE == 1) support_pts_wave<Mem>(g, dir, s.A);
        else support_pts<Mem, PLANAR>(g, dir, s.A);
        s.keys |= kA;
    }
}

// `(simplex['A'] == point).all()` against every value of the old dict (gjk.py:281-294)
template <class Mem>
__device__ __forceinline__ bool vert_matches(const Ctx<Mem>& g, const Vert& o, bool has, bool haspts, const V3& A)
{
    if (has && eq(A, o.v)) return true;
    if (haspts && eq(A, point(g.mem, g.P1, o.i1)) && eq(A, point(g.mem, g.P2, o.i2))) return true;
    return false;
}

template <class Mem>
__device__ __forceinline__ bool matches_old(const Ctx<Mem>& g, const Simplex& o, const V3& A)
{
    if (vert_matches(g, o.A, o.keys & kA, o.keys & kA, A)) return true;
    if (vert_matches(g, o.B, o.keys & kB, o.keys & kB, A)) return true;
    if (vert_matches(g, o.C, o.keys & kC, o.keys & kC, A)) return true;
    if (vert_matches(g, o.D, o.keys & kD, o.keys & kDpts, A)) return true;
    if ((o.keys & kColl) && A.x == 1.0 && A.y == 1.0 && A.z == 1.0) return true;
    return false;
}

// matches_old with ONE LDS round trip in the usual case: a vertex's "pts" clause needs A == its poly1 point, whose first test
// is A.x == that point's x -- the four x are fetched together, and only if one of them is equal (A is a Minkowski difference:
// hardly ever) the clause-by-clause form above decides.  Same truth value in every case.
template <class Mem>
__device__ __forceinline__ bool matches_old_batched(const Ctx<Mem>& g, const Simplex& o, const V3& A)
{
    const bool hA = o.keys & kA, hB = o.keys & kB, hC = o.keys & kC, hD = o.keys & kD, pD = o.keys & kDpts;
    // (the indices of absent entries are zero or left from an earlier round of the SAME pair -- the sweeps reset them when a lane
    // takes a new pair --, so always in range of this pair's first hull)
    const double xa = g.mem(g.P1.base + o.A.i1), xb = g.mem(g.P1.base + o.B.i1), xc = g.mem(g.P1.base + o.C.i1),
                 xd = g.mem(g.P1.base + o.D.i1);
    const bool maybe = (hA & (A.x == xa)) | (hB & (A.x == xb)) | (hC & (A.x == xc)) | (pD & (A.x == xd));
    if (maybe) return matches_old(g, o, A);
    return (hA && eq(A, o.A.v)) || (hB && eq(A, o.B.v)) || (hC && eq(A, o.C.v)) || (hD && eq(A, o.D.v)) ||
           ((o.keys & kColl) && A.x == 1.0 && A.y == 1.0 && A.z == 1.0);
}

// Brent cycle detector for minimumDistance's `while True` (gjk.py:277): the loop body is a function
// of (live simplex entries, search direction) alone, so an exact return to a checkpointed state
// proves that the reference never exits.  One checkpoint, refreshed after 1, 2, 4, ... rounds.
struct Checkpoint {
    int keys, power, lam;
    int a1, a2, b1, b2, c1, c2, d1, d2;
    V3 dir;
    __device__ __forceinline__ void take(const Simplex& s, const V3& d)
    {
        keys = s.keys; dir = d;
        a1 = s.A.i1; a2 = s.A.i2; b1 = s.B.i1; b2 = s.B.i2;
        c1 = s.C.i1; c2 = s.C.i2; d1 = s.D.i1; d2 = s.D.i2;
    }
    __device__ __forceinline__ void start(const Simplex& s, const V3& d) { take(s, d); power = 1; lam = 0; }
    __device__ __forceinline__ bool same(const Simplex& s, const V3& d) const
    {
        if (s.keys != keys || !eq(d, dir)) return false;
        if ((keys & kA) && (s.A.i1 != a1 || s.A.i2 != a2)) return false;
        if ((keys & kB) && (s.B.i1 != b1 || s.B.i2 != b2)) return false;
        if ((keys & kC) && (s.C.i1 != c1 || s.C.i2 != c2)) return false;
        if ((keys & (kD | kDpts)) && (s.D.i1 != d1 || s.D.i2 != d2)) return false;
        return true;
    }
    // after one round that did not converge: true = the state repeats
    __device__ __forceinline__ bool step(const Simplex& s, const V3& d)
    {
        if (same(s, d)) return true;
        if (++lam == power) { take(s, d); power *= 2; lam = 0; }
        return false;
    }
};

struct Result {
    int flag, status, n_support;
    V3 c1, c2;
    double dist;
};

template <class Mem>
__device__ __forceinline__ void seg_result(const Ctx<Mem>& g, const Vert& A, const Vert& O, Result& r)
{
    const double t = origin_to_line(A.v, O.v, r.dist);
    const V3 a1 = point(g.mem, g.P1, A.i1), o1 = point(g.mem, g.P1, O.i1);
    const V3 a2 = point(g.mem, g.P2, A.i2), o2 = point(g.mem, g.P2, O.i2);
    r.c1 = V3{ (1 - t) * a1.x + t * o1.x, (1 - t) * a1.y + t * o1.y, (1 - t) * a1.z + t * o1.z };
    r.c2 = V3{ (1 - t) * a2.x + t * o2.x, (1 - t) * a2.y + t * o2.y, (1 - t) * a2.z + t * o2.z };
}

// gjk.py:299-360: closest points / distance from the converged (restored) simplex
template <class Mem>
__device__ __forceinline__ void closest_from_simplex(const Ctx<Mem>& g, const Simplex& s, Result& r);

// gjk.py:230-270 gjkNew + 273-360 minimumDistance
// WAVE: one pair per wavefront, see support_pts_wave
template <class Mem, bool PLANAR = false, int WAVE = 0>
__device__ __forceinline__ void run(Ctx<Mem>& g, int max_iter, int md_cap, Result& r)
{
    Simplex s;
    s.keys = 0;
    s.A = Vert{ V3{ 0, 0, 0 }, 0, 0 };
    s.B = s.A; s.C = s.A; s.D = s.A;
    V3 dir{ 1.0, 0.0, 0.0 };
    const double qnan = __builtin_nan("");
    r.flag = -1; r.status = OBTG_ST_MAXITER;
    r.c1 = V3{ qnan, qnan, qnan }; r.c2 = r.c1; r.dist = qnan;
    for (int it = 0; it < max_iter; ++it) {
        do_simplex<Mem, PLANAR, WAVE>(g, s, dir);
        if (s.keys & kColl) { r.flag = 0; r.status = OBTG_ST_OK; break; }
        if (dotb(s.A.v, dir) < 0) {
            Simplex old = s;
            Checkpoint chk;
            chk.start(s, dir);
            bool conv = false, cycle = false;
            for (int rr = 0; rr < md_cap; ++rr) {
                old = s;
                do_simplex<Mem, PLANAR, WAVE>(g, s, dir);
                if (matches_old_batched(g, old, s.A.v)) { conv = true; break; }
                if (chk.step(s, dir)) { cycle = true; break; }
            }
            r.flag = 1;
            if (!conv) { r.status = cycle ? OBTG_ST_CYCLE : OBTG_ST_MD_CAP; break; }
            r.status = OBTG_ST_OK;
            closest_from_simplex(g, old, r);
            break;
        }
    }
    r.n_support = g.n_support;
}

// gjkNew + minimumDistance for four calls at a time, one per 16-lane row (support_pts_quarter).  `run` above has two loops
// -- the sign search of gjkNew and, entered from inside it, minimumDistance's `while True` -- and rows that are in different
// loops would take turns (a row inside the inner loop runs it to the end while the others wait).  Here the two are ONE loop
// with a phase per row, so that every trip is one doSimplex of every live row: the four calls cost about as many trips as
// the longest of them.  Per row the sequence of operations is `run`'s, hence the same bits.
template <class Mem>
__device__ __forceinline__ void run_quarter(Ctx<Mem>& g, int max_iter, int md_cap, Result& r)
{
    Simplex s;
    s.keys = 0;
    s.A = Vert{ V3{ 0, 0, 0 }, 0, 0 };
    s.B = s.A; s.C = s.A; s.D = s.A;
    Simplex old = s;
    Checkpoint chk;
    chk.start(s, V3{ 0, 0, 0 });
    V3 dir{ 1.0, 0.0, 0.0 };
    const double qnan = __builtin_nan("");
    r.flag = -1; r.status = OBTG_ST_MAXITER;
    r.c1 = V3{ qnan, qnan, qnan }; r.c2 = r.c1; r.dist = qnan;
    {
        const int l = threadIdx.x & 15;
        g.own1 = point(g.mem, g.P1, l < g.P1.K ? l : 0);
        g.own2 = point(g.mem, g.P2, l < g.P2.K ? l : 0);
    }
    int phase = 0, it = 0, rr = 0;
    bool live = max_iter > 0, conv = false;
    while (live) {
        if (phase) old = s;
        do_simplex<Mem, false, 2>(g, s, dir);
        if (!phase) {
            if (s.keys & kColl) { r.flag = 0; r.status = OBTG_ST_OK; live = false; }
            else if (dotb(s.A.v, dir) < 0) { phase = 1; chk.start(s, dir); }
            else if (++it >= max_iter) live = false;
        } else {
            if (matches_old_batched(g, old, s.A.v)) { conv = true; live = false; }
            else if (chk.step(s, dir)) { r.flag = 1; r.status = OBTG_ST_CYCLE; live = false; }
            else if (++rr >= md_cap) { r.flag = 1; r.status = OBTG_ST_MD_CAP; live = false; }
        }
    }
    if (conv) {
        r.flag = 1; r.status = OBTG_ST_OK;
        closest_from_simplex(g, old, r);
    }
    r.n_support = g.n_support;
}

template <class Mem>
__device__ __forceinline__ void closest_from_simplex(const Ctx<Mem>& g, const Simplex& s, Result& r)
{
    if (s.keys & kC) {
        const V3 A0 = neg(s.A.v), AB = sub(s.B.v, s.A.v), AC = sub(s.C.v, s.A.v);
        const V3 ABC = cross(AB, AC);
        if (dotb(cross(ABC, AC), A0) >= 0) {
            seg_result(g, s.A, s.C, r);
        } else if (dotb(cross(AB, ABC), A0) >= 0) {
            seg_result(g, s.A, s.B, r);
        } else {
            // gjk.py:440-477 weightedOriginToPlane; `a**2` (gjk.py:460) is libm's pow(a, 2.0), one ulp away from a * a now and
            // then: restated in libm_pow2.h, so that closest points and distance are the reference's to the bit
            const V3 N = cross(sub(s.B.v, s.A.v), sub(s.C.v, s.A.v));
            const double nn = normb(N);
            const V3 n{ N.x / nn, N.y / nn, N.z / nn };
            const double tq = (n.x * s.A.v.x + n.y * s.A.v.y + n.z * s.A.v.z) /
#ifdef OBTG_SQUARE_AS_PRODUCT      // (A/B builds only: rounds 1-4's a * a)
                              (n.x * n.x + n.y * n.y + n.z * n.z);
#else
                              (obtg_square_as_libm_pow(n.x) + obtg_square_as_libm_pow(n.y) + obtg_square_as_libm_pow(n.z));
#endif
            const V3 cp{ tq * n.x, tq * n.y, tq * n.z };
            r.dist = __builtin_sqrt(dot3(cp, cp));
            const V3 PA = sub(s.A.v, cp), PB = sub(s.B.v, cp), PC = sub(s.C.v, cp);
            const double al = normb(cross(PB, PC)) / nn;
            const double be = normb(cross(PC, PA)) / nn;
            const double ga = 1 - al - be;
            const V3 a1 = point(g.mem, g.P1, s.A.i1), b1 = point(g.mem, g.P1, s.B.i1),
                     c1 = point(g.mem, g.P1, s.C.i1);
            const V3 a2 = point(g.mem, g.P2, s.A.i2), b2 = point(g.mem, g.P2, s.B.i2),
                     c2 = point(g.mem, g.P2, s.C.i2);
            r.c1.x = (al * (s.A.v.x + a2.x) + be * (s.B.v.x + b2.x)) + ga * (s.C.v.x + c2.x);
            r.c1.y = (al * (s.A.v.y + a2.y) + be * (s.B.v.y + b2.y)) + ga * (s.C.v.y + c2.y);
            r.c1.z = (al * (s.A.v.z + a2.z) + be * (s.B.v.z + b2.z)) + ga * (s.C.v.z + c2.z);
            r.c2.x = (al * (a1.x - s.A.v.x) + be * (b1.x - s.B.v.x)) + ga * (c1.x - s.C.v.x);
            r.c2.y = (al * (a1.y - s.A.v.y) + be * (b1.y - s.B.v.y)) + ga * (c1.y - s.C.v.y);
            r.c2.z = (al * (a1.z - s.A.v.z) + be * (b1.z - s.B.v.z)) + ga * (c1.z - s.C.v.z);
        }
    } else if (s.keys & kB) {
        seg_result(g, s.A, s.B, r);
    } else {
        r.dist = normb(s.A.v);
        r.c1 = point(g.mem, g.P1, s.A.i1);
        r.c2 = point(g.mem, g.P2, s.A.i2);
    }
}


// =====================================================================================
//  Planar state machine.  When every point of both sets has z == 0 the 3-D arithmetic of
//  gjk.py degenerates: every cross product is either (+-0, +-0, w) or lies in the plane, the
//  z terms of every dot product are +-0, and ABC.A0 is always 0, so the tetrahedron cases are
//  unreachable and "on the ABC plane" (gjk.py:616-618) is the only exit of the third branch.
//  The formulas below are the surviving non-zero terms of the reference's expressions, in the
//  reference's operation order: results are identical up to the sign of exact zeros, which no
//  comparison observes.
//    cz(a,b)            = a.x*b.y - a.y*b.x              (z of np.cross)
//    cross((0,0,w), v)  = (-(w*v.y),  w*v.x)
//    cross(v, (0,0,w))  = (  v.y*w , -(v.x*w))
//    dotb(a,b)          = fma(a.y,b.y, a.x*b.x);   dot(a,b) = a.x*b.x + a.y*b.y
// =====================================================================================
struct V2 { double x, y; };
// support indices packed i1 | i2 << 16 (one register per vertex: the planar sweeps copy simplices every step)
struct Vert2 {
    V2 v;
    int ii;
    __device__ __forceinline__ int i1() const { return ii & 0xffff; }
    __device__ __forceinline__ int i2() const { return (int)((unsigned)ii >> 16); }
};
__device__ __forceinline__ int pack_ii(int i1, int i2) { return i1 | (i2 << 16); }
struct Simplex2 { Vert2 A, B, C; int keys; };

__device__ __forceinline__ double cz(const V2& a, const V2& b) { return a.x * b.y - a.y * b.x; }
__device__ __forceinline__ double dotb2(const V2& a, const V2& b) { return __builtin_fma(a.y, b.y, a.x * b.x); }
__device__ __forceinline__ double dot2(const V2& a, const V2& b) { return a.x * b.x + a.y * b.y; }
__device__ __forceinline__ V2 sub2(const V2& a, const V2& b) { return V2{ a.x - b.x, a.y - b.y }; }
__device__ __forceinline__ V2 neg2(const V2& a) { return V2{ -a.x, -a.y }; }
__device__ __forceinline__ bool eq2(const V2& a, const V2& b) { return a.x == b.x && a.y == b.y; }

template <class Mem>
__device__ __forceinline__ V2 point2(const Mem& mem, const Poly& P, int k)
{
    return V2{ mem(P.base + k), mem(P.base + P.cs + k) };
}

template <class Mem>
__device__ __forceinline__ void support_pts2(Ctx<Mem>& g, const V2& dir, Vert2& out)
{
    const V3 d{ dir.x, dir.y, 0.0 };
    const V3 nd{ -dir.x, -dir.y, 0.0 };
    int i1 = 0, i2 = 0;
    double m1 = sdot<Mem, true>(g.mem, g.P1, 0, d);
    double m2 = sdot<Mem, true>(g.mem, g.P2, 0, nd);
    const int K1 = g.P1.K, K2 = g.P2.K, Kmin = K1 < K2 ? K1 : K2;
    int i = 1;
    for (; i < Kmin; ++i) {
        const double c1 = sdot<Mem, true>(g.mem, g.P1, i, d);
        const double c2 = sdot<Mem, true>(g.mem, g.P2, i, nd);
        if (c1 > m1) { m1 = c1; i1 = i; }
        if (c2 > m2) { m2 = c2; i2 = i; }
    }
    for (int j = i; j < K1; ++j) {
        const double c1 = sdot<Mem, true>(g.mem, g.P1, j, d);
        if (c1 > m1) { m1 = c1; i1 = j; }
    }
    for (int j = i; j < K2; ++j) {
        const double c2 = sdot<Mem, true>(g.mem, g.P2, j, nd);
        if (c2 > m2) { m2 = c2; i2 = j; }
    }
    out.ii = pack_ii(i1, i2);
    out.v = sub2(point2(g.mem, g.P1, i1), point2(g.mem, g.P2, i2));
    if (g.trace && g.n_support < g.trace_cap) {
        g.trace[2 * g.n_support] = (short)i1;
        g.trace[2 * g.n_support + 1] = (short)i2;
    }
    g.n_support++;
}

// doSimplex (gjk.py:505-642) for planar inputs, without the trailing supportPts.
// Every case fetches a support afterwards (the 4-point case cannot occur).
__device__ __forceinline__ void simplex_update2(Simplex2& s, V2& dir)
{
    if (!(s.keys & kA)) {
        // 0pt: keep direction
    } else if (!(s.keys & kB)) {
        s.B = s.A; s.keys |= kB;
        dir = neg2(dir);
    } else if (!(s.keys & kC)) {
        // weightedOriginToLine (gjk.py:397-437)
        double t = 0.0;
        if (!eq2(s.A.v, s.B.v)) {
            const V2 v = sub2(s.B.v, s.A.v);
            t = -dot2(v, s.A.v) / dot2(v, v);
            if (t > 1) t = 1; else if (t < 0) t = 0;
        }
        dir.x = -((1 - t) * s.A.v.x + t * s.B.v.x);
        dir.y = -((1 - t) * s.A.v.y + t * s.B.v.y);
        s.C = s.A; s.keys |= kC;
    } else {
        const V2 A0 = neg2(s.A.v), AB = sub2(s.B.v, s.A.v), AC = sub2(s.C.v, s.A.v);
        const double w = cz(AB, AC);                                   // ABC = (0, 0, w)
        const V2 t1{ -(w * AC.y), w * AC.x };                          // ABC x AC
        const V2 t2{ AB.y * w, -(AB.x * w) };                          // AB x ABC
        const double uab = cz(AB, A0);
        const V2 dAB{ -(uab * AB.y), uab * AB.x };                     // (AB x A0) x AB
        const bool ab_pos = dotb2(AB, A0) > 0;
        if (dotb2(t1, A0) > 0) {
            if (dotb2(AC, A0) > 0) {
                const double u = cz(AC, A0);
                dir = V2{ -(u * AC.y), u * AC.x };                     // (AC x A0) x AC
                s.B = s.A;
            } else if (ab_pos) {
                dir = dAB; s.C = s.A;
            } else {
                dir = s.A.v; s.keys = 0;                               // +A (gjk.py:595)
            }
        } else if (dotb2(t2, A0) > 0) {
            if (ab_pos) { dir = dAB; s.C = s.A; }
            else { dir = neg2(s.A.v); s.keys = 0; }
        } else {
            s.keys |= kColl;                                           // ABC.A0 == 0 always
            dir = V2{ 0.0, 0.0 };
        }
    }
}

template <class Mem>
__device__ __forceinline__ bool vert_matches2(const Ctx<Mem>& g, const Vert2& o, bool has, const V2& A)
{
    if (!has) return false;
    if (eq2(A, o.v)) return true;
    // A == p1 and A == p2 forces p1 == p2, i.e. the old vertex p1 - p2 is exactly zero: test that
    // in registers first, the point loads are then needed (almost) never
    if (o.v.x != 0.0 || o.v.y != 0.0) return false;
    return eq2(A, point2(g.mem, g.P1, o.i1())) && eq2(A, point2(g.mem, g.P2, o.i2()));
}

template <class Mem>
__device__ __forceinline__ bool matches_old2(const Ctx<Mem>& g, const Simplex2& o, const V2& A)
{
    if (vert_matches2(g, o.A, o.keys & kA, A)) return true;
    if (vert_matches2(g, o.B, o.keys & kB, A)) return true;
    if (vert_matches2(g, o.C, o.keys & kC, A)) return true;
    return false;   // a 'collision': True value would need A == (1,1,1): impossible with z == 0
}

__device__ __forceinline__ Simplex lift(const Simplex2& s)
{
    Simplex r;
    r.A = Vert{ V3{ s.A.v.x, s.A.v.y, 0.0 }, s.A.i1(), s.A.i2() };
    r.B = Vert{ V3{ s.B.v.x, s.B.v.y, 0.0 }, s.B.i1(), s.B.i2() };
    r.C = Vert{ V3{ s.C.v.x, s.C.v.y, 0.0 }, s.C.i1(), s.C.i2() };
    r.D = r.A;
    r.keys = s.keys;
    return r;
}

// gjk.py:299-360 on a planar simplex: which feature is closest -- point A, segment A-B, segment A-C, or the triangle's
// interior -- decided with the planar forms of closest_from_simplex's expressions (the ones simplex_update2 uses), then ONE
// segment evaluation (weightedOriginToLine, gjk.py:397-437, dot's two-term sums; np.linalg.norm(A) = sqrt(dotb(A, A)) for the
// point).  The interior case (about one call in a hundred) takes the general evaluation on the lifted simplex.  The planar sweeps'
// phase 2 has evaluated their records this way since round 1; same values as closest_from_simplex(lift(s)).
template <class Mem>
__device__ __forceinline__ void closest_from_simplex2(const Ctx<Mem>& g, const Simplex2& s, Result& r)
{
    const V2 a1 = point2(g.mem, g.P1, s.A.i1()), a2 = point2(g.mem, g.P2, s.A.i2());
    const V2 A = s.A.v;
    int which = 0;                    // 0: point A, 1: segment A-B, 2: segment A-C, 3: the triangle's interior
    if (s.keys & kC) {
        const V2 A0 = neg2(A), AB = sub2(s.B.v, A), AC = sub2(s.C.v, A);
        const double w = cz(AB, AC);
        const V2 t1{ -(w * AC.y), w * AC.x };
        const V2 t2{ AB.y * w, -(AB.x * w) };
        which = (dotb2(t1, A0) >= 0) ? 2 : ((dotb2(t2, A0) >= 0) ? 1 : 3);
    } else if (s.keys & kB) which = 1;
    if (which == 3) { closest_from_simplex(g, lift(s), r); return; }
    double t = 0.0, rad = dotb2(A, A);
    V2 o1 = a1, o2 = a2;
    if (which != 0) {
        const Vert2& O = which == 2 ? s.C : s.B;
        o1 = point2(g.mem, g.P1, O.i1()); o2 = point2(g.mem, g.P2, O.i2());
        rad = dot2(A, A);                                           // identical points (gjk.py:417-419)
        if (!eq2(A, O.v)) {
            const V2 v = sub2(O.v, A);
            t = -dot2(v, A) / dot2(v, v);
            if (t > 1) t = 1; else if (t < 0) t = 0;
            const V2 cp{ (1 - t) * A.x + t * O.v.x, (1 - t) * A.y + t * O.v.y };
            rad = dot2(cp, cp);
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
}

// ---- the planar machine for FOUR calls at a time, one per 16-lane row (round 6: `_minDist` on 2-D curves, which is every
//      driver that uses it -- Examples/ComplexObstacles.py, DrivingOnATrack.py; bezier.py:1294-1308 pads them with z = 0).
//      run_quarter spends its trips on the 3-D expressions although, with every z zero, two thirds of their terms are exact
//      zeros and the tetrahedron cases cannot occur; this is run_quarter on simplex_update2 / matches_old2: per row the
//      operations that survive, in the reference's order, hence run_quarter's bits on such inputs (the planar sweeps have
//      held simplex_update2 to the 3-D machine and to the reference's traces since round 1).

// supportPts of a row's planar call (see support_pts_quarter; sdot<PLANAR>'s two-term products)
template <class Mem>
__device__ __forceinline__ void support_pts_quarter2(Ctx<Mem>& g, const V2& dir, Vert2& out)
{
    const int lane = threadIdx.x & 63, l = lane & 15;
    const V2 nd = neg2(dir);
    const bool have1 = l < g.P1.K, have2 = l < g.P2.K;
    double v1 = have1 ? g.own1.x * dir.x + g.own1.y * dir.y : -__builtin_inf();
    double v2 = have2 ? g.own2.x * nd.x + g.own2.y * nd.y : -__builtin_inf();
    // point 0's value NaN: maxd starts as NaN and nothing is ever greater (the row's lane 0 holds point 0)
    const unsigned long long nan1 = __ballot(v1 != v1), nan2 = __ballot(v2 != v2);
    const int row0 = lane & 48;
    const int i1 = row_argmax(v1, (nan1 >> row0) & 1), i2 = row_argmax(v2, (nan2 >> row0) & 1);
    out.ii = pack_ii(i1, i2);
    out.v = sub2(point2(g.mem, g.P1, i1), point2(g.mem, g.P2, i2));
    g.n_support++;
}

// Checkpoint for planar states: (live simplex entries, direction); the kD / kDpts entries of the 3-D state never exist here
// and a direction's z is an exact zero, so `same` is true exactly when Checkpoint::same is on the lifted state -- the cycle
// is reported at the same round.
struct Checkpoint2 {
    int keys, power, lam, a, b, c;
    V2 dir;
    __device__ __forceinline__ void take(const Simplex2& s, const V2& d) { keys = s.keys; dir = d; a = s.A.ii; b = s.B.ii; c = s.C.ii; }
    __device__ __forceinline__ void start(const Simplex2& s, const V2& d) { take(s, d); power = 1; lam = 0; }
    __device__ __forceinline__ bool same(const Simplex2& s, const V2& d) const
    {
        if (s.keys != keys || !eq2(d, dir)) return false;
        if ((keys & kA) && s.A.ii != a) return false;
        if ((keys & kB) && s.B.ii != b) return false;
        if ((keys & kC) && s.C.ii != c) return false;
        return true;
    }
    __device__ __forceinline__ bool step(const Simplex2& s, const V2& d)
    {
        if (same(s, d)) return true;
        if (++lam == power) { take(s, d); power *= 2; lam = 0; }
        return false;
    }
};

// run_quarter for point sets whose z are all zero (the caller has checked)
template <class Mem>
__device__ __forceinline__ void run_quarter2(Ctx<Mem>& g, int max_iter, int md_cap, Result& r)
{
    Simplex2 s;
    s.keys = 0;
    s.A = Vert2{ V2{ 0, 0 }, 0 };
    s.B = s.A; s.C = s.A;
    Simplex2 old = s;
    Checkpoint2 chk;
    chk.start(s, V2{ 0, 0 });
    V2 dir{ 1.0, 0.0 };
    const double qnan = __builtin_nan("");
    r.flag = -1; r.status = OBTG_ST_MAXITER;
    r.c1 = V3{ qnan, qnan, qnan }; r.c2 = r.c1; r.dist = qnan;
    {
        const int l = threadIdx.x & 15;
        const V2 p1 = point2(g.mem, g.P1, l < g.P1.K ? l : 0), p2 = point2(g.mem, g.P2, l < g.P2.K ? l : 0);
        g.own1 = V3{ p1.x, p1.y, 0.0 };
        g.own2 = V3{ p2.x, p2.y, 0.0 };
    }
    int phase = 0, it = 0, rr = 0;
    bool live = max_iter > 0, conv = false;
    // ---- the first two doSimplex steps as straight-line code.  They always happen (unless max_iter stops the call after
    //      one) and always with the directions (1, 0) and (-1, -0) (gjk.py:247, 544): zero points -- keep the direction,
    //      fetch A --, then one point -- B <- A, turn the direction round, fetch A.  Their four support scans are ONE pass: with
    //      these directions sdot's products are x * 1 + y * 0 and x * -1 + y * -0, so the supports are "first index of the
    //      largest x" / "of the largest -x" of each set.  An average call has 3.7 steps; as trips of the loop below these two
    //      cost what the others cost (simplex update, scan, convergence test, checkpoint: ~500 instructions each).  State after
    //      them: exactly the loop's (same keys, vertices, direction, counters, phase, checkpoint).
    if (live) {
        const int lane = threadIdx.x & 63, l = lane & 15, row0 = lane & 48;
        const bool have1 = l < g.P1.K, have2 = l < g.P2.K;
        const V2 nd = neg2(dir);                                  // (-1, -0)
        const double a_hi = have1 ? g.own1.x * dir.x + g.own1.y * dir.y : -__builtin_inf();      // step 1, set 1: direction
        const double b_lo = have2 ? g.own2.x * nd.x + g.own2.y * nd.y : -__builtin_inf();        // step 1, set 2: -direction
        const double a_lo = have1 ? g.own1.x * nd.x + g.own1.y * nd.y : -__builtin_inf();        // step 2, set 1: the direction turned round
        const double b_hi = have2 ? g.own2.x * dir.x + g.own2.y * dir.y : -__builtin_inf();      // step 2, set 2
        const unsigned long long n_ah = __ballot(a_hi != a_hi), n_bl = __ballot(b_lo != b_lo), n_al = __ballot(a_lo != a_lo),
                                 n_bh = __ballot(b_hi != b_hi);
        const int i_ah = row_argmax(a_hi, (n_ah >> row0) & 1), i_bl = row_argmax(b_lo, (n_bl >> row0) & 1);
        const int i_al = row_argmax(a_lo, (n_al >> row0) & 1), i_bh = row_argmax(b_hi, (n_bh >> row0) & 1);
        // step 1 (no point yet): A
        s.A.ii = pack_ii(i_ah, i_bl);
        s.A.v = sub2(point2(g.mem, g.P1, i_ah), point2(g.mem, g.P2, i_bl));
        s.keys = kA;
        g.n_support++;
        if (dotb2(s.A.v, dir) < 0) { phase = 1; chk.start(s, dir); }
        else if (++it >= max_iter) live = false;
        if (live) {
            // step 2 (one point): B <- A, the direction turned round, A
            if (phase) old = s;
            s.B = s.A; s.keys |= kB;
            dir = nd;
            s.A.ii = pack_ii(i_al, i_bh);
            s.A.v = sub2(point2(g.mem, g.P1, i_al), point2(g.mem, g.P2, i_bh));
            g.n_support++;
            if (!phase) {
                if (dotb2(s.A.v, dir) < 0) { phase = 1; chk.start(s, dir); }
                else if (++it >= max_iter) live = false;
            } else {
                if (matches_old2(g, old, s.A.v)) { conv = true; live = false; }
                else if (chk.step(s, dir)) { r.flag = 1; r.status = OBTG_ST_CYCLE; live = false; }
                else if (++rr >= md_cap) { r.flag = 1; r.status = OBTG_ST_MD_CAP; live = false; }
            }
        }
    }
    while (live) {
        if (phase) old = s;
        simplex_update2(s, dir);
        support_pts_quarter2<Mem>(g, dir, s.A);
        s.keys |= kA;
        if (!phase) {
            if (s.keys & kColl) { r.flag = 0; r.status = OBTG_ST_OK; live = false; }
            else if (dotb2(s.A.v, dir) < 0) { phase = 1; chk.start(s, dir); }
            else if (++it >= max_iter) live = false;
        } else {
            if (matches_old2(g, old, s.A.v)) { conv = true; live = false; }
            else if (chk.step(s, dir)) { r.flag = 1; r.status = OBTG_ST_CYCLE; live = false; }
            else if (++rr >= md_cap) { r.flag = 1; r.status = OBTG_ST_MD_CAP; live = false; }
        }
    }
    if (conv) {
        r.flag = 1; r.status = OBTG_ST_OK;
        closest_from_simplex2(g, old, r);
    }
    r.n_support = g.n_support;
}

}  // namespace gjk
}  // namespace obtg
