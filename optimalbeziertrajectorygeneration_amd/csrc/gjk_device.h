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

// gjk.py:87-114
template <class Mem>
__device__ __forceinline__ int support_idx(const Mem& mem, const Poly& P, const V3& d)
{
    int best = 0;
    double maxd = dot3(point(mem, P, 0), d);
    for (int i = 1; i < P.K; ++i) {
        const double cur = dot3(point(mem, P, i), d);
        if (cur > maxd) { maxd = cur; best = i; }
    }
    return best;
}

template <class Mem>
struct Ctx {
    Mem mem;
    Poly P1, P2;
    short* trace;      // nullable, [trace_cap][2]
    int trace_cap;
    int n_support;
};

// gjk.py:493-501
template <class Mem>
__device__ __forceinline__ void support_pts(Ctx<Mem>& g, const V3& dir, Vert& out)
{
    const int i1 = support_idx(g.mem, g.P1, dir);
    const int i2 = support_idx(g.mem, g.P2, neg(dir));
    out.i1 = i1; out.i2 = i2;
    out.v = sub(point(g.mem, g.P1, i1), point(g.mem, g.P2, i2));
    if (g.trace && g.n_support < g.trace_cap) {
        g.trace[2 * g.n_support] = (short)i1;
        g.trace[2 * g.n_support + 1] = (short)i2;
    }
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

// gjk.py:565-642
template <class Mem>
__device__ __forceinline__ void simplex3(Ctx<Mem>& g, Simplex& s, V3& dir)
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
    support_pts(g, dir, s.A);
    s.keys |= kA;
}

// gjk.py:505-561, 646-681
template <class Mem>
__device__ __forceinline__ void do_simplex(Ctx<Mem>& g, Simplex& s, V3& dir)
{
    if (!(s.keys & kA)) {
        support_pts(g, dir, s.A);
        s.keys |= kA;
    } else if (!(s.keys & kB)) {
        s.B = s.A; s.keys |= kB;
        dir = neg(dir);
        support_pts(g, dir, s.A);
    } else if (!(s.keys & kC)) {
        double dist;
        const double t = origin_to_line(s.A.v, s.B.v, dist);
        dir.x = -((1 - t) * s.A.v.x + t * s.B.v.x);
        dir.y = -((1 - t) * s.A.v.y + t * s.B.v.y);
        dir.z = -((1 - t) * s.A.v.z + t * s.B.v.z);
        s.C = s.A; s.keys |= kC;
        support_pts(g, dir, s.A);
    } else if (!(s.keys & kD)) {
        simplex3(g, s, dir);
    } else {
        const V3 A0 = neg(s.A.v), AB = sub(s.B.v, s.A.v), AC = sub(s.C.v, s.A.v), AD = sub(s.D.v, s.A.v);
        const V3 ABC = cross(AB, AC), ACD = cross(AC, AD), ADB = cross(AD, AB);
        if (dotb(ABC, A0) > 0) {
            s.keys &= ~kD;                       // pop('D') only; 'Dpts' stays (gjk.py:660)
            simplex3(g, s, dir);
        } else if (dotb(ACD, A0) > 0) {
            s.B = s.C; s.C = s.D; s.keys &= ~(kD | kDpts);
            simplex3(g, s, dir);
        } else if (dotb(ADB, A0) > 0) {
            s.C = s.B; s.B = s.D; s.keys &= ~(kD | kDpts);
            simplex3(g, s, dir);
        } else {
            s.keys |= kColl;
            dir = V3{ 0.0, 0.0, 0.0 };
        }
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

// gjk.py:230-270 gjkNew + 273-360 minimumDistance
template <class Mem>
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
        do_simplex(g, s, dir);
        if (s.keys & kColl) { r.flag = 0; r.status = OBTG_ST_OK; break; }
        if (dotb(s.A.v, dir) < 0) {
            Simplex old = s;
            bool conv = false;
            for (int rr = 0; rr < md_cap; ++rr) {
                old = s;
                do_simplex(g, s, dir);
                if (matches_old(g, old, s.A.v)) { conv = true; break; }
            }
            r.flag = 1;
            if (!conv) { r.status = OBTG_ST_MD_CAP; break; }
            r.status = OBTG_ST_OK;
            s = old;
            if (s.keys & kC) {
                const V3 A0 = neg(s.A.v), AB = sub(s.B.v, s.A.v), AC = sub(s.C.v, s.A.v);
                const V3 ABC = cross(AB, AC);
                if (dotb(cross(ABC, AC), A0) >= 0) {
                    seg_result(g, s.A, s.C, r);
                } else if (dotb(cross(AB, ABC), A0) >= 0) {
                    seg_result(g, s.A, s.B, r);
                } else {
                    // gjk.py:440-477 weightedOriginToPlane (a**2 taken as a*a: the reference's
                    // libm pow(a, 2.0) can differ from it by one ulp of the denominator)
                    const V3 N = cross(sub(s.B.v, s.A.v), sub(s.C.v, s.A.v));
                    const double nn = normb(N);
                    const V3 n{ N.x / nn, N.y / nn, N.z / nn };
                    const double tq = (n.x * s.A.v.x + n.y * s.A.v.y + n.z * s.A.v.z) /
                                      (n.x * n.x + n.y * n.y + n.z * n.z);
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
            break;
        }
    }
    r.n_support = g.n_support;
}

}  // namespace gjk
}  // namespace obtg
