// A textbook Gilbert-Johnson-Keerthi distance between the convex hulls of two point sets -- NOT the reference's
// gjkNew.  gjkNew (gjk/gjk.py:230-360) returns the distance between two particular hull points as soon as a support
// point fails `A . d < 0` and declares convergence when a support repeats: a non-minimal answer on about 30 % of
// separated pairs, and loops that never exit on some 3-D inputs.  The robust entry points
// (obtg_gjk_true_pairs, obtg_min_dist2poly_robust) need a distance that IS the hull distance:
//
//   v   = point of the current simplex (of the Minkowski difference P1 - P2) closest to the origin
//   w   = support point of P1 - P2 in direction -v
//   |v| >= dist(hull1, hull2) >= (v . w) / |v|          (w's supporting plane separates the origin from the hull)
//
// and the loop ends when the two bounds agree within a relative eps, so the returned |v| carries that guarantee.
// The closest point of a simplex (up to a tetrahedron) is found region by region (vertex / edge / face Voronoi
// tests, C. Ericson, Real-Time Collision Detection, section 5.1); barycentric weights give the closest points on the
// two hulls.  Host- and device-callable (tests/test_true_gjk_host.py runs it on the CPU against a QP solver).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define OBTG_HD __host__ __device__ __forceinline__
#else
#define OBTG_HD inline
#endif

namespace obtg {
namespace tgjk {

struct P3 { double x, y, z; };
OBTG_HD P3 sub(const P3& a, const P3& b) { return P3{ a.x - b.x, a.y - b.y, a.z - b.z }; }
OBTG_HD double dot(const P3& a, const P3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
OBTG_HD P3 cross(const P3& a, const P3& b) { return P3{ a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
OBTG_HD P3 comb2(double a, const P3& p, double b, const P3& q) { return P3{ a * p.x + b * q.x, a * p.y + b * q.y, a * p.z + b * q.z }; }

struct Vtx { P3 v; int i1, i2; };          // Minkowski vertex p1[i1] - p2[i2]

struct Simplex {
    Vtx s[4];
    double lam[4];                          // barycentric weights of the closest point
    int n;
};

// closest point to the origin on segment s0-s1; keeps only the vertices that carry weight
OBTG_HD P3 closest_segment(Simplex& S)
{
    const P3 a = S.s[0].v, b = S.s[1].v, ab = sub(b, a);
    const double den = dot(ab, ab);
    double t = den > 0.0 ? -dot(a, ab) / den : 0.0;
    if (t <= 0.0) { S.n = 1; S.lam[0] = 1.0; return a; }
    if (t >= 1.0) { S.s[0] = S.s[1]; S.n = 1; S.lam[0] = 1.0; return b; }
    S.lam[0] = 1.0 - t; S.lam[1] = t;
    return comb2(1.0 - t, a, t, b);
}

// closest point to the origin on triangle (a, b, c) given by vertex slots ia, ib, ic of S; writes the reduced
// simplex to R.  Voronoi regions in the order vertex A, B, edge AB, vertex C, edge AC, edge BC, interior.
OBTG_HD P3 closest_triangle(const Simplex& S, int ia, int ib, int ic, Simplex& R)
{
    const P3 a = S.s[ia].v, b = S.s[ib].v, c = S.s[ic].v;
    const P3 ab = sub(b, a), ac = sub(c, a);
    const double d1 = -dot(ab, a), d2 = -dot(ac, a);                 // ap = -a
    if (d1 <= 0.0 && d2 <= 0.0) { R.n = 1; R.s[0] = S.s[ia]; R.lam[0] = 1.0; return a; }
    const double d3 = -dot(ab, b), d4 = -dot(ac, b);
    if (d3 >= 0.0 && d4 <= d3) { R.n = 1; R.s[0] = S.s[ib]; R.lam[0] = 1.0; return b; }
    const double vc = d1 * d4 - d3 * d2;
    if (vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0) {
        const double t = d1 / (d1 - d3);
        R.n = 2; R.s[0] = S.s[ia]; R.s[1] = S.s[ib]; R.lam[0] = 1.0 - t; R.lam[1] = t;
        return comb2(1.0 - t, a, t, b);
    }
    const double d5 = -dot(ab, c), d6 = -dot(ac, c);
    if (d6 >= 0.0 && d5 <= d6) { R.n = 1; R.s[0] = S.s[ic]; R.lam[0] = 1.0; return c; }
    const double vb = d5 * d2 - d1 * d6;
    if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) {
        const double t = d2 / (d2 - d6);
        R.n = 2; R.s[0] = S.s[ia]; R.s[1] = S.s[ic]; R.lam[0] = 1.0 - t; R.lam[1] = t;
        return comb2(1.0 - t, a, t, c);
    }
    const double va = d3 * d6 - d5 * d4;
    if (va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0) {
        const double t = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        R.n = 2; R.s[0] = S.s[ib]; R.s[1] = S.s[ic]; R.lam[0] = 1.0 - t; R.lam[1] = t;
        return comb2(1.0 - t, b, t, c);
    }
    const double sum = va + vb + vc;
    if (!(sum > 0.0)) {                      // degenerate (collinear) triangle: fall back to its longest edge pair
        R.n = 2; R.s[0] = S.s[ia]; R.s[1] = S.s[ib];
        P3 best = closest_segment(R);
        Simplex T; T.n = 2; T.s[0] = S.s[ia]; T.s[1] = S.s[ic];
        const P3 q = closest_segment(T);
        if (dot(q, q) < dot(best, best)) { R = T; best = q; }
        T.n = 2; T.s[0] = S.s[ib]; T.s[1] = S.s[ic];
        const P3 q2 = closest_segment(T);
        if (dot(q2, q2) < dot(best, best)) { R = T; best = q2; }
        return best;
    }
    const double v = vb / sum, w = vc / sum, u = 1.0 - v - w;
    R.n = 3; R.s[0] = S.s[ia]; R.s[1] = S.s[ib]; R.s[2] = S.s[ic];
    R.lam[0] = u; R.lam[1] = v; R.lam[2] = w;
    return P3{ u * a.x + v * b.x + w * c.x, u * a.y + v * b.y + w * c.y, u * a.z + v * b.z + w * c.z };
}

// closest point to the origin on the current simplex; reduces S to the supporting sub-simplex.
// Returns false when the origin is inside a (non-degenerate) tetrahedron: the hulls intersect.
OBTG_HD bool closest_on_simplex(Simplex& S, P3& v)
{
    if (S.n == 1) { S.lam[0] = 1.0; v = S.s[0].v; return true; }
    if (S.n == 2) { v = closest_segment(S); return true; }
    if (S.n == 3) { Simplex R; R.n = 0; v = closest_triangle(S, 0, 1, 2, R); S = R; return true; }
    // tetrahedron: the origin is outside face (i, j, k) when it lies on the other side of that face than the
    // fourth vertex; the answer is the nearest of the closest points of such faces
    const int F[4][4] = { { 0, 1, 2, 3 }, { 0, 2, 3, 1 }, { 0, 3, 1, 2 }, { 1, 3, 2, 0 } };
    bool any = false;
    double best = 0.0;
    Simplex Rb; Rb.n = 0;
    P3 vb{ 0, 0, 0 };
    for (int f = 0; f < 4; ++f) {
        const P3 a = S.s[F[f][0]].v, b = S.s[F[f][1]].v, c = S.s[F[f][2]].v, d = S.s[F[f][3]].v;
        const P3 nrm = cross(sub(b, a), sub(c, a));
        const double so = -dot(nrm, a);                    // origin side
        const double sd = dot(nrm, sub(d, a));             // fourth vertex side
        const bool outside = (sd == 0.0) ? true : (so * sd < 0.0);      // flat tetrahedron: treat every face
        if (!outside) continue;
        Simplex R; R.n = 0;
        const P3 q = closest_triangle(S, F[f][0], F[f][1], F[f][2], R);
        const double qq = dot(q, q);
        if (!any || qq < best) { any = true; best = qq; Rb = R; vb = q; }
    }
    if (!any) { v = P3{ 0, 0, 0 }; return false; }
    S = Rb; v = vb;
    return true;
}

struct Result {
    double dist;        // |v|: distance between the hulls within a relative eps (0 when they intersect)
    double lower;       // proven lower bound (v . w) / |v| at exit
    P3 c1, c2;          // closest points on hull 1 / hull 2
    int flag;           // 1 separated, 0 intersecting / touching
    int iters;
    int status;         // 0 converged WITH the certificate dist - lower <= eps * dist; 1 iteration cap; 2 stalled at rounding
                        // level (a support point repeated, or the simplex step did not get closer) before the certificate
                        // closed.  In every case dist is a distance between hull points (an upper bound) and `lower` a
                        // proven lower bound: callers that need the certificate check status == 0 or compare the two.
};

// Sup(dir, i1, i2): indices of the support points of set 1 in direction dir and of set 2 in direction -dir;
// Pt1(i), Pt2(i): the points.  eps: relative gap between upper and lower bound at exit.
template <class Sup, class Pt1, class Pt2>
OBTG_HD Result true_distance(Sup sup, Pt1 pt1, Pt2 pt2, double eps, double abs_tol, int max_iter)
{
    Simplex S;
    S.n = 1;
    S.s[0].i1 = 0; S.s[0].i2 = 0;
    S.s[0].v = sub(pt1(0), pt2(0));
    S.lam[0] = 1.0;
    P3 v = S.s[0].v;
    Result r;
    r.flag = 1; r.status = 1; r.lower = 0.0; r.iters = 0;
    for (int it = 0; it < max_iter; ++it) {
        r.iters = it + 1;
        const double vv = dot(v, v);
        if (vv <= abs_tol * abs_tol) { r.flag = 0; r.status = 0; break; }
        int i1, i2;
        sup(P3{ -v.x, -v.y, -v.z }, i1, i2);
        const P3 w = sub(pt1(i1), pt2(i2));
        const double vw = dot(v, w);
        if (vw > 0.0) { const double lb = vw / sqrt(vv); if (lb > r.lower) r.lower = lb; }
        if (vv - vw <= eps * vv) { r.status = 0; break; }                 // upper and lower bound agree
        bool dup = false;
        for (int q = 0; q < S.n; ++q) dup = dup || (S.s[q].i1 == i1 && S.s[q].i2 == i2);
        if (dup) { r.status = 2; break; }                                  // no new vertex, certificate not closed: stalled
        const Simplex S_before = S;
        S.s[S.n].v = w; S.s[S.n].i1 = i1; S.s[S.n].i2 = i2;
        S.n++;
        P3 nv;
        if (!closest_on_simplex(S, nv)) { v = nv; r.flag = 0; r.status = 0; break; }
        if (!(dot(nv, nv) < vv)) { S = S_before; r.status = 2; break; }    // no closer: keep the better v and its simplex
        v = nv;                                                            // (S and its weights describe nv)
    }
    if (r.flag == 0) {
        r.dist = 0.0; r.lower = 0.0;
        r.c1 = pt1(S.s[0].i1); r.c2 = r.c1;
        return r;
    }
    r.dist = sqrt(dot(v, v));
    if (r.lower > r.dist) r.lower = r.dist;
    P3 c1{ 0, 0, 0 }, c2{ 0, 0, 0 };
    for (int q = 0; q < S.n; ++q) {
        const P3 a = pt1(S.s[q].i1), b = pt2(S.s[q].i2);
        c1.x += S.lam[q] * a.x; c1.y += S.lam[q] * a.y; c1.z += S.lam[q] * a.z;
        c2.x += S.lam[q] * b.x; c2.y += S.lam[q] * b.y; c2.z += S.lam[q] * b.z;
    }
    r.c1 = c1; r.c2 = c2;
    return r;
}

}  // namespace tgjk
}  // namespace obtg
