// Host build of csrc/gjk_true.h (the same source the HIP kernels compile) for CPU-side unit tests:
//   g++ -O2 -shared -fPIC tests/native/true_gjk_host.cpp -o <tmp>/libtrue_gjk_host.so
#include "../../optimalbeziertrajectorygeneration_amd/csrc/gjk_true.h"

using namespace obtg::tgjk;

// pts: [K][3]; out = dist, lower, c1[3], c2[3], flag, iters, status
extern "C" void true_gjk_host(const double* p1, int k1, const double* p2, int k2, double eps, double abs_tol, int max_iter,
                              double* out)
{
    auto pt1 = [&](int i) { return P3{ p1[3 * i], p1[3 * i + 1], p1[3 * i + 2] }; };
    auto pt2 = [&](int i) { return P3{ p2[3 * i], p2[3 * i + 1], p2[3 * i + 2] }; };
    auto sup = [&](const P3& d, int& i1, int& i2) {
        i1 = 0; i2 = 0;
        double m1 = dot(pt1(0), d), m2 = -dot(pt2(0), d);
        for (int i = 1; i < k1; ++i) { const double c = dot(pt1(i), d); if (c > m1) { m1 = c; i1 = i; } }
        for (int i = 1; i < k2; ++i) { const double c = -dot(pt2(i), d); if (c > m2) { m2 = c; i2 = i; } }
    };
    const Result r = true_distance(sup, pt1, pt2, eps, abs_tol, max_iter);
    out[0] = r.dist; out[1] = r.lower;
    out[2] = r.c1.x; out[3] = r.c1.y; out[4] = r.c1.z;
    out[5] = r.c2.x; out[6] = r.c2.y; out[7] = r.c2.z;
    out[8] = r.flag; out[9] = r.iters; out[10] = r.status;
}
