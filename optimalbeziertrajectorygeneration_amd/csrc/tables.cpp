// Host-side coefficient tables (the reference's elevMatrix / prodMatrix / bezProductCoefficients,
// bezier.py:1127-1208, in the sparse forms the kernels consume).
#include <cmath>

#include "obtg_internal.h"

namespace obtg {

// scipy.special.binom restricted to integer arguments: 0 outside 0 <= k <= n
// (the reference relies on that for the band structure, bezier.py:1145).
double binom(int n, int k)
{
    if (k < 0 || k > n) return 0.0;
    if (k > n - k) k = n - k;
    long double r = 1.0L;
    for (int i = 1; i <= k; ++i) r = r * (long double)(n - k + i) / (long double)i;
    return (double)r;
}

std::vector<double> binom_row(int n)
{
    std::vector<double> r(n + 1);
    for (int k = 0; k <= n; ++k) r[k] = binom(n, k);
    return r;
}

// Equal-degree product weights w(k,j) = C(n,j) C(n,k-j) / C(2n,k) (bezier.py:1183-1208),
// folded over the symmetry (j, k-j) <-> (k-j, j) and pre-multiplied by the reference's
// normSquare factor dim/2 (bezier.py:884, 1744-1756).  Layout [2n+1][n+1]; only entries with
// max(0,k-n) <= j <= k/2 are non-zero:  W2[k][j] = (dim/2) * w(k,j) * (2 if j != k-j else 1).  Behind them the separable
// form of the same weights: C(n, .)[n+1], then S[2n+1].
std::vector<double> folded_product_weights(int n, int dim)
{
    int L = 2 * n + 1, nc = n + 1;
    std::vector<double> W((size_t)L * nc, 0.0);
    for (int k = 0; k < L; ++k) {
        double den = binom(2 * n, k);
        for (int j = (k - n > 0 ? k - n : 0); 2 * j <= k; ++j) {
            double w = binom(n, j) * binom(n, k - j) / den;
            if (j != k - j) w *= 2.0;
            W[(size_t)k * nc + j] = w * (0.5 * dim);
        }
    }
    // the separable form of the same weights (bern_device.h normsq_coeffs): C(n, j), then S_k = (dim/2) / C(2n, k), doubled for odd k
    for (int j = 0; j < nc; ++j) W.push_back(binom(n, j));
    for (int k = 0; k < L; ++k) W.push_back(((k & 1) ? 1.0 : 0.5) * dim / binom(2 * n, k));
    return W;
}

// Degree elevation of an L_in-coefficient curve by R (bezier.py:1127-1147): T[j][k] = C(N,j) C(R,k-j) / C(N+R,k),
// N = L_in-1, k = 0..N+R, j = 0..N, in two layouts.
// (a) in the operand order of v_mfma_f64_16x16x4_f64 (bern_device.h elev_rows_mfma): B fragment
// (n-tile t, k-step s) = 64 doubles, lane l holds T[j = 4s + (l >> 4)][k = 16t + (l & 15)], zero outside the matrix;
// layout [NT = ceil((L_in+R)/16)][KS = ceil(L_in/4)][64].  Entries through long double binomials: C(4n, j) C(4R, k-j)
// of the angular rate's elevation (4R up to 1000) overflows binary64 before the division brings it back below 1.
static long double binom_ld(int n, int k)
{
    if (k < 0 || k > n) return 0.0L;
    if (k > n - k) k = n - k;
    long double r = 1.0L;
    for (int i = 1; i <= k; ++i) r = r * (long double)(n - k + i) / (long double)i;
    return r;
}

std::vector<double> elev_table_frag(int L_in, int R)
{
    const int N = L_in - 1, Lr = L_in + R, NT = (Lr + 15) / 16, KS = (L_in + 3) / 4;
    std::vector<double> F((size_t)NT * KS * 64, 0.0);
    for (int t = 0; t < NT; ++t)
        for (int s = 0; s < KS; ++s)
            for (int l = 0; l < 64; ++l) {
                const int j = 4 * s + (l >> 4), k = 16 * t + (l & 15);
                if (j <= N && k < Lr)
                    F[((size_t)t * KS + s) * 64 + l] = (double)(binom_ld(N, j) * binom_ld(R, k - j) / binom_ld(N + R, k));
            }
    return F;
}

// (b) dense and transposed, Tt[k][j], through the same long double route (the lane-per-item forms of the same chain
// read rows of it)
std::vector<double> elev_table_T_ld(int L_in, int R)
{
    const int N = L_in - 1, Lr = L_in + R;
    std::vector<double> T((size_t)Lr * L_in, 0.0);
    for (int k = 0; k < Lr; ++k)
        for (int j = 0; j <= N; ++j) T[(size_t)k * L_in + j] = (double)(binom_ld(N, j) * binom_ld(R, k - j) / binom_ld(N + R, k));
    return T;
}

// Tables of the double-double recompute of ill-conditioned angular-rate rows (bern_kernels.hip k_angrate_dd), every
// entry as (hi, lo) with hi + lo = the long double value (64-bit mantissa: 5e-20 relative, against conditions of up to
// 1e6 on those rows).  Layout, all for degree n, dim 2:
//   wn[2n+1][n+1][2]     plain product weights C(n,j) C(n,k-j) / C(2n,k)
//   w2n[2n+1][n+1][2]    folded weights of the square at degree n (x2 off the diagonal)
//   w22n[4n+1][2n+1][2]  folded weights of the square at degree 2n
//   ratio[n+1][2]        c / n                    (diff_elev1's elevation by one)
//   row4[4R+1][2]        C(4R, m) 2^-e            (e as elev_conv_padded's normalisation)
//   sc4[4n+1][2]         C(4n, k)                 (beyond 2^53 from degree 14 on)
// returned back to back; offsets in DdTables.
static void push_dd(std::vector<double>& t, long double v)
{
    const double hi = (double)v;
    t.push_back(hi);
    t.push_back((double)(v - (long double)hi));
}

DdTables angrate_dd_tables(int n, int R, std::vector<double>& t)
{
    DdTables o{};
    t.clear();
    const int L2 = 2 * n + 1, L4 = 4 * n + 1, nc = n + 1;
    o.wn = 0;
    for (int k = 0; k < L2; ++k)
        for (int j = 0; j < nc; ++j) push_dd(t, (k - j >= 0 && k - j <= n) ? binom_ld(n, j) * binom_ld(n, k - j) / binom_ld(2 * n, k) : 0.0L);
    o.w2n = (int)t.size();
    for (int k = 0; k < L2; ++k)
        for (int j = 0; j < nc; ++j) {
            long double w = 0.0L;
            if (j >= (k - n > 0 ? k - n : 0) && 2 * j <= k) { w = binom_ld(n, j) * binom_ld(n, k - j) / binom_ld(2 * n, k); if (j != k - j) w *= 2.0L; }
            push_dd(t, w);
        }
    o.w22n = (int)t.size();
    for (int k = 0; k < L4; ++k)
        for (int j = 0; j < L2; ++j) {
            long double w = 0.0L;
            if (j >= (k - 2 * n > 0 ? k - 2 * n : 0) && 2 * j <= k) { w = binom_ld(2 * n, j) * binom_ld(2 * n, k - j) / binom_ld(4 * n, k); if (j != k - j) w *= 2.0L; }
            push_dd(t, w);
        }
    o.ratio = (int)t.size();
    for (int c = 0; c < nc; ++c) push_dd(t, (long double)c / (long double)n);
    o.row4 = (int)t.size();
    int e = 0;
    (void)std::frexp(binom(4 * R, 2 * R), &e);
    for (int m = 0; m <= 4 * R; ++m) push_dd(t, std::ldexp(binom_ld(4 * R, m), -e));
    o.sc4 = (int)t.size();
    for (int k = 0; k < L4; ++k) push_dd(t, binom_ld(4 * n, k));
    return o;
}

// Degree elevation by R of an L_in-coefficient curve as a scaled convolution: three rows back to back
//   scale[L_in] = C(N, j);  binp[R + 2 L_in - 1 + 8] = C(R, m), m = -(L_in-1) .. R+L_in-1+8;  inv[L_in+R+8] = 1/C(N+R, k)
// (8 = kConvPad of bern_device.h: kernels that produce blocks of 8 output columns read that far past the end)
std::vector<double> elev_conv_tables(int L_in, int R)
{
    return elev_conv_padded(L_in, R, 8, false, true);
}

// The same elevation for kernels that walk a window of the binomial row (k_dynamics_elev):
//   scale[L_in] = C(N, j) | row[(L_in-1) + (R+1) + (L_in-1) + extra] = C(R, m) for m = -(L_in-1) .. , zero outside 0..R
//   | (with_inv) inv[L_in + R + extra] = 1 / C(N+R, k), zero past the end.
// normalise: the row is divided by the power of two below its largest entry (a quotient of two elevations does not
// see the factor; keeps C(4R, .) x C(4n, .) x values far from overflow).
std::vector<double> elev_conv_padded(int L_in, int R, int extra, bool normalise, bool with_inv)
{
    const int N = L_in - 1;
    std::vector<double> t;
    for (int j = 0; j <= N; ++j) t.push_back(binom(N, j));
    int e = 0;
    if (normalise) (void)std::frexp(binom(R, R / 2), &e);
    for (int m = -(L_in - 1); m <= R + (L_in - 1) + extra; ++m) t.push_back(std::ldexp(binom(R, m), -e));
    if (with_inv)
        for (int k = 0; k < L_in + R + extra; ++k) t.push_back(k <= N + R ? 1.0 / binom(N + R, k) : 0.0);
    return t;
}

}  // namespace obtg

// pow through a volatile pointer: clang folds a literal pow(x, 2.0) into x * x, and the two are one ulp apart on 0.09 % of inputs
// with glibc 2.35 (libm_pow2.h)
namespace obtg {
double square_as_python(double x)
{
    static double (*volatile libm_pow)(double, double) = std::pow;
    return libm_pow(x, 2.0);
}
double cube_as_python(double x)
{
    static double (*volatile libm_pow)(double, double) = std::pow;
    return libm_pow(x, 3.0);
}
}  // namespace obtg
