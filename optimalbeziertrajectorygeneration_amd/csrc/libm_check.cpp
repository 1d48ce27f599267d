// Load-time check of the one environment fact the device code bakes in: `a**2` of gjk/gjk.py:460 is libm's pow(a, 2.0), and
// csrc/libm_pow2.h restates the pow of glibc 2.35's FMA build for the device.  On a host whose libm rounds pow differently the
// host-side squares (square_as_python: maxSep**2 ..., through the live libm as Python forms them) and the reference's own
// fixtures would no longer be what the device computes -- by one ulp, on about one input in a thousand.  obtg_libm_pow_matches()
// says so: the same header compiled for the host (this unit: -ffp-contract=off, as for the device) against the live pow on a
// few thousand inputs of the kind the device sees (unit-vector components), once per process.
#include <cmath>
#include <cstdint>

#define OBTG_P2_TABLE static const
#define OBTG_P2_FUNC static inline
#include "libm_pow2.h"

extern "C" __attribute__((visibility("default"))) int obtg_libm_pow_matches(void)
{
    static int cached = -1;
    if (cached >= 0) return cached;
    double (*volatile live_pow)(double, double) = std::pow;        // (volatile: clang folds a literal pow(x, 2.0) into x * x)
    uint64_t s = 0x9e3779b97f4a7c15ull;
    int ok = 1;
    for (int i = 0; i < 4096 && ok; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;                    // xorshift64
        const double u = (double)(s >> 11) * (1.0 / 9007199254740992.0);        // [0, 1)
        double x = (i & 1) ? u : -u;
        if ((i & 7) == 7) x = std::ldexp(x, -(int)((s >> 3) & 63));               // small components
        if ((i & 15) == 8) x = 1.0 + std::ldexp(u - 0.5, -20);                    // the neighbourhood of 1
        const double a = obtg_square_as_libm_pow(x), b = live_pow(x, 2.0);
        uint64_t ua, ub;
        __builtin_memcpy(&ua, &a, 8); __builtin_memcpy(&ub, &b, 8);
        if (ua != ub) ok = 0;
    }
    cached = ok;
    return cached;
}
