// csrc/fast_log.h compiled by the host compiler, next to logl: worst error in ulp over the quotients of 16-bit pixel pairs the
// row-pair statistic takes logarithms of, and over random normal doubles.  Prints "<worst over pairs> <worst over all> <share
// of pairs where it differs from the host libm's log> <worst of log_ratio_u16> <share of pairs where log_ratio_u16 != log_normal of the quotient>".
#include <stdint.h>
#include <stdio.h>
#include <string.h>
// log_ratio_u16's reciprocal here: the exact one made wrong by up to 2^-22 (the device's v_rcp_f64 is good to 2^-23), the error's
// sign and size varying with the operand -- the exactness of the quotient inside must not depend on the reciprocal's last 30 bits
static inline double coarse_rcp(double p) {
    uint64_t b;
    memcpy(&b, &p, 8);
    b = (b ^ (b >> 29)) * 0x9e3779b97f4a7c15ull;
    const double k = (double)(int64_t)(b >> 11) / 9007199254740992.0 * 2.0 - 1.0;      // in [-1, 1)
    return (1.0 / p) * (1.0 + k * 2.384185791015625e-07);
}
#define SHG_FASTLOG_TEST_RCP(p) coarse_rcp(p)
#include "fast_log.h"

static double ulp_err(double got, long double want) {
    if (want == 0) return got == 0 ? 0 : 1e9;
    int e;
    frexp((double)want, &e);
    return (double)fabsl(((long double)got - want) / ldexpl(1.0L, e - 53));
}
static uint64_t state = 88172645463325252ull;
static uint64_t next() { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return state; }

int main() {
    double worst_pairs = 0, worst_all = 0, worst_ratio = 0;
    long differ = 0, n = 0, ratio_differs = 0;
    for (long it = 0; it < 4000000; ++it) {
        int a = 1 + (int)(next() % 65535), b = 1 + (int)(next() % 65535);
        if (it % 2) { b = a + (int)(next() % 2001) - 1000; b = b < 1 ? 1 : (b > 65535 ? 65535 : b); }   // neighbouring rows: ratios near 1
        const double r = (double)a / (double)b;
        const double f = shg::log_normal(r);
        const double e = ulp_err(f, logl((long double)r));
        worst_pairs = e > worst_pairs ? e : worst_pairs;
        differ += f != log(r);
        ++n;
        // the one-reciprocal form the kernel uses for plain 16-bit rows: the same logarithm of the same (correctly rounded) quotient
        const double g = shg::log_ratio_u16((unsigned)a, (unsigned)b);
        const double eg = ulp_err(g, logl((long double)r));
        worst_ratio = eg > worst_ratio ? eg : worst_ratio;
        ratio_differs += g != f;
    }
    // every quotient with a small denominator and a sweep of near-equal pairs: the quotient inside log_ratio_u16 is a / b to the bit
    // (checked through the logarithm: a wrong last bit of q moves log q by ~50 ulp for q near 1)
    for (int b = 1; b <= 65535; b += 1 + b / 64)
        for (int a = (b > 300 ? b - 300 : 1); a <= b + 300 && a <= 65535; ++a) {
            const double r = (double)a / (double)b;
            const double eg = ulp_err(shg::log_ratio_u16((unsigned)a, (unsigned)b), logl((long double)r));
            worst_ratio = eg > worst_ratio ? eg : worst_ratio;
        }
    for (long it = 0; it < 2000000; ++it) {
        uint64_t bits = (next() & 0x000fffffffffffffull) | ((uint64_t)(1 + next() % 2046) << 52);
        double x;
        memcpy(&x, &bits, 8);
        const double e = ulp_err(shg::log_normal(x), logl((long double)x));
        worst_all = e > worst_all ? e : worst_all;
    }
    printf("%.4f %.4f %.6f %.4f %.6f\n", worst_pairs, worst_all, (double)differ / (double)n, worst_ratio, (double)ratio_differs / (double)n);
    return 0;
}
