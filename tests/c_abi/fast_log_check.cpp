// csrc/fast_log.h compiled by the host compiler, next to logl: worst error in ulp over the quotients of 16-bit pixel pairs the
// row-pair statistic takes logarithms of, and over random normal doubles.  Prints "<worst over pairs> <worst over all> <share
// of pairs where it differs from the host libm's log>".
#include <stdint.h>
#include <stdio.h>
#include <string.h>
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
    double worst_pairs = 0, worst_all = 0;
    long differ = 0, n = 0;
    for (long it = 0; it < 4000000; ++it) {
        int a = 1 + (int)(next() % 65535), b = 1 + (int)(next() % 65535);
        if (it % 2) { b = a + (int)(next() % 2001) - 1000; b = b < 1 ? 1 : (b > 65535 ? 65535 : b); }   // neighbouring rows: ratios near 1
        const double r = (double)a / (double)b;
        const double f = shg::log_normal(r);
        const double e = ulp_err(f, logl((long double)r));
        worst_pairs = e > worst_pairs ? e : worst_pairs;
        differ += f != log(r);
        ++n;
    }
    for (long it = 0; it < 2000000; ++it) {
        uint64_t bits = (next() & 0x000fffffffffffffull) | ((uint64_t)(1 + next() % 2046) << 52);
        double x;
        memcpy(&x, &bits, 8);
        const double e = ulp_err(shg::log_normal(x), logl((long double)x));
        worst_all = e > worst_all ? e : worst_all;
    }
    printf("%.4f %.4f %.6f\n", worst_pairs, worst_all, (double)differ / (double)n);
    return 0;
}
