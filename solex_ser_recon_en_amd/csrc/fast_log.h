// log(x) for x positive, finite and normal: the algorithm of fdlibm's __ieee754_log (error below 1 ulp) with its polynomials in
// fused multiply-adds, in ~40 float64 instructions where the device library's log takes ~85 (it carries the result in two
// doubles).  The row-pair statistic of the transversalium correction (solex_util.py:383-395) takes a logarithm per pixel pair
// and is bound by exactly these instructions.  np.log, the device library and this agree to the last bit or differ in it, as
// any two correctly working libm's do; DESIGN.md section 4 says what that means for parity.
// Plain C++ so that tests/ can compile the same text with the host compiler and put it next to logl.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define SHG_FASTLOG_FN __host__ __device__ __forceinline__
#else
#define SHG_FASTLOG_FN inline
#endif

namespace shg {

// a * b + K for a constant K.  On the device the constant sits in scalar registers and the multiply-add names three sources and a
// separate destination: the compiler's own choice for fma(a, b, K) is the two-operand v_fmac_f64, which first copies K into the
// destination (one v_mov_b64 per term of the polynomials below: 8 of the 36 instructions of a logarithm near 1).
SHG_FASTLOG_FN double fma_const(double a, double b, double K) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(K));
    return r;
#else
    return fma(a, b, K);
#endif
}

// n / d for normal operands whose quotient is normal: reciprocal, two Newton steps, one correction of the quotient
SHG_FASTLOG_FN double div_normal(double n, double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rcp(d);
#else
    double y = 1.0 / d;
#endif
    double e = fma(-d, y, 1.0);
    y = fma(y, e, y);
    e = fma(-d, y, 1.0);
    y = fma(y, e, y);
    const double q = n * y;
    const double r = fma(-d, q, n);
    return fma(r, y, q);
}

SHG_FASTLOG_FN double log_normal(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
#if defined(__HIP_DEVICE_COMPILE__)
    double m = __builtin_amdgcn_frexp_mant(x);           // [0.5, 1)
    int k = __builtin_amdgcn_frexp_exp(x);
#else
    int k;
    double m = frexp(x, &k);
#endif
    const bool low = m < 0.70710678118654752440;         // x = 2^k (1 + f), sqrt(2)/2 <= 1 + f < sqrt(2)
    m = low ? m + m : m;
    k = low ? k - 1 : k;
    const double f = m - 1.0;
    const double s = div_normal(f, 2.0 + f);
    const double z = s * s, w = z * z;
    const double t1 = w * fma_const(w, fma_const(w, Lg6, Lg4), Lg2);
    const double t2 = z * fma_const(w, fma_const(w, fma_const(w, Lg7, Lg5), Lg3), Lg1);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    return fma(dk, ln2_hi, -((hfsq - fma(s, hfsq + R, dk * ln2_lo)) - f));
}

// log(fl(a / b)) for two 16-bit pixel values a, b >= 1 (np.log(strip1 / strip0), solex_util.py:393): the quotient correctly
// rounded, then the algorithm above -- with ONE reciprocal for both divisions (a / b, and f / (2 + f) inside the logarithm) where
// the compiler's a / b and div_normal spend two reciprocals, four Newton steps and the scale / fix-up instructions of a general
// IEEE division.
//   * z ~ 1 / (b (a + b)) (both factors exact: below 2^33), ONE Newton step on the hardware's reciprocal (v_rcp_f64: relative
//     error <= 2^-23 by its description, 2^-24.4 measured on gfx950 -- tools/probes/rcp_f64_error.hip -- so 2^-45 after the step); then 1 / b ~ y = z (a + b) and 1 / (a + b) ~ u = z b, both to 2^-45;
//   * q = fl(a / b) EXACTLY: q0 = a y, r = a - b q0 (one fma; exact: the difference spans < 40 bits), q = fl(q0 + r y).
//     q0 + r y = a / b + (r / b) e with |e| <= 2^-45 and |r / b| <= 2^-44 a / b: within 2^-36 ulp of a / b, and a quotient of
//     integers below 2^16 that is not a double lies at least 2^-17 ulp from every rounding boundary
//     (|a / b - m| = |a 2^j - b (2 k + 1)| / (b 2^j) >= 1 / (b 2^j)): the rounding cannot differ.  (tests/c_abi/fast_log_check.cpp
//     runs this text with a reciprocal that is wrong by up to 2^-22 and compares q's logarithm with that of the compiler's a / b.)
//   * sqrt(2) / 2 <= q < sqrt(2) (neighbouring rows of a sunlit disk: always, but for a handful of pixels): k = 0, f = q - 1,
//     d = fl(2 + f), and 1 / d ~ b / (a + b) = b (z b) to a few ulp -- one Newton step and the same quotient correction give
//     s = f / d as div_normal does.  Any other q: other(q) -- log_normal(q) for the two-argument form.
//   * a zero pixel never takes the short way: a = 0 gives q = 0 exactly, b = 0 a NaN (the reciprocal of 0 is inf, and inf * 0),
//     neither passes the range test -- a caller that may see zeros tells them apart inside other() (k_rowpair_stats does: round 6,
//     the two compares with zero left the path every pair takes).
template <typename Other>
SHG_FASTLOG_FN double log_ratio_u16(unsigned a_px, unsigned b_px, Other other) {
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    const double a = (double)a_px, b = (double)b_px;
    const double t = a + b, p = b * t;
#if defined(__HIP_DEVICE_COMPILE__)
    double z = __builtin_amdgcn_rcp(p);
#elif defined(SHG_FASTLOG_TEST_RCP)
    double z = SHG_FASTLOG_TEST_RCP(p);                  // the host check's stand-in for the hardware's 2^-23 reciprocal
#else
    double z = 1.0 / p;
#endif
    z = fma(z, fma(-p, z, 1.0), z);
    const double u = z * b;                              // ~ 1 / (a + b)
    const double y = z * t;                              // ~ 1 / b
    const double q0 = a * y;
    const double q = fma(fma(-b, q0, a), y, q0);         // fl(a / b)
    if (!(q >= 0.70710678118654752440 && q < 1.41421356237309504880)) return other(q);
    const double f = q - 1.0, d = 2.0 + f;
    double v = b * u;                                    // ~ 1 / (q + 1)
    v = fma(v, fma(-d, v, 1.0), v);
    const double s0 = f * v;
    const double s = fma(fma(-d, s0, f), v, s0);
    const double zz = s * s, w = zz * zz;
    const double t1 = w * fma_const(w, fma_const(w, Lg6, Lg4), Lg2);
    const double t2 = zz * fma_const(w, fma_const(w, fma_const(w, Lg7, Lg5), Lg3), Lg1);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    return f - (hfsq - s * (hfsq + R));                  // (log_normal's last line with k = 0: -(x - f) = f - x to the bit)
}

SHG_FASTLOG_FN double log_ratio_u16(unsigned a_px, unsigned b_px) {                    // both pixels >= 1
    return log_ratio_u16(a_px, b_px, [](double q) { return log_normal(q); });
}

}  // namespace shg
