// Host control plane of the SHG path: the 1-D and scalar arithmetic between the kernels.
//
// The reference does these steps with NumPy / SciPy calls on a few thousand values at most
// (polynomial fits of the line trace, the mode of the residuals, the limb-point selection, the
// ellipse fit, the Savitzky-Golay trend of the row ratios).  They are restated here in C++ so that a
// whole pipeline stage is ONE call that holds no interpreter lock: several scans can then be in
// flight in one process (Solex_recon.solex_do_work), each on its own HIP stream.
//
// Parity.  The raw disks are bit-exact only if the line fit is, so everything on the way to `fit`
// follows NumPy operation by operation (np.vander's running products, the sequential column norms,
// pairwise summation in np.mean / np.std, np.around as multiply-rint-divide, Horner's rule in
// polyval) and the least-squares solve is LAPACK's dgelsd itself: _lib.py hands over the address of
// the routine inside the OpenBLAS that NumPy loaded (shg_host_bind_lapack), called with NumPy's own
// workspace query.  Without a bound LAPACK a Householder QR solves the same system (same mathematics,
// last bits may differ).  The ellipse fit and the 2x2 correction-matrix algebra stay NumPy calls in the
// Python layer (limb_fit.two_step, ellipse_to_circle.get_correction_matrix): phi and ratio steer every
// sample position of the warp, a 1e-12 difference already flips pixels by one grey level, and NumPy's
// BLAS-backed products cannot be reproduced bit for bit from here.
//
// Every function cites the reference lines it replaces.  Compiled with -ffp-contract=off.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <vector>
#include "shg_common.h"

namespace shg {
namespace host {

// Python's and NumPy's scalar `**` are libm's pow(), whose result is not always the correctly rounded one (x ** 2 can
// differ from x * x in the last bit); the compiler would fold pow(x, 2.0) into x*x and pow(x, 0.5) into sqrt(x), so
// the calls that must match the reference go through a volatile pointer.
static double (*volatile libm_pow)(double, double) = pow;

// ---- LAPACK bridge ---------------------------------------------------------------------------
// ILP64 Fortran interface of OpenBLAS as NumPy 2.x bundles it (symbol scipy_dgelsd_64_).
typedef void (*dgelsd_fn)(const int64_t* m, const int64_t* n, const int64_t* nrhs, double* a, const int64_t* lda,
                          double* b, const int64_t* ldb, double* s, const double* rcond, int64_t* rank, double* work,
                          const int64_t* lwork, int64_t* iwork, int64_t* info);
static std::atomic<dgelsd_fn> g_dgelsd{nullptr};
typedef int64_t (*mode_pick_fn)(const int64_t* neg_counts, int64_t n);
static std::atomic<mode_pick_fn> g_mode_pick{nullptr};

// Householder QR least squares (full column rank assumed; used only when no LAPACK is bound).
static int lstsq_qr(const double* a, int64_t m, int64_t n, const double* b, double* x) {
    std::vector<double> q((size_t)m * n), r(b, b + m);
    for (int64_t i = 0; i < m; ++i)
        for (int64_t j = 0; j < n; ++j) q[(size_t)j * m + i] = a[i * n + j];        // column major
    const int64_t k = std::min(m, n);
    for (int64_t j = 0; j < k; ++j) {
        double* col = &q[(size_t)j * m];
        long double nrm = 0;
        for (int64_t i = j; i < m; ++i) nrm += (long double)col[i] * col[i];
        nrm = sqrtl(nrm);
        if (nrm == 0) continue;
        const double alpha = col[j] > 0 ? -(double)nrm : (double)nrm;
        std::vector<double> v(col + j, col + m);
        v[0] -= alpha;
        long double vn = 0;
        for (double t : v) vn += (long double)t * t;
        if (vn == 0) continue;
        auto reflect = [&](double* y) {
            long double d = 0;
            for (int64_t i = j; i < m; ++i) d += (long double)v[i - j] * y[i];
            const double f = (double)(2 * d / vn);
            for (int64_t i = j; i < m; ++i) y[i] -= f * v[i - j];
        };
        for (int64_t c = j; c < n; ++c) reflect(&q[(size_t)c * m]);
        reflect(r.data());
    }
    for (int64_t j = n - 1; j >= 0; --j) {
        if (j >= m) { x[j] = 0; continue; }
        long double s = r[j];
        for (int64_t c = j + 1; c < n; ++c) s -= (long double)q[(size_t)c * m + j] * x[c];
        const double d = q[(size_t)j * m + j];
        x[j] = d != 0 ? (double)(s / d) : 0.0;
    }
    return 0;
}

// np.linalg.lstsq(a, b, rcond)[0] for a row-major [m][n] matrix and one right-hand side:
// umath_linalg's call sequence (Fortran-order copies, ldb = max(m, n), workspace sizes from a query).
static int lstsq(const double* a, int64_t m, int64_t n, const double* b, double rcond, double* x) {
    dgelsd_fn f = g_dgelsd.load();
    if (!f) return lstsq_qr(a, m, n, b, x);
    const int64_t nrhs = 1, lda = std::max<int64_t>(1, m), ldb = std::max<int64_t>(1, std::max(m, n));
    std::vector<double> af((size_t)m * n), bf((size_t)ldb, 0.0), s((size_t)std::min(m, n) + 1);
    for (int64_t i = 0; i < m; ++i)
        for (int64_t j = 0; j < n; ++j) af[(size_t)j * lda + i] = a[i * n + j];
    for (int64_t i = 0; i < m; ++i) bf[i] = b[i];
    int64_t rank = 0, info = 0, lwork = -1, iwork_q = 0;
    double work_q = 0;
    f(&m, &n, &nrhs, af.data(), &lda, bf.data(), &ldb, s.data(), &rcond, &rank, &work_q, &lwork, &iwork_q, &info);
    if (info != 0) { set_error("lstsq: dgelsd workspace query failed (info %lld)", (long long)info); return SHG_E_LINALG; }
    lwork = (int64_t)work_q;
    std::vector<double> work((size_t)std::max<int64_t>(1, lwork));
    std::vector<int64_t> iwork((size_t)std::max<int64_t>(1, iwork_q));
    f(&m, &n, &nrhs, af.data(), &lda, bf.data(), &ldb, s.data(), &rcond, &rank, work.data(), &lwork, iwork.data(), &info);
    if (info > 0) { set_error("SVD did not converge in Linear Least Squares"); return SHG_E_LINALG; }
    if (info < 0) { set_error("lstsq: dgelsd argument %lld", (long long)-info); return SHG_E_LINALG; }
    for (int64_t j = 0; j < n; ++j) x[j] = bf[j];
    return 0;
}

// ---- NumPy building blocks ----------------------------------------------------------------------
// np.add.reduce over a contiguous float64 vector: NumPy's pairwise summation (blocks of 128, 8 accumulators).
static double pairwise_sum(const double* a, int64_t n) {
    if (n < 8) {
        double res = 0.;
        for (int64_t i = 0; i < n; ++i) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int k = 0; k < 8; ++k) r[k] = a[k];
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; ++k) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
}

static double np_mean(const double* a, int64_t n) { return pairwise_sum(a, n) / (double)n; }

// np.std(a): sqrt(mean(|a - mean(a)|^2)), NumPy's _var
static double np_std(const double* a, int64_t n) {
    const double mu = np_mean(a, n);
    std::vector<double> d((size_t)n);
    for (int64_t i = 0; i < n; ++i) { const double t = a[i] - mu; d[i] = t * t; }
    return sqrt(pairwise_sum(d.data(), n) / (double)n);
}

// np.polyfit(x, y, 3) -> coefficients, highest power first (numpy/lib/_polynomial_impl.py)
static int polyfit3_desc(const double* x, const double* y, int64_t n, double* c4) {
    if (n <= 0) { set_error("expected non-empty vector for x"); return SHG_E_TYPE; }
    std::vector<double> lhs((size_t)n * 4);
    for (int64_t i = 0; i < n; ++i) {                      // np.vander: running products x, x*x, (x*x)*x
        const double x1 = x[i], x2 = x1 * x1, x3 = x2 * x1;
        double* r = &lhs[(size_t)i * 4];
        r[0] = x3; r[1] = x2; r[2] = x1; r[3] = 1.0;
    }
    double scale[4] = {0, 0, 0, 0};
    for (int64_t i = 0; i < n; ++i)                         // (lhs * lhs).sum(axis=0): row after row
        for (int j = 0; j < 4; ++j) { const double v = lhs[(size_t)i * 4 + j]; scale[j] += v * v; }
    for (int j = 0; j < 4; ++j) scale[j] = sqrt(scale[j]);
    for (int64_t i = 0; i < n; ++i)
        for (int j = 0; j < 4; ++j) lhs[(size_t)i * 4 + j] /= scale[j];
    const double rcond = (double)n * 2.220446049250313e-16;
    if (int e = lstsq(lhs.data(), n, 4, y, rcond, c4)) return e;
    for (int j = 0; j < 4; ++j) c4[j] /= scale[j];
    return 0;
}

// numpy.polynomial.polynomial.polyfit(x, y, 3) (polyutils._fit) -> coefficients, lowest power first
static int polyfit3_asc(const double* x, const double* y, int64_t n, double* c4) {
    if (n <= 0) { set_error("expected non-empty vector for x"); return SHG_E_TYPE; }
    std::vector<double> lhs((size_t)n * 4);
    double scl[4] = {0, 0, 0, 0};
    for (int64_t i = 0; i < n; ++i) {                      // polyvander: v0 = x*0 + 1, v1 = x, vk = v(k-1) * x
        double* r = &lhs[(size_t)i * 4];
        r[0] = x[i] * 0 + 1; r[1] = x[i]; r[2] = r[1] * x[i]; r[3] = r[2] * x[i];
    }
    for (int64_t i = 0; i < n; ++i)
        for (int j = 0; j < 4; ++j) { const double v = lhs[(size_t)i * 4 + j]; scl[j] += v * v; }
    for (int j = 0; j < 4; ++j) { scl[j] = sqrt(scl[j]); if (scl[j] == 0) scl[j] = 1; }
    for (int64_t i = 0; i < n; ++i)
        for (int j = 0; j < 4; ++j) lhs[(size_t)i * 4 + j] /= scl[j];
    const double rcond = (double)n * 2.220446049250313e-16;
    if (int e = lstsq(lhs.data(), n, 4, y, rcond, c4)) return e;
    for (int j = 0; j < 4; ++j) c4[j] /= scl[j];
    return 0;
}

// numpy.polynomial.polynomial.polyval(x, p) with p lowest power first: c0 = p3 + x*0; c0 = p[k] + c0*x
static inline double polyval_asc(const double* p, double x) {
    double c0 = p[3] + x * 0;
    c0 = p[2] + c0 * x;
    c0 = p[1] + c0 * x;
    c0 = p[0] + c0 * x;
    return c0;
}

// np.polyval(p, x) with p highest power first: y = 0; y = y*x + pk
static inline double polyval_desc(const double* p, double x) {
    double y = 0.0;
    for (int k = 0; k < 4; ++k) y = y * x + p[k];
    return y;
}

// NumPy's _lerp for np.percentile's linear method
static inline double np_lerp(double a, double b, double gamma) {
    const double diff = b - a;
    return gamma >= 0.5 ? b - diff * (1 - gamma) : a + diff * gamma;
}

}  // namespace host
}  // namespace shg

using namespace shg::host;

extern "C" int shg_host_bind_lapack(void* dgelsd_ilp64) {
    g_dgelsd.store(reinterpret_cast<dgelsd_fn>(dgelsd_ilp64));
    return 0;
}

extern "C" int shg_host_set_mode_pick(shg_mode_pick_fn pick) {
    g_mode_pick.store(pick);
    return 0;
}

extern "C" int shg_host_lapack_bound(void) { return g_dgelsd.load() != nullptr; }

extern "C" int shg_host_polyfit3(const double* host_x, const double* host_y, int64_t n, double* host_coef4) {
    SHG_REQUIRE(host_x && host_y && host_coef4, SHG_E_ARG, "shg_host_polyfit3: null pointer");
    return polyfit3_desc(host_x, host_y, n, host_coef4);
}

// ---- a3: detect_bord on the row means (solex_util.py:165-172) ------------------------------------
extern "C" int shg_host_detect_bord(const double* host_row_means, int64_t n, int64_t* lb, int64_t* ub) {
    SHG_HOST_TIME("host detect_bord");
    SHG_REQUIRE(host_row_means && lb && ub && n > 0, SHG_E_ARG, "shg_host_detect_bord: bad argument");
    std::vector<double> s(host_row_means, host_row_means + n);
    std::nth_element(s.begin(), s.begin() + n / 2, s.end());                        // (a full sort of 2000 values took 47 us per scan)
    const double upper = s[n / 2];
    const double med = (n & 1) ? upper : (*std::max_element(s.begin(), s.begin() + n / 2) + upper) / 2.0;      // np.median
    const double thr = med / 5;
    int64_t first = 0, last = n - 1;                    // np.argmax of an all-False mask is 0: lb = 0, ub = n - 1
    for (int64_t i = 0; i < n; ++i) if (host_row_means[i] > thr) { first = i; break; }
    for (int64_t i = n - 1; i >= 0; --i) if (host_row_means[i] > thr) { last = i; break; }
    *lb = first;
    *ub = last;
    return 0;
}

// ---- a4: cubic fit of the line trace (solex_util.py:233-259) ---------------------------------------
// trace_blur: argmin of the blurred mean image over columns [12, iw-13), relative to column 12 (:231);
// trace_sharp: argmin of the mean image over all columns (:242).  y1, y2: the clipped sunlit range.
// Out: p4 (lowest power first), fit[ih][4] = [floor(c), c - floor(c), y, c], mask_good[y2-y1] (may be NULL).
extern "C" int shg_host_line_fit(const int32_t* host_trace_blur, const int32_t* host_trace_sharp, int64_t ih,
                                 int64_t y1, int64_t y2, int32_t blur_offset, double* host_p4, double* host_fit,
                                 uint8_t* host_mask_good) {
    SHG_HOST_TIME("host line_fit");
    SHG_REQUIRE(host_trace_blur && host_trace_sharp && host_p4 && host_fit, SHG_E_ARG, "shg_host_line_fit: null pointer");
    SHG_REQUIRE(ih > 0 && y1 >= 0 && y2 <= ih, SHG_E_ARG, "shg_host_line_fit: rows [%lld, %lld) outside the image", (long long)y1, (long long)y2);
    const int64_t n = std::max<int64_t>(y2 - y1, 0);
    std::vector<double> rows((size_t)n), mi((size_t)n), sharp((size_t)n), delta((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        rows[i] = (double)(y1 + i);
        mi[i] = (double)((int64_t)blur_offset + host_trace_blur[y1 + i]);
        sharp[i] = (double)host_trace_sharp[y1 + i];
    }
    double c[4], p[4];
    if (int e = polyfit3_desc(rows.data(), mi.data(), n, c)) return e;               // :233
    for (int k = 0; k < 4; ++k) p[k] = c[3 - k];
    for (int64_t i = 0; i < n; ++i) delta[i] = polyval_asc(p, rows[i]) - mi[i];     // :235
    const double stdv = np_std(delta.data(), n);
    std::vector<double> xs, ys;
    xs.reserve((size_t)n); ys.reserve((size_t)n);
    for (int64_t i = 0; i < n; ++i)
        if (fabs(delta[i] / stdv) < 3) { xs.push_back(rows[i]); ys.push_back(mi[i]); }   // :236-237 (NaN compares false)
    if (int e = polyfit3_desc(xs.data(), ys.data(), (int64_t)xs.size(), c)) return e;   // :238
    for (int k = 0; k < 4; ++k) p[k] = c[3 - k];

    // mode of the sharp residuals, rounded to 0.1 (:243-247): np.unique + np.argpartition(-counts, kth=2)[:2][0]
    std::vector<double> ds((size_t)n), rounded((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        ds[i] = polyval_asc(p, rows[i]) - sharp[i];
        rounded[i] = rint(ds[i] * 10.0) / 10.0;                                       // np.around(x, 1)
    }
    std::vector<double> values;
    std::vector<int64_t> counts;
    // np.unique(rounded, return_counts=True).  The rounded residuals are k / 10 for whole k within a few hundred of each other: counted
    // by k (k -> k / 10 never decreases, and k / 10 is the very double rint(x * 10) / 10 is); anything else -- a NaN, a wild
    // residual -- takes the sort (16 us of a scan's critical path for 2000 rows)
    bool counted = n > 0;
    double kmin = 0, kmax = 0;
    std::vector<double> ks((size_t)n);
    for (int64_t i = 0; i < n && counted; ++i) {
        const double k = ks[i] = rint(ds[i] * 10.0);
        if (!(fabs(k) < 1e9)) { counted = false; break; }
        kmin = i == 0 || k < kmin ? k : kmin;
        kmax = i == 0 || k > kmax ? k : kmax;
    }
    if (counted && kmax - kmin <= 65536.0) {
        // (-0.0 and 0.0 share a bin, as they share a run of the sort -- there the run's first element, whichever the sort left first,
        // is the value reported; here +0.0 unless every zero of the bin is -0.0.  The value only ever enters |ds - shift| < 5 below.)
        std::vector<int64_t> bins((size_t)(kmax - kmin) + 1, 0);
        bool plus_zero = false;
        for (int64_t i = 0; i < n; ++i) {
            ++bins[(size_t)(ks[i] - kmin)];
            plus_zero = plus_zero || (ks[i] == 0.0 && !std::signbit(ks[i]));
        }
        for (size_t b = 0; b < bins.size(); ++b)
            if (bins[b]) {
                const double k = (double)b + kmin;
                values.push_back(k == 0.0 && !plus_zero ? -0.0 : k / 10.0);
                counts.push_back(bins[b]);
            }
    } else {
        std::vector<double> sorted(rounded);
        std::sort(sorted.begin(), sorted.end());
        for (size_t i = 0; i < sorted.size(); ++i) {
            if (i == 0 || sorted[i] != sorted[i - 1]) { values.push_back(sorted[i]); counts.push_back(1); }   // NaNs stay apart, as in NumPy < 1.21; equal_nan groups them since
            else ++counts.back();
        }
    }
    if (values.size() < 3) {
        shg::set_error("kth(=2) out of bounds (%zu)", values.size());
        return SHG_E_VALUE;
    }
    // np.argpartition(-counts, kth=2)[:2][0] is ONE of the two most frequent values: which one is up to NumPy's
    // selection algorithm (scalar introselect or the x86-simd-sort kernels, by CPU).  The binding registers NumPy's
    // own argpartition for this one decision (shg_host_set_mode_pick); without it: the first most frequent value,
    // which is what the scalar introselect returns.
    size_t best = 0;
    for (size_t i = 1; i < counts.size(); ++i) if (counts[i] > counts[best]) best = i;
    if (mode_pick_fn pick = g_mode_pick.load()) {
        std::vector<int64_t> neg(counts.size());
        for (size_t i = 0; i < counts.size(); ++i) neg[i] = -counts[i];
        const int64_t got = pick(neg.data(), (int64_t)neg.size());
        if (got < 0 || got >= (int64_t)counts.size()) {
            shg::set_error("shg_host_line_fit: the registered mode picker returned %lld for %zu values", (long long)got, counts.size());
            return SHG_E_RUNTIME;
        }
        best = (size_t)got;
    }
    const double shift = values[best];
    xs.clear(); ys.clear();
    for (int64_t i = 0; i < n; ++i) {
        const bool good = fabs(ds[i] - shift) < 5;                                     // :253
        if (host_mask_good) host_mask_good[i] = good ? 1 : 0;
        if (good) { xs.push_back(rows[i]); ys.push_back(sharp[i]); }
    }
    if (int e = polyfit3_desc(xs.data(), ys.data(), (int64_t)xs.size(), c)) return e;   // :255
    for (int k = 0; k < 4; ++k) { p[k] = c[3 - k]; host_p4[k] = p[k]; }
    for (int64_t y = 0; y < ih; ++y) {                                                  // :258-259
        const double cv = polyval_asc(p, (double)y);
        const double fl = floor(cv);
        double* f = host_fit + y * 4;
        f[0] = fl; f[1] = cv - fl; f[2] = (double)y; f[3] = cv;
    }
    return 0;
}

// ---- a5: clamped sample columns and weights (solex_util.py:113-123) --------------------------------
extern "C" int shg_host_column_plan(const double* host_fit, int64_t ih, int64_t iw, const int32_t* host_shifts,
                                    int n_shifts, int32_t* host_ind_l, double* host_lw, double* host_rw) {
    SHG_HOST_TIME("host column_plan");
    SHG_REQUIRE(host_fit && host_shifts && host_ind_l && host_lw && host_rw, SHG_E_ARG, "shg_host_column_plan: null pointer");
    SHG_REQUIRE(ih > 0 && iw >= 2 && n_shifts > 0, SHG_E_ARG, "shg_host_column_plan: bad size");
    for (int s = 0; s < n_shifts; ++s)
        for (int64_t y = 0; y < ih; ++y) {
            const double v = host_fit[y * 4] + 1.0 * (double)host_shifts[s];          // fit[:,0] + np.ones(ih)*shift
            int64_t col = (int64_t)v;                                                   // .astype(int): truncation
            if (!(v == v) || v >= 9.2e18 || v <= -9.2e18) col = INT64_MIN;             // NumPy's cast of NaN / out of range
            if (col < 0) col = 0;
            if (col > iw - 2) col = iw - 2;
            host_ind_l[(int64_t)s * ih + y] = (int32_t)col;
        }
    for (int64_t y = 0; y < ih; ++y) {
        host_lw[y] = 1.0 - host_fit[y * 4 + 1];
        host_rw[y] = 1.0 - host_lw[y];
    }
    return 0;
}

// ---- a8: get_flood_image's threshold (ellipse_to_circle.py:159-225) ----------------------------------
// total = np.sum(image); over data = blurred[blurred < very_bright]: mn, mx, counts = np.histogram(data, 20)[0].
extern "C" int shg_host_flood_threshold(double total, int64_t h, int64_t w, double mn, double mx,
                                        const int64_t* host_counts20, double* thresh_out) {
    SHG_HOST_TIME("host flood_threshold");
    SHG_REQUIRE(host_counts20 && thresh_out && h > 0 && w > 0, SHG_E_ARG, "shg_host_flood_threshold: bad argument");
    const double thresh = 0.9 * total / (double)(h * w);
    if (mn == mx) { mn -= 0.5; mx += 0.5; }                      // np.histogram's range for constant data
    double bins[21];
    const double step = (mx - mn) / 20.0;                        // np.linspace(mn, mx, 21)
    for (int i = 0; i < 21; ++i) bins[i] = (double)i * step + mn;
    bins[20] = mx;
    // Polynomial.fit(bins[1:], n, 3).convert().coef: fit on the domain mapped to [-1, 1], then back
    double x[20], y[20];
    const double lo = bins[1], hi = bins[20];
    double d0 = lo, d1 = hi;
    if (d0 == d1) { d0 -= 1; d1 += 1; }
    const double oldlen = d1 - d0;
    const double off = (d1 * -1.0 - d0 * 1.0) / oldlen, scl = 2.0 / oldlen;     // pu.mapparms(dom, [-1, 1])
    for (int i = 0; i < 20; ++i) { x[i] = off + scl * bins[i + 1]; y[i] = (double)host_counts20[i]; }
    double cf[4];
    if (int e = polyfit3_asc(x, y, 20, cf)) return e;
    // coefficients of cf(off + scl*x) by Horner's rule on coefficient arrays
    double acc[4] = {cf[3], 0, 0, 0};
    int len = 1;
    for (int k = 2; k >= 0; --k) {
        double nxt[4] = {0, 0, 0, 0};
        for (int j = 0; j <= len; ++j) {
            double t = 0;
            if (j < len) t = acc[j] * off;
            if (j > 0) t = (j < len) ? t + acc[j - 1] * scl : acc[j - 1] * scl;
            nxt[j] = t;
        }
        ++len;
        nxt[0] += cf[k];
        for (int j = 0; j < 4; ++j) acc[j] = nxt[j];
    }
    const double d = acc[0], c = acc[1], b = acc[2], a = acc[3];
    (void)d;
    const double disc = 4 * b * b - 12 * a * c;
    const double thresh2 = disc >= 0 ? (-2 * b + sqrt(disc)) / (6 * a) : thresh;
    int start_i = -1;
    for (int i = 0; i < 20; ++i)
        if (bins[i] <= thresh2 && thresh2 < bins[i + 1]) start_i = i;
    if (start_i == -1) { *thresh_out = thresh; return 0; }
    int i = start_i;
    while (0 < i && i < 19) {
        if (host_counts20[i - 1] < host_counts20[i]) --i;
        else if (host_counts20[i + 1] < host_counts20[i]) ++i;
        else break;
    }
    if (i >= 1) --i;
    *thresh_out = bins[i];
    return 0;
}

// ---- a8: limb points from the labelled canny edges (ellipse_to_circle.py:251-291) ----------------------
// idx[m]: edge pixels y*w + x in raster order; root[m]: smallest linear index of each pixel's 8-connected
// component (sorting the distinct roots gives scipy.ndimage.label's numbering).  The NUM_REG = 2 largest regions
// (picked by size VALUE, list.index semantics: equal sizes resolve to the first such region), those of them that
// own a vertex of the convex hull of their union, rows cropped by 1.7 % top and bottom.
// out_sel[m]: 1 where the pixel is a limb point.  Returns SHG_E_QHULL where scipy.spatial.ConvexHull raises.
extern "C" int shg_host_limb_points(const int32_t* host_idx, const int32_t* host_root, int64_t m, int64_t h, int64_t w,
                                    uint8_t* host_out_sel, int64_t* n_selected) {
    SHG_HOST_TIME("host limb_points");
    SHG_REQUIRE(host_idx && host_root && host_out_sel && n_selected && h > 0 && w > 0, SHG_E_ARG, "shg_host_limb_points: bad argument");
    SHG_REQUIRE(m > 0, SHG_E_RUNTIME, "ellipse fit: could not find any edges of the solar disk");
    std::vector<int32_t> uniq(host_root, host_root + m);
    std::sort(uniq.begin(), uniq.end());
    uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
    const int nf = (int)uniq.size();
    std::vector<int32_t> lab((size_t)m);
    std::vector<int64_t> sizes((size_t)nf + 1, 0);
    for (int64_t i = 0; i < m; ++i) {
        lab[i] = (int32_t)(std::lower_bound(uniq.begin(), uniq.end(), host_root[i]) - uniq.begin()) + 1;
        ++sizes[lab[i]];
    }
    sizes[0] = -1;
    std::vector<int64_t> desc(sizes);
    std::sort(desc.begin(), desc.end(), std::greater<int64_t>());
    int chosen[2], n_chosen = std::min(nf, 2);
    for (int k = 0; k < n_chosen; ++k)
        chosen[k] = (int)(std::find(sizes.begin(), sizes.end(), desc[k]) - sizes.begin());
    std::vector<uint8_t> member((size_t)nf + 1, 0);
    for (int k = 0; k < n_chosen; ++k) member[chosen[k]] = 1;

    struct Pt { int64_t r, c; int32_t lab; };
    std::vector<Pt> pts;
    pts.reserve((size_t)m);
    int64_t r_min = INT64_MAX, r_max = INT64_MIN;
    for (int64_t i = 0; i < m; ++i)
        if (member[lab[i]]) {
            const int64_t r = host_idx[i] / w, c = host_idx[i] % w;
            pts.push_back({r, c, lab[i]});
            r_min = std::min(r_min, r);
            r_max = std::max(r_max, r);
        }
    // convex hull of the selected pixels (Andrew's monotone chain, exact integer arithmetic): with integer coordinates
    // the strict vertices are exactly the vertices Qhull reports
    std::vector<Pt> s(pts);
    std::sort(s.begin(), s.end(), [](const Pt& a, const Pt& b) { return a.r != b.r ? a.r < b.r : a.c < b.c; });
    auto cross = [](const Pt& o, const Pt& a, const Pt& b) { return (a.r - o.r) * (b.c - o.c) - (a.c - o.c) * (b.r - o.r); };
    std::vector<Pt> hull(2 * s.size() + 2);
    size_t k = 0;
    for (size_t i = 0; i < s.size(); ++i) {
        while (k >= 2 && cross(hull[k - 2], hull[k - 1], s[i]) <= 0) --k;
        hull[k++] = s[i];
    }
    for (size_t i = s.size() - 1, t = k + 1; i > 0; --i) {
        while (k >= t && cross(hull[k - 2], hull[k - 1], s[i - 1]) <= 0) --k;
        hull[k++] = s[i - 1];
    }
    if (k > 1) --k;
    if (s.size() < 3 || k < 3) {
        shg::set_error("QH6154 / QH6013: the limb pixels are collinear or fewer than three (Qhull cannot build an initial simplex)");
        return SHG_E_QHULL;
    }
    std::vector<uint8_t> on_hull((size_t)nf + 1, 0);
    for (size_t i = 0; i < k; ++i) on_hull[hull[i].lab] = 1;
    std::vector<uint8_t> keep((size_t)nf + 1, 0);
    for (int c = 0; c < n_chosen; ++c) if (on_hull[chosen[c]]) keep[chosen[c]] = 1;
    const double dx = (double)(r_max - r_min);
    const double crop = 0.017;
    int64_t ra = (int64_t)((double)r_min + dx * crop), rb = (int64_t)((double)r_max - dx * crop);   // int(): truncation
    ra = std::min(std::max<int64_t>(ra, 0), h);                  // slice semantics of mask[ra:rb, :] (both are >= 0 here)
    rb = std::min(std::max<int64_t>(rb, 0), h);
    int64_t cnt = 0;
    for (int64_t i = 0; i < m; ++i) {
        const int64_t r = host_idx[i] / w;
        const uint8_t v = keep[lab[i]] && r >= ra && r < rb;
        host_out_sel[i] = v;
        cnt += v;
    }
    *n_selected = cnt;
    return 0;
}

// ---- a9: chord bounds of the transversalium rows (solex_util.py:384-391) ---------------------------------
// xa, xb [max(y2-y1,1)] (entry 0 unused): NumPy-normalised slice [a, b) of row y1+i.
extern "C" int shg_host_chord_bounds(double cx, double cy, double r, double b0, double b2, int64_t y1, int64_t y2,
                                     int64_t w, int32_t* host_xa, int32_t* host_xb) {
    SHG_HOST_TIME("host chord_bounds");
    SHG_REQUIRE(host_xa && host_xb && w > 0, SHG_E_ARG, "shg_host_chord_bounds: bad argument");
    const int64_t count = std::max<int64_t>(y2 - y1, 1);
    for (int64_t i = 0; i < count; ++i) host_xa[i] = host_xb[i] = 0;
    for (int64_t y = y1 + 1; y < y2; ++y) {
        const double v = r * r - ((double)y - cy) * ((double)y - cy);
        if (v < 0) { shg::set_error("transversalium: row outside the disk circle (complex chord length)"); return SHG_E_TYPE; }
        // math.floor((r**2 - (y-cy)**2) ** 0.5): Python's ** is libm's pow, which need not round as sqrt does -- but both lie within an
        // ulp of the root, so their floors can only differ when the root is within a few ulp of a whole number: pow is asked there
        // (and nowhere else: it took 18 us of a scan's 24 us here, on every scan's critical path)
        double root = sqrt(v);
        const double near = rint(root);
        if (fabs(root - near) <= 4.0 * 2.220446049250313e-16 * (near > 1.0 ? near : 1.0)) root = shg::host::libm_pow(v, 0.5);
        const double dx = floor(root);
        int64_t a = (int64_t)ceil(std::max(cx - dx, b0)), b = (int64_t)floor(std::min(cx + dx, b2));
        a = a < 0 ? std::max<int64_t>(a + w, 0) : std::min(a, w);   // slice(a, b).indices(w)
        b = b < 0 ? std::max<int64_t>(b + w, 0) : std::min(b, w);
        host_xa[y - y1] = (int32_t)a;
        host_xb[y - y1] = (int32_t)std::max(a, b);
    }
    return 0;
}

// ---- a9: row correction factors from the robust log-ratios (solex_util.py:400-404, 456-472) ----------------
// ratios[k][n], interior[k][n] = correlate1d(ratios, taps[::-1]) (the interior of savgol_filter, from the GPU, or NULL
// to compute it here), taps[window] = scipy.signal.savgol_coeffs(window, 3).  out[k][n] = 1 + (exp(-cumsum(r - trend -
// mean)) - 1) * taper, or the untapered correction when tapered == 0 (the stubborn branch, :404).
extern "C" int shg_host_transversalium_factors(const double* host_ratios, const double* host_interior, int64_t k, int64_t n,
                                               const double* host_taps, int64_t window, int tapered, double* host_out) {
    SHG_HOST_TIME("host transversalium_factors");
    SHG_REQUIRE(host_ratios && host_taps && host_out && k > 0 && n > 0, SHG_E_ARG, "shg_host_transversalium_factors: bad argument");
    if (window > n) { shg::set_error("If mode is 'interp', window_length must be less than or equal to the size of x."); return SHG_E_VALUE; }
    SHG_REQUIRE(window >= 1 && (window & 1), SHG_E_VALUE, "window_length must be odd and positive, got %lld", (long long)window);
    const int64_t half = window / 2;
    std::vector<double> trend((size_t)n), xs((size_t)window), det((size_t)n), taper((size_t)n, 1.0);
    for (int64_t i = 0; i < window; ++i) xs[i] = (double)i;
    if (tapered) {                                                   // the piecewise taper t(x), a = 0.05 (:460-470)
        const double a = 0.05;
        for (int64_t x = 0; x < n; ++x) {
            if ((double)x < a * n / 2) taper[x] = 0.5 * (1 - cos(2 * M_PI * (double)x / (a * n)));
            else break;
        }
        for (int64_t x = n - 1; x >= 0; --x) {
            if ((double)x > (double)n / 2 && (double)(n - x) < a * n / 2) taper[x] = 0.5 * (1 - cos(2 * M_PI * (double)(n - x) / (a * n)));
            else break;
        }
    }
    for (int64_t row = 0; row < k; ++row) {
        const double* y = host_ratios + row * n;
        if (host_interior) {
            memcpy(trend.data(), host_interior + row * n, sizeof(double) * (size_t)n);
        } else {
            // scipy.ndimage.correlate1d(y, taps[::-1], mode='constant') in NI_Correlate1D's order of operations
            // (the same three forms as k_correlate1d_rows): w[j], j = -half..half, centred weights
            std::vector<double> wv((size_t)window);
            for (int64_t i = 0; i < window; ++i) wv[i] = host_taps[window - 1 - i];
            const double* wc = wv.data() + half;
            bool sym = true, anti = true;
            for (int64_t i = 1; i <= half; ++i) {
                if (fabs(wc[i] - wc[-i]) > 2.220446049250313e-16) sym = false;
                if (fabs(wc[i] + wc[-i]) > 2.220446049250313e-16) anti = false;
            }
            auto at = [&](int64_t j) { return (j >= 0 && j < n) ? y[j] : 0.0; };
            for (int64_t x = 0; x < n; ++x) {
                double t;
                if (sym) {
                    t = at(x) * wc[0];
                    for (int64_t j = -half; j < 0; ++j) t += (at(x + j) + at(x - j)) * wc[j];
                } else if (anti) {
                    t = at(x) * wc[0];
                    for (int64_t j = -half; j < 0; ++j) t += (at(x + j) - at(x - j)) * wc[j];
                } else {
                    t = at(x + half) * wc[half];
                    for (int64_t j = -half; j < half; ++j) t += at(x + j) * wc[j];
                }
                trend[x] = t;
            }
        }
        double c4[4];                                                // the two edges: SciPy's _fit_edge
        if (int e = polyfit3_desc(xs.data(), y, window, c4)) return e;
        for (int64_t i = 0; i < half; ++i) trend[i] = polyval_desc(c4, (double)i);
        if (int e = polyfit3_desc(xs.data(), y + (n - window), window, c4)) return e;
        for (int64_t i = 0; i < half; ++i) trend[n - half + i] = polyval_desc(c4, (double)(window - half + i));
        for (int64_t i = 0; i < n; ++i) det[i] = y[i] - trend[i];
        const double mu = np_mean(det.data(), n);
        double run = 0;
        double* out = host_out + row * n;
        for (int64_t i = 0; i < n; ++i) {
            run += det[i] - mu;                                      // np.cumsum
            const double corr = exp(-run);
            out[i] = tapered ? 1.0 + (corr - 1.0) * taper[i] : corr;
        }
    }
    return 0;
}

// np.percentile(values, q) (method 'linear') on n values: the two 0-based order statistics and the _lerp weight
extern "C" int shg_host_percentile_plan(int64_t n, double q, int64_t* rank_lo, int64_t* rank_hi, double* gamma) {
    SHG_REQUIRE(rank_lo && rank_hi && gamma && n > 0, SHG_E_ARG, "shg_host_percentile_plan: bad argument");
    const double virt = (double)(n - 1) * (q / 100.0);
    const double fl = floor(virt);
    int64_t lo = (int64_t)fl;
    lo = std::min(std::max<int64_t>(lo, 0), n - 1);
    *rank_lo = lo;
    *rank_hi = std::min(lo + 1, n - 1);
    *gamma = virt - fl;
    return 0;
}

extern "C" double shg_host_lerp(double a, double b, double gamma) { return np_lerp(a, b, gamma); }

// ---- a8 / a6 / a7: the limb geometry with NumPy's own BLAS / LAPACK calls --------------------------------
// LsqEllipse().fit (Halir & Flusser), two_step, get_correction_matrix and correct_image's geometry
// (ellipse_to_circle.py:39-122).  phi and ratio steer every sample position of the warp: a 1e-12 difference
// already moves pixels by a grey level, so the scatter matrices, the 3x3 inverses and the eigen-decomposition
// must come out of the very routines NumPy calls, called the way NumPy's matmul / inv / eig call them
// (umath/matmul.c.src: row-major cblas_dsyrk for A @ A.T, cblas_dgemm with the transposition flags the operand
// strides imply, cblas_dgemv for a one-column right operand; umath_linalg: dgesv against the identity for inv,
// dgeev with a workspace query for eig).  shg_host_bind_blas hands over those five entry points of the OpenBLAS
// NumPy loaded.  What cannot be shared are libm-level transcendentals (cos / sin / arctan of a scalar, which NumPy
// may serve from SVML): one-ulp differences there change a sample position by ~1e-13 px, far below a grey level.
namespace shg {
namespace host {

typedef void (*cblas_dgemm_fn)(int order, int ta, int tb, int64_t m, int64_t n, int64_t k, double alpha, const double* a,
                               int64_t lda, const double* b, int64_t ldb, double beta, double* c, int64_t ldc);
typedef void (*cblas_dsyrk_fn)(int order, int uplo, int trans, int64_t n, int64_t k, double alpha, const double* a,
                               int64_t lda, double beta, double* c, int64_t ldc);
typedef void (*cblas_dgemv_fn)(int order, int trans, int64_t m, int64_t n, double alpha, const double* a, int64_t lda,
                               const double* x, int64_t incx, double beta, double* y, int64_t incy);
typedef void (*dgesv_fn)(const int64_t* n, const int64_t* nrhs, double* a, const int64_t* lda, int64_t* ipiv, double* b,
                         const int64_t* ldb, int64_t* info);
typedef void (*dgeev_fn)(const char* jobvl, const char* jobvr, const int64_t* n, double* a, const int64_t* lda, double* wr,
                         double* wi, double* vl, const int64_t* ldvl, double* vr, const int64_t* ldvr, double* work,
                         const int64_t* lwork, int64_t* info, size_t, size_t);
struct Blas {
    cblas_dgemm_fn gemm;
    cblas_dsyrk_fn syrk;
    cblas_dgemv_fn gemv;
    dgesv_fn gesv;
    dgeev_fn geev;
};
static std::atomic<const Blas*> g_blas{nullptr};
static Blas g_blas_store;

// ---- built-in stand-ins, for hosts where NumPy's own routines cannot be found ----------------------------------------------
// shg_host_bind_blas gets its five entry points from the OpenBLAS that NumPy's wheel bundles (_lib.py); a NumPy built against
// another BLAS (MKL, a system OpenBLAS, Accelerate) does not export them under names the binding knows.  The limb geometry
// then runs on the routines below: the same algebra on the same operands, plain loops instead of OpenBLAS's kernels -- results
// agree with NumPy's to ~1e-12 relative (tests/test_hostmath_cpu.py), they are no longer bit-identical.  Same signatures as
// the cblas / LAPACK entry points, only the cases the callers above use (n <= 3 for gesv / geev).
namespace builtin {
inline double at(const double* a, int64_t ld, bool row_major, bool trans, int64_t i, int64_t j) {      // op(A)[i][j]
    if (trans) { const int64_t t = i; i = j; j = t; }
    return row_major ? a[i * ld + j] : a[i + j * ld];
}
void gemm(int order, int ta, int tb, int64_t m, int64_t n, int64_t k, double alpha, const double* a, int64_t lda, const double* b,
          int64_t ldb, double beta, double* c, int64_t ldc) {
    const bool rm = order == 101;
    for (int64_t i = 0; i < m; ++i)
        for (int64_t j = 0; j < n; ++j) {
            long double acc = 0;
            for (int64_t l = 0; l < k; ++l) acc += (long double)at(a, lda, rm, ta == 112, i, l) * at(b, ldb, rm, tb == 112, l, j);
            double& out = rm ? c[i * ldc + j] : c[i + j * ldc];
            out = (double)(alpha * acc) + (beta != 0.0 ? beta * out : 0.0);
        }
}
void syrk(int order, int uplo, int trans, int64_t n, int64_t k, double alpha, const double* a, int64_t lda, double beta, double* c, int64_t ldc) {
    const bool rm = order == 101;
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < n; ++j) {
            const bool upper = j >= i;
            if ((uplo == 121) != upper && i != j) continue;
            long double acc = 0;
            for (int64_t l = 0; l < k; ++l) acc += (long double)at(a, lda, rm, trans == 112, i, l) * at(a, lda, rm, trans == 112, j, l);
            double& out = rm ? c[i * ldc + j] : c[i + j * ldc];
            out = (double)(alpha * acc) + (beta != 0.0 ? beta * out : 0.0);
        }
}
void gemv(int order, int trans, int64_t m, int64_t n, double alpha, const double* a, int64_t lda, const double* x, int64_t incx, double beta,
          double* y, int64_t incy) {
    const bool rm = order == 101, tr = trans == 112;
    const int64_t rows = tr ? n : m, cols = tr ? m : n;
    for (int64_t i = 0; i < rows; ++i) {
        long double acc = 0;
        for (int64_t j = 0; j < cols; ++j) acc += (long double)at(a, lda, rm, tr, i, j) * x[j * incx];
        y[i * incy] = (double)(alpha * acc) + (beta != 0.0 ? beta * y[i * incy] : 0.0);
    }
}
// column-major A (n x n), B (n x nrhs): Gaussian elimination with partial pivoting, the solution in B
void gesv(const int64_t* pn, const int64_t* pnrhs, double* a, const int64_t* plda, int64_t* ipiv, double* b, const int64_t* pldb, int64_t* info) {
    const int64_t n = *pn, nrhs = *pnrhs, lda = *plda, ldb = *pldb;
    *info = 0;
    for (int64_t c = 0; c < n; ++c) {
        int64_t p = c;
        for (int64_t r = c + 1; r < n; ++r)
            if (fabs(a[r + c * lda]) > fabs(a[p + c * lda])) p = r;
        ipiv[c] = p + 1;
        if (a[p + c * lda] == 0.0) { *info = c + 1; return; }
        if (p != c) {
            for (int64_t j = 0; j < n; ++j) std::swap(a[c + j * lda], a[p + j * lda]);
            for (int64_t j = 0; j < nrhs; ++j) std::swap(b[c + j * ldb], b[p + j * ldb]);
        }
        for (int64_t r = c + 1; r < n; ++r) {
            const double f = a[r + c * lda] / a[c + c * lda];
            for (int64_t j = c; j < n; ++j) a[r + j * lda] -= f * a[c + j * lda];
            for (int64_t j = 0; j < nrhs; ++j) b[r + j * ldb] -= f * b[c + j * ldb];
        }
    }
    for (int64_t j = 0; j < nrhs; ++j)
        for (int64_t r = n - 1; r >= 0; --r) {
            long double sacc = b[r + j * ldb];
            for (int64_t c = r + 1; c < n; ++c) sacc -= (long double)a[r + c * lda] * b[c + j * ldb];
            b[r + j * ldb] = (double)(sacc / a[r + r * lda]);
        }
}
// right eigenvectors of a real 3 x 3 matrix (column major): roots of the characteristic cubic (closed form, polished by Newton
// steps in extended precision), every eigenvector as the cross product of the two most independent rows of A - lambda I, scaled
// to unit length.  A complex pair is reported through wi (its vectors are not formed: the callers reject that case).
void geev(const char*, const char*, const int64_t* pn, double* a, const int64_t* plda, double* wr, double* wi, double*, const int64_t*,
          double* vr, const int64_t* pldvr, double* work, const int64_t* lwork, int64_t* info, size_t, size_t) {
    *info = 0;
    if (*lwork == -1) { work[0] = 1.0; return; }
    if (*pn != 3) { *info = -3; return; }
    const int64_t lda = *plda, ldv = *pldvr;
    long double m[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m[i][j] = a[i + j * lda];
    long double scale = 0;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) scale = fabsl(m[i][j]) > scale ? fabsl(m[i][j]) : scale;
    if (scale == 0) scale = 1;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m[i][j] /= scale;
    const long double tr = m[0][0] + m[1][1] + m[2][2];
    const long double c1 = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) + (m[0][0] * m[2][2] - m[0][2] * m[2][0]) + (m[1][1] * m[2][2] - m[1][2] * m[2][1]);
    const long double det = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
                            m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
    // t^3 + p t + q = 0 with lambda = t + tr / 3
    const long double sh = tr / 3, p = c1 - tr * tr / 3, q = -2 * tr * tr * tr / 27 + tr * c1 / 3 - det;
    const long double disc = q * q / 4 + p * p * p / 27;
    long double lam[3];
    int n_real = 3;
    if (disc > 0) {
        const long double sq = sqrtl(disc), u = cbrtl(-q / 2 + sq), v = cbrtl(-q / 2 - sq);
        lam[0] = u + v + sh;
        n_real = 1;
        const long double re = -(u + v) / 2 + sh, im = (u - v) * sqrtl(3.0L) / 2;
        if (fabsl(im) <= 1e-15L * (fabsl(re) + 1)) { lam[1] = lam[2] = re; n_real = 3; }
        else { lam[1] = re; lam[2] = im; }
    } else {
        const long double r = sqrtl(-p / 3), arg = p == 0 ? 0 : (3 * q / (2 * p)) / r;
        const long double th = acosl(arg > 1 ? 1 : (arg < -1 ? -1 : arg)) / 3;
        const long double pi = 3.141592653589793238462643383279502884L;
        for (int k = 0; k < 3; ++k) lam[k] = 2 * r * cosl(th - 2 * pi * k / 3) + sh;
    }
    for (int k = 0; k < (n_real == 3 ? 3 : 1); ++k)                          // Newton polish on the characteristic polynomial
        for (int it = 0; it < 4; ++it) {
            const long double x = lam[k], f = ((x - tr) * x + c1) * x - det, df = (3 * x - 2 * tr) * x + c1;
            if (df == 0) break;
            lam[k] = x - f / df;
        }
    if (n_real == 3) {                                                       // LAPACK's order is not promised; ascending is as good as any
        for (int k = 0; k < 3; ++k) { wr[k] = (double)(lam[k] * scale); wi[k] = 0.0; }
    } else {
        wr[0] = (double)(lam[0] * scale); wi[0] = 0.0;
        wr[1] = wr[2] = (double)(lam[1] * scale);
        wi[1] = (double)(lam[2] * scale); wi[2] = -wi[1];
    }
    for (int k = 0; k < 3; ++k) {
        double* v = vr + k * ldv;
        v[0] = v[1] = v[2] = 0.0;
        if (wi[k] != 0.0) continue;
        long double b[3][3];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) b[i][j] = m[i][j] - (i == j ? lam[k] : 0);
        long double best[3] = {0, 0, 0}, best_n = -1;
        for (int r0 = 0; r0 < 3; ++r0)
            for (int r1 = r0 + 1; r1 < 3; ++r1) {
                const long double c[3] = {b[r0][1] * b[r1][2] - b[r0][2] * b[r1][1], b[r0][2] * b[r1][0] - b[r0][0] * b[r1][2],
                                          b[r0][0] * b[r1][1] - b[r0][1] * b[r1][0]};
                const long double nn = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
                if (nn > best_n) { best_n = nn; best[0] = c[0]; best[1] = c[1]; best[2] = c[2]; }
            }
        if (best_n <= 0) { best[0] = 1; best[1] = best[2] = 0; best_n = 1; }    // A - lambda I = 0: any vector
        const long double inv = 1 / sqrtl(best_n);
        for (int i = 0; i < 3; ++i) v[i] = (double)(best[i] * inv);
    }
}
const Blas kBuiltin = {gemm, syrk, gemv, gesv, geev};
}  // namespace builtin

static inline const Blas& active_blas() {
    const Blas* b = g_blas.load();
    return b ? *b : builtin::kBuiltin;
}

constexpr int kRowMajor = 101, kColMajor = 102, kNoTrans = 111, kTrans = 112, kUpper = 121;

// A (m x k, row major) @ B.  b_is_transposed_view: B is the .T view of a row-major (n x k) array `b`.
static void matmul(const Blas& bl, const double* a, const double* b, double* c, int64_t m, int64_t k, int64_t n, bool b_is_transposed_view) {
    if (b_is_transposed_view) bl.gemm(kRowMajor, kNoTrans, kTrans, m, n, k, 1.0, a, k, b, k, 0.0, c, n);
    else bl.gemm(kRowMajor, kNoTrans, kNoTrans, m, n, k, 1.0, a, k, b, n, 0.0, c, n);
}

// A @ A.T for a row-major (n x k) array: NumPy's syrk path, upper triangle mirrored
static void gram(const Blas& bl, const double* a, double* c, int64_t n, int64_t k) {
    for (int64_t i = 0; i < n * n; ++i) c[i] = 0.0;
    bl.syrk(kRowMajor, kUpper, kNoTrans, n, k, 1.0, a, k, 0.0, c, n);
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = i + 1; j < n; ++j) c[j * n + i] = c[i * n + j];
}

// np.linalg.inv of a row-major n x n matrix (n <= 3)
static int np_inv(const Blas& bl, const double* a, double* out, int64_t n) {
    double af[9], bf[9];
    int64_t ipiv[3], info = 0;
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < n; ++j) { af[i + j * n] = a[i * n + j]; bf[i + j * n] = i == j ? 1.0 : 0.0; }
    bl.gesv(&n, &n, af, &n, ipiv, bf, &n, &info);
    if (info != 0) { set_error("Singular matrix"); return SHG_E_LINALG; }
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < n; ++j) out[i * n + j] = bf[i + j * n];
    return 0;
}

// lsq-ellipse 2.0: LsqEllipse().fit(points).as_parameters() on points[n][2]
static int fit_ellipse_np(const Blas& bl, const double* pts, int64_t n, double center[2], double* width, double* height, double* phi) {
    if (n < 1) { set_error("index 0 is out of bounds for axis 0 with size 0"); return SHG_E_INDEX; }
    std::vector<double> d1((size_t)3 * n), d2((size_t)3 * n);               // np.vstack([...]) : rows x^2, xy, y^2 / x, y, 1
    for (int64_t i = 0; i < n; ++i) {
        const double x = pts[2 * i], y = pts[2 * i + 1];
        d1[i] = x * x; d1[n + i] = x * y; d1[2 * n + i] = y * y;
        d2[i] = x; d2[n + i] = y; d2[2 * n + i] = 1.0;
    }
    double S1[9], S2[9], S3[9], S3i[9], P[9], Q[9], R[9], C1i[9], M[9];
    gram(bl, d1.data(), S1, 3, n);                                          // D1.T @ D1
    matmul(bl, d1.data(), d2.data(), S2, 3, n, 3, true);                   // D1.T @ D2
    gram(bl, d2.data(), S3, 3, n);                                          // D2.T @ D2
    const double C1[9] = {0., 0., 2., 0., -1., 0., 2., 0., 0.};
    if (int e = np_inv(bl, S3, S3i, 3)) return e;
    if (int e = np_inv(bl, C1, C1i, 3)) return e;
    matmul(bl, S2, S3i, P, 3, 3, 3, false);                                // S2 @ inv(S3)
    matmul(bl, P, S2, Q, 3, 3, 3, true);                                   // ... @ S2.T
    for (int i = 0; i < 9; ++i) R[i] = S1[i] - Q[i];
    matmul(bl, C1i, R, M, 3, 3, 3, false);                                 // inv(C1) @ (...)
    // np.linalg.eig(M)
    double af[9], wr[3], wi[3], vr[9], vl[9], wq = 0;
    int64_t n3 = 3, lwork = -1, info = 0;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) af[i + 3 * j] = M[i * 3 + j];
    bl.geev("N", "V", &n3, af, &n3, wr, wi, vl, &n3, vr, &n3, &wq, &lwork, &info, 1, 1);
    if (info != 0) { set_error("eig: workspace query failed"); return SHG_E_LINALG; }
    lwork = (int64_t)wq;
    std::vector<double> work((size_t)std::max<int64_t>(lwork, 1));
    bl.geev("N", "V", &n3, af, &n3, wr, wi, vl, &n3, vr, &n3, work.data(), &lwork, &info, 1, 1);
    if (info != 0) { set_error("Eigenvalues did not converge"); return SHG_E_LINALG; }
    if (wi[0] != 0.0 || wi[1] != 0.0 || wi[2] != 0.0) { set_error("ellipse fit: complex eigenvalues (the limb points do not describe an ellipse)"); return SHG_E_RUNTIME; }
    // eigvec[i][j] = vr[i + 3 j]; cond = 4 * eigvec[0] * eigvec[2] - eigvec[1] ** 2; a1 = eigvec[:, cond > 0]
    int cols[3], kpos = 0;
    for (int j = 0; j < 3; ++j) {
        const double cond = 4 * (vr[0 + 3 * j] * vr[2 + 3 * j]) - libm_pow(vr[1 + 3 * j], 2.0);
        if (cond > 0) cols[kpos++] = j;
    }
    if (kpos == 0) { set_error("index 0 is out of bounds for axis 0 with size 0"); return SHG_E_INDEX; }
    double a1[9], S3n[9], S3ni[9], T1[9], a2[9];
    for (int i = 0; i < 3; ++i) for (int c = 0; c < kpos; ++c) a1[i * kpos + c] = vr[i + 3 * cols[c]];
    for (int i = 0; i < 9; ++i) S3n[i] = -S3[i];
    if (int e = np_inv(bl, S3n, S3ni, 3)) return e;
    matmul(bl, S3ni, S2, T1, 3, 3, 3, true);                               // inv(-S3) @ S2.T
    // ... @ a1: one column -> NumPy's gemv form (the row-major matrix read as column-major and transposed)
    if (kpos == 1) bl.gemv(kColMajor, kTrans, 3, 3, 1.0, T1, 3, a1, 1, 0.0, a2, 1);
    else matmul(bl, T1, a1, a2, 3, 3, kpos, false);
    double coef[6];                                                          // np.vstack([a1, a2]).ravel()[:6]
    for (int j = 0; j < 6; ++j) { const int r = j / kpos, c = j % kpos; coef[j] = r < 3 ? a1[r * kpos + c] : a2[(r - 3) * kpos + c]; }
    const double a = coef[0], b = coef[1] / 2., c = coef[2], d = coef[3] / 2., f = coef[4] / 2., g = coef[5];
    const double x0 = (c * d - b * f) / (libm_pow(b, 2.) - a * c);
    const double y0 = (a * f - b * d) / (libm_pow(b, 2.) - a * c);
    const double numerator = 2 * ((((a * libm_pow(f, 2.) + c * libm_pow(d, 2.)) + g * libm_pow(b, 2.)) - 2 * b * d * f) - a * c * g);
    const double root = sqrt(1 + 4 * b * b / ((a - c) * (a - c)));
    center[0] = x0;
    center[1] = y0;
    *width = sqrt(numerator / ((b * b - a * c) * ((c - a) * root - (c + a))));
    *height = sqrt(numerator / ((b * b - a * c) * ((a - c) * root - (c + a))));
    *phi = .5 * atan((2. * b) / (a - c));
    return 0;
}

static void rot2(double x, double m[4]) { m[0] = cos(x); m[1] = sin(x); m[2] = -sin(x); m[3] = cos(x); }

// get_correction_matrix(phi, r): (np.linalg.inv(correction_matrix), theta), ellipse_to_circle.py:39-50
static int correction_matrix_np(const Blas& bl, double phi, double r, double inv4[4], double* theta_out) {
    double rp[4], rm[4], dg[4] = {r, 0, 0, 1}, t[4], st[4], rt[4], cm[4];
    rot2(phi, rp);
    rot2(-phi, rm);
    matmul(bl, rp, dg, t, 2, 2, 2, false);
    matmul(bl, t, rm, st, 2, 2, 2, false);
    const double theta = atan(st[2] / st[0]);
    rot2(theta, rt);
    matmul(bl, rt, st, cm, 2, 2, 2, false);
    cm[2] = 0;
    const double dd = cm[3];
    for (int i = 0; i < 4; ++i) cm[i] /= dd;
    if (int e = np_inv(bl, cm, inv4, 2)) return e;
    *theta_out = theta;
    return 0;
}

}  // namespace host
}  // namespace shg

extern "C" int shg_host_bind_blas(void* cblas_dgemm_ilp64, void* cblas_dsyrk_ilp64, void* cblas_dgemv_ilp64, void* dgesv_ilp64,
                                  void* dgeev_ilp64) {
    if (!cblas_dgemm_ilp64 || !cblas_dsyrk_ilp64 || !cblas_dgemv_ilp64 || !dgesv_ilp64 || !dgeev_ilp64) {
        g_blas.store(nullptr);
        return 0;
    }
    g_blas_store.gemm = reinterpret_cast<cblas_dgemm_fn>(cblas_dgemm_ilp64);
    g_blas_store.syrk = reinterpret_cast<cblas_dsyrk_fn>(cblas_dsyrk_ilp64);
    g_blas_store.gemv = reinterpret_cast<cblas_dgemv_fn>(cblas_dgemv_ilp64);
    g_blas_store.gesv = reinterpret_cast<dgesv_fn>(dgesv_ilp64);
    g_blas_store.geev = reinterpret_cast<dgeev_fn>(dgeev_ilp64);
    g_blas.store(&g_blas_store);
    return 0;
}

extern "C" int shg_host_blas_bound(void) { return g_blas.load() != nullptr; }

// NumPy's own routines when they are bound (bit-identical geometry), the built-in stand-ins otherwise
#define SHG_NEED_BLAS(who) const Blas& bl = active_blas()

extern "C" int shg_host_fit_ellipse(const double* host_points, int64_t n, double* host_center2, double* width, double* height,
                                    double* phi) {
    SHG_REQUIRE(host_points && host_center2 && width && height && phi, SHG_E_ARG, "shg_host_fit_ellipse: null pointer");
    SHG_NEED_BLAS("shg_host_fit_ellipse");
    return fit_ellipse_np(bl, host_points, n, host_center2, width, height, phi);
}

extern "C" int shg_host_correction_matrix(double phi, double r, double* host_inv4, double* theta_out) {
    SHG_REQUIRE(host_inv4 && theta_out, SHG_E_ARG, "shg_host_correction_matrix: null pointer");
    SHG_NEED_BLAS("shg_host_correction_matrix");
    return correction_matrix_np(bl, phi, r, host_inv4, theta_out);
}

// two_step (ellipse_to_circle.py:62-91): fit, drop the points inside the ellipse by more than the largest outward
// residual, refit, bring phi within pi/4 of zero.  points[n][2] = (row, col).  kept[n] (may be NULL): 1 for the points of the
// second fit.  out: center[2] (row, col), height, phi, ratio, outline200 = return_fit(n_points=100) (may be NULL).
extern "C" int shg_host_two_step(const double* host_points, int64_t n, double* host_center2, double* height_out, double* phi_out,
                                 double* ratio_out, uint8_t* host_kept, int64_t* n_kept, double* host_outline200) {
    SHG_REQUIRE(host_points && host_center2 && height_out && phi_out && ratio_out && n_kept, SHG_E_ARG, "shg_host_two_step: null pointer");
    SHG_NEED_BLAS("shg_host_two_step");
    double center[2], width, height, phi, mat[4], theta;
    if (int e = fit_ellipse_np(bl, host_points, n, center, &width, &height, &phi)) return e;
    if (int e = correction_matrix_np(bl, phi, height / width, mat, &theta)) return e;
    std::vector<double> values((size_t)n);
    double vmax = -INFINITY;
    for (int64_t i = 0; i < n; ++i) {                                     // Xr = mat @ (points - center).T * height; norm - 1
        const double dx = host_points[2 * i] - center[0], dy = host_points[2 * i + 1] - center[1];
        const double xr = (mat[0] * dx + mat[1] * dy) * height, yr = (mat[2] * dx + mat[3] * dy) * height;
        values[i] = sqrt(xr * xr + yr * yr) - 1;
        if (values[i] > vmax) vmax = values[i];
    }
    std::vector<double> kept;
    kept.reserve((size_t)n * 2);
    for (int64_t i = 0; i < n; ++i) {
        const bool k = values[i] > -vmax;
        if (host_kept) host_kept[i] = k;
        if (k) { kept.push_back(host_points[2 * i]); kept.push_back(host_points[2 * i + 1]); }
    }
    *n_kept = (int64_t)kept.size() / 2;
    if (int e = fit_ellipse_np(bl, kept.data(), *n_kept, center, &width, &height, &phi)) return e;
    if (host_outline200) {
        const double step = (2 * M_PI - 0.0) / 99.0;                         // np.linspace(0, 2 * np.pi, 100)
        for (int i = 0; i < 100; ++i) {
            const double t = i == 99 ? 2 * M_PI : (double)i * step + 0.0;
            host_outline200[2 * i] = center[0] + width * cos(t) * cos(phi) - height * sin(t) * sin(phi);
            host_outline200[2 * i + 1] = center[1] + width * cos(t) * sin(phi) + height * sin(t) * cos(phi);
        }
    }
    double ratio = width / height;
    for (int it = 0; it < 2; ++it) {
        if (phi > M_PI / 4) { phi -= M_PI / 2; ratio = 1 / ratio; height = height / ratio; }
        if (phi < -M_PI / 4) { phi += M_PI / 2; ratio = 1 / ratio; height = height / ratio; }
    }
    host_center2[0] = center[0];
    host_center2[1] = center[1];
    *height_out = height;
    *phi_out = phi;
    *ratio_out = ratio;
    return 0;
}

// correct_image's geometry (ellipse_to_circle.py:100-122): mat3[9] (row major), inv_mat[4], origin[2], det, theta,
// out_h, out_w for an h x w image.
extern "C" int shg_host_warp_geometry(double phi, double ratio, int64_t h, int64_t w, double* host_mat3_9, double* host_inv4,
                                      double* host_origin2, double* det_out, double* theta_out, int64_t* out_h, int64_t* out_w) {
    SHG_REQUIRE(host_mat3_9 && host_inv4 && host_origin2 && det_out && theta_out && out_h && out_w, SHG_E_ARG,
                "shg_host_warp_geometry: null pointer");
    SHG_NEED_BLAS("shg_host_warp_geometry");
    double mat[4], inv[4];
    if (int e = correction_matrix_np(bl, phi, ratio, mat, theta_out)) return e;
    if (int e = np_inv(bl, mat, inv, 2)) return e;
    const double corners[8] = {0, 0, 0, (double)h, (double)w, 0, (double)w, (double)h};        // [4][2]
    double nc[8];                                                                                // (inv_mat @ corners.T) as [2][4]
    matmul(bl, inv, corners, nc, 2, 2, 4, true);
    double xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
    for (int i = 0; i < 4; ++i) {
        xmin = std::min(xmin, nc[i]); xmax = std::max(xmax, nc[i]);
        ymin = std::min(ymin, nc[4 + i]); ymax = std::max(ymax, nc[4 + i]);
    }
    const double m3[9] = {mat[0], mat[1], 0, mat[2], mat[3], 0, 0, 0, 1};
    const double tr[9] = {1, 0, xmin, 0, 1, ymin, 0, 0, 1};
    matmul(bl, m3, tr, host_mat3_9, 3, 3, 3, false);
    if (!(host_mat3_9[3] == 0 && host_mat3_9[4] == 1 && host_mat3_9[5] == 0 && host_mat3_9[6] == 0 && host_mat3_9[7] == 0 && host_mat3_9[8] == 1)) {
        shg::set_error("correct_image: the correction never moves rows (ellipse_to_circle.py:48-49); got rows [%g %g %g] [%g %g %g]",
                       host_mat3_9[3], host_mat3_9[4], host_mat3_9[5], host_mat3_9[6], host_mat3_9[7], host_mat3_9[8]);
        return SHG_E_RUNTIME;
    }
    for (int i = 0; i < 4; ++i) host_inv4[i] = inv[i];
    host_origin2[0] = xmin;
    host_origin2[1] = ymin;
    *det_out = mat[0] * mat[3] - mat[1] * mat[2];
    *out_h = (int64_t)ceil(ymax - ymin);
    *out_w = (int64_t)ceil(xmax - xmin);
    return 0;
}

// ellipse_to_circle after get_edge_list (ellipse_to_circle.py:303-314): two_step on the limb points X (row, col in disk
// pixels), correct_image's geometry for the h x w disk, the circle of the corrected image and the borders of the kept
// points.  host_geom16 = ellipse centre x, y, height, phi, ratio | circle cx, cy, r | borders[4] | mat3 row 0 (h00, h01,
// h02) | theta.  host_dims2 = (out_h, out_w).  host_kept [n] and host_outline200 may be NULL.
extern "C" int shg_host_limb_geometry(const double* host_points, int64_t n, int64_t h, int64_t w, double* host_geom16,
                                      int64_t* host_dims2, uint8_t* host_kept, int64_t* n_kept, double* host_outline200) {
    SHG_HOST_TIME("host limb_geometry");
    SHG_REQUIRE(host_points && host_geom16 && host_dims2 && n_kept, SHG_E_ARG, "shg_host_limb_geometry: null pointer");
    SHG_NEED_BLAS("shg_host_limb_geometry");
    std::vector<uint8_t> kept_local;
    if (!host_kept) { kept_local.resize((size_t)std::max<int64_t>(n, 1)); host_kept = kept_local.data(); }
    double center[2], height, phi, ratio;
    if (int e = shg_host_two_step(host_points, n, center, &height, &phi, &ratio, host_kept, n_kept, host_outline200)) return e;
    double mat3[9], inv[4], origin[2], det, theta;
    if (int e = shg_host_warp_geometry(phi, ratio, h, w, mat3, inv, origin, &det, &theta, &host_dims2[0], &host_dims2[1])) return e;
    const double cxy[2] = {center[1], center[0]};                             // (x, y): the swap at :305
    double nc[2];
    bl.gemv(kColMajor, kTrans, 2, 2, 1.0, inv, 2, cxy, 1, 0.0, nc, 1);       // inv_mat @ center
    double* g = host_geom16;
    g[0] = cxy[0]; g[1] = cxy[1]; g[2] = height; g[3] = phi; g[4] = ratio;
    g[5] = nc[0] - origin[0];
    g[6] = nc[1] - origin[1];
    g[7] = height * sqrt(fabs(ratio / det));
    // borders: (np.linalg.inv(mat3) @ X_f3.T).T over the kept points, X_f3 = (x, y, 1)
    double inv3[9];
    if (int e = np_inv(bl, mat3, inv3, 3)) return e;
    const int64_t m = *n_kept;
    std::vector<double> xf3((size_t)std::max<int64_t>(m, 1) * 3), tr((size_t)std::max<int64_t>(m, 1) * 3);
    for (int64_t i = 0, j = 0; i < n; ++i)
        if (host_kept[i]) { xf3[3 * j] = host_points[2 * i + 1]; xf3[3 * j + 1] = host_points[2 * i]; xf3[3 * j + 2] = 1.0; ++j; }
    matmul(bl, inv3, xf3.data(), tr.data(), 3, 3, m, true);                  // [3][m]
    double bx0 = INFINITY, by0 = INFINITY, bx1 = -INFINITY, by1 = -INFINITY;
    for (int64_t j = 0; j < m; ++j) {
        bx0 = std::min(bx0, tr[j]); bx1 = std::max(bx1, tr[j]);
        by0 = std::min(by0, tr[m + j]); by1 = std::max(by1, tr[m + j]);
    }
    g[8] = bx0; g[9] = by0; g[10] = bx1; g[11] = by1;
    g[12] = mat3[0]; g[13] = mat3[1]; g[14] = mat3[2]; g[15] = theta;
    return 0;
}
