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
// last bits may differ).  The limb geometry (ellipse fit, 2x2 algebra) has no bit-exact
// counterpart in the reference's stack to begin with (lsq-ellipse is unpinned); it is computed in
// extended precision and agrees with the NumPy restatement to ~1e-12.
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
    SHG_REQUIRE(host_row_means && lb && ub && n > 0, SHG_E_ARG, "shg_host_detect_bord: bad argument");
    std::vector<double> s(host_row_means, host_row_means + n);
    std::sort(s.begin(), s.end());
    const double med = (n & 1) ? s[n / 2] : (s[n / 2 - 1] + s[n / 2]) / 2.0;      // np.median
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
    std::vector<double> sorted(rounded);
    std::sort(sorted.begin(), sorted.end());
    std::vector<double> values;
    std::vector<int64_t> counts;
    for (size_t i = 0; i < sorted.size(); ++i) {
        if (i == 0 || sorted[i] != sorted[i - 1]) { values.push_back(sorted[i]); counts.push_back(1); }   // NaNs stay apart, as in NumPy < 1.21; equal_nan groups them since
        else ++counts.back();
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

// ---- a8: LsqEllipse (Halir & Flusser's numerically stable direct least squares fit) -------------------
// points[n][2] (first, second coordinate as given).  out: center[2], width, height, phi -- lsq-ellipse 2.0's
// as_parameters(), ellipse_to_circle.py:57-59.  Extended-precision normal equations: the scatter matrices of pixel
// coordinates reach 1e18.
namespace {
typedef long double ld;

static bool inv3(const ld a[3][3], ld out[3][3]) {
    const ld c00 = a[1][1] * a[2][2] - a[1][2] * a[2][1], c01 = a[1][2] * a[2][0] - a[1][0] * a[2][2],
             c02 = a[1][0] * a[2][1] - a[1][1] * a[2][0];
    const ld det = a[0][0] * c00 + a[0][1] * c01 + a[0][2] * c02;
    if (det == 0) return false;
    out[0][0] = c00 / det; out[0][1] = (a[0][2] * a[2][1] - a[0][1] * a[2][2]) / det; out[0][2] = (a[0][1] * a[1][2] - a[0][2] * a[1][1]) / det;
    out[1][0] = c01 / det; out[1][1] = (a[0][0] * a[2][2] - a[0][2] * a[2][0]) / det; out[1][2] = (a[0][2] * a[1][0] - a[0][0] * a[1][2]) / det;
    out[2][0] = c02 / det; out[2][1] = (a[0][1] * a[2][0] - a[0][0] * a[2][1]) / det; out[2][2] = (a[0][0] * a[1][1] - a[0][1] * a[1][0]) / det;
    return true;
}

static int fit_ellipse(const double* pts, int64_t n, double center[2], double* width, double* height, double* phi) {
    if (n < 5) { shg::set_error("ellipse fit: %lld limb points (at least 5 needed)", (long long)n); return SHG_E_RUNTIME; }
    // centre the coordinates for the accumulation (the algebra below is translation covariant: the conic is moved back)
    ld mx = 0, my = 0;
    for (int64_t i = 0; i < n; ++i) { mx += pts[2 * i]; my += pts[2 * i + 1]; }
    mx /= n; my /= n;
    ld S1[3][3] = {{0}}, S2[3][3] = {{0}}, S3[3][3] = {{0}};
    for (int64_t i = 0; i < n; ++i) {
        const ld x = pts[2 * i] - mx, y = pts[2 * i + 1] - my;
        const ld d1[3] = {x * x, x * y, y * y}, d2[3] = {x, y, 1};
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) { S1[a][b] += d1[a] * d1[b]; S2[a][b] += d1[a] * d2[b]; S3[a][b] += d2[a] * d2[b]; }
    }
    ld S3i[3][3];
    if (!inv3(S3, S3i)) { shg::set_error("Singular matrix"); return SHG_E_LINALG; }
    // T = -S3^-1 S2^T ;  M = C1^-1 (S1 + S2 T)
    ld T[3][3], R[3][3], M[3][3];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) { ld t = 0; for (int c = 0; c < 3; ++c) t += S3i[a][c] * S2[b][c]; T[a][b] = -t; }
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) { ld t = S1[a][b]; for (int c = 0; c < 3; ++c) t += S2[a][c] * T[c][b]; R[a][b] = t; }
    for (int b = 0; b < 3; ++b) { M[0][b] = R[2][b] / 2; M[1][b] = -R[1][b]; M[2][b] = R[0][b] / 2; }
    // eigenvalues of M: roots of l^3 - tr l^2 + c1 l - det
    const ld tr = M[0][0] + M[1][1] + M[2][2];
    const ld c1 = M[0][0] * M[1][1] - M[0][1] * M[1][0] + M[0][0] * M[2][2] - M[0][2] * M[2][0] + M[1][1] * M[2][2] - M[1][2] * M[2][1];
    const ld det = M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
                   M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
    // depressed cubic t^3 + p t + q, l = t + tr/3
    const ld sh = tr / 3;
    const ld p = c1 - tr * tr / 3, q = -2 * tr * tr * tr / 27 + tr * c1 / 3 - det;
    ld roots[3];
    int n_roots = 0;
    const ld disc = q * q / 4 + p * p * p / 27;
    if (disc > 0) {
        const ld sq = sqrtl(disc);
        roots[n_roots++] = cbrtl(-q / 2 + sq) + cbrtl(-q / 2 - sq) + sh;
    } else if (p == 0) {
        roots[n_roots++] = sh;
    } else {
        const ld rr = 2 * sqrtl(-p / 3);
        ld arg = 3 * q / (p * rr);
        arg = arg > 1 ? 1 : (arg < -1 ? -1 : arg);
        const ld th = acosl(arg) / 3;
        for (int k2 = 0; k2 < 3; ++k2) roots[n_roots++] = rr * cosl(th - 2 * (ld)M_PIl * k2 / 3) + sh;
    }
    auto charpoly = [&](ld l) { return ((l - tr) * l + c1) * l - det; };
    auto dchar = [&](ld l) { return (3 * l - 2 * tr) * l + c1; };
    ld best_a1[3] = {0, 0, 0};
    bool found = false;
    for (int r = 0; r < n_roots && !found; ++r) {
        ld l = roots[r];
        for (int it = 0; it < 4; ++it) { const ld d = dchar(l); if (d == 0) break; l -= charpoly(l) / d; }    // polish
        ld A[3][3];
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) A[a][b] = M[a][b] - (a == b ? l : 0);
        // null vector: the largest cross product of two rows
        ld bestn = -1, v[3] = {0, 0, 0};
        for (int a = 0; a < 3; ++a)
            for (int b = a + 1; b < 3; ++b) {
                const ld cx = A[a][1] * A[b][2] - A[a][2] * A[b][1], cy = A[a][2] * A[b][0] - A[a][0] * A[b][2],
                         cz = A[a][0] * A[b][1] - A[a][1] * A[b][0];
                const ld nn = cx * cx + cy * cy + cz * cz;
                if (nn > bestn) { bestn = nn; v[0] = cx; v[1] = cy; v[2] = cz; }
            }
        if (!(bestn > 0)) continue;
        const ld nv = sqrtl(bestn);
        for (int a = 0; a < 3; ++a) v[a] /= nv;
        if (4 * v[0] * v[2] - v[1] * v[1] > 0) { found = true; for (int a = 0; a < 3; ++a) best_a1[a] = v[a]; }
    }
    if (!found) { shg::set_error("ellipse fit: no elliptical solution (the limb points do not describe an ellipse)"); return SHG_E_RUNTIME; }
    ld a2[3];
    for (int a = 0; a < 3; ++a) { ld t = 0; for (int c = 0; c < 3; ++c) t += T[a][c] * best_a1[c]; a2[a] = t; }
    // conic in centred coordinates: A x^2 + B xy + C y^2 + D x + E y + F; move back by (mx, my)
    const ld A_ = best_a1[0], B_ = best_a1[1], C_ = best_a1[2];
    const ld D_ = a2[0] - 2 * A_ * mx - B_ * my, E_ = a2[1] - 2 * C_ * my - B_ * mx;
    const ld F_ = a2[2] + A_ * mx * mx + B_ * mx * my + C_ * my * my - a2[0] * mx - a2[1] * my;
    const ld a = A_, b = B_ / 2, c = C_, d = D_ / 2, f = E_ / 2, g = F_;
    const ld den = b * b - a * c;
    const ld x0 = (c * d - b * f) / den, y0 = (a * f - b * d) / den;
    const ld numerator = 2 * (a * f * f + c * d * d + g * b * b - 2 * b * d * f - a * c * g);
    const ld root = sqrtl(1 + 4 * b * b / ((a - c) * (a - c)));
    center[0] = (double)x0;
    center[1] = (double)y0;
    *width = (double)sqrtl(numerator / (den * ((c - a) * root - (c + a))));
    *height = (double)sqrtl(numerator / (den * ((a - c) * root - (c + a))));
    *phi = (double)(0.5L * atanl((2 * b) / (a - c)));
    return 0;
}

static void rot2(double x, double m[2][2]) { m[0][0] = cos(x); m[0][1] = sin(x); m[1][0] = -sin(x); m[1][1] = cos(x); }
static void mul2(const double a[2][2], const double b[2][2], double o[2][2]) {
    double t[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) t[i][j] = a[i][0] * b[0][j] + a[i][1] * b[1][j];
    memcpy(o, t, sizeof(t));
}
static bool inv2(const double a[2][2], double o[2][2]) {
    const double det = a[0][0] * a[1][1] - a[0][1] * a[1][0];
    if (det == 0) return false;
    const double t[2][2] = {{a[1][1] / det, -a[0][1] / det}, {-a[1][0] / det, a[0][0] / det}};
    memcpy(o, t, sizeof(t));
    return true;
}
}  // namespace

extern "C" int shg_host_fit_ellipse(const double* host_points, int64_t n, double* host_center2, double* width,
                                    double* height, double* phi) {
    SHG_REQUIRE(host_points && host_center2 && width && height && phi, SHG_E_ARG, "shg_host_fit_ellipse: null pointer");
    return fit_ellipse(host_points, n, host_center2, width, height, phi);
}

// get_correction_matrix(phi, r) (ellipse_to_circle.py:39-50): inverse correction matrix (row major 2x2), theta
extern "C" int shg_host_correction_matrix(double phi, double r, double* host_inv4, double* theta_out) {
    SHG_REQUIRE(host_inv4 && theta_out, SHG_E_ARG, "shg_host_correction_matrix: null pointer");
    double rp[2][2], rm[2][2], st[2][2], dg[2][2] = {{r, 0}, {0, 1}}, rt[2][2], cm[2][2], inv[2][2];
    rot2(phi, rp);
    rot2(-phi, rm);
    mul2(rp, dg, st);
    mul2(st, rm, st);
    const double theta = atan(st[1][0] / st[0][0]);
    rot2(theta, rt);
    mul2(rt, st, cm);
    cm[1][0] = 0;
    const double d = cm[1][1];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) cm[i][j] /= d;
    if (!inv2(cm, inv)) { shg::set_error("Singular matrix"); return SHG_E_LINALG; }
    host_inv4[0] = inv[0][0]; host_inv4[1] = inv[0][1]; host_inv4[2] = inv[1][0]; host_inv4[3] = inv[1][1];
    *theta_out = theta;
    return 0;
}

// two_step (ellipse_to_circle.py:62-91): fit, drop the points inside the ellipse by more than the largest outward
// residual, refit, bring phi within pi/4 of zero.  points[n][2] = (row, col).  kept[n] (may be NULL): 1 for the points
// of the second fit.  out: center[2] (row, col), height, phi, ratio, outline[100][2] (may be NULL).
extern "C" int shg_host_two_step(const double* host_points, int64_t n, double* host_center2, double* height_out,
                                 double* phi_out, double* ratio_out, uint8_t* host_kept, int64_t* n_kept,
                                 double* host_outline200) {
    SHG_REQUIRE(host_points && host_center2 && height_out && phi_out && ratio_out && n_kept, SHG_E_ARG, "shg_host_two_step: null pointer");
    double center[2], width, height, phi;
    if (int e = fit_ellipse(host_points, n, center, &width, &height, &phi)) return e;
    double mat[4], theta;
    if (int e = shg_host_correction_matrix(phi, height / width, mat, &theta)) return e;
    std::vector<double> values((size_t)n);
    double vmax = -INFINITY;
    for (int64_t i = 0; i < n; ++i) {
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
    if (int e = fit_ellipse(kept.data(), *n_kept, center, &width, &height, &phi)) return e;
    if (host_outline200)
        for (int i = 0; i < 100; ++i) {                                   // reg.return_fit(n_points=100)
            const double t = (double)i * (2 * M_PI / 99.0);
            const double tt = i == 99 ? 2 * M_PI : t;
            host_outline200[2 * i] = center[0] + width * cos(tt) * cos(phi) - height * sin(tt) * sin(phi);
            host_outline200[2 * i + 1] = center[1] + width * cos(tt) * sin(phi) + height * sin(tt) * cos(phi);
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

// correct_image's geometry (ellipse_to_circle.py:100-122): everything derived from (phi, ratio) and the image shape.
// out: mat3[9] (row major), inv_mat[4], origin[2], det, theta, out_h, out_w.
extern "C" int shg_host_warp_geometry(double phi, double ratio, int64_t h, int64_t w, double* host_mat3_9,
                                      double* host_inv4, double* host_origin2, double* det_out, double* theta_out,
                                      int64_t* out_h, int64_t* out_w) {
    SHG_REQUIRE(host_mat3_9 && host_inv4 && host_origin2 && det_out && theta_out && out_h && out_w, SHG_E_ARG,
                "shg_host_warp_geometry: null pointer");
    double m4[4];
    if (int e = shg_host_correction_matrix(phi, ratio, m4, theta_out)) return e;
    const double mat[2][2] = {{m4[0], m4[1]}, {m4[2], m4[3]}};
    double inv[2][2];
    if (!inv2(mat, inv)) { shg::set_error("Singular matrix"); return SHG_E_LINALG; }
    const double corners[4][2] = {{0, 0}, {0, (double)h}, {(double)w, 0}, {(double)w, (double)h}};
    double xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
    for (auto& c : corners) {
        const double x = inv[0][0] * c[0] + inv[0][1] * c[1], y = inv[1][0] * c[0] + inv[1][1] * c[1];
        xmin = std::min(xmin, x); xmax = std::max(xmax, x); ymin = std::min(ymin, y); ymax = std::max(ymax, y);
    }
    const double new_h = ymax - ymin, new_w = xmax - xmin;
    // mat3 = [[mat, 0], [0, 1]] @ translate(origin)
    double m3[9] = {mat[0][0], mat[0][1], mat[0][0] * xmin + mat[0][1] * ymin,
                    mat[1][0], mat[1][1], mat[1][0] * xmin + mat[1][1] * ymin, 0, 0, 1};
    if (!(m3[3] == 0 && m3[4] == 1 && m3[5] == 0)) {
        shg::set_error("correct_image: the correction never moves rows (ellipse_to_circle.py:48-49); got row [%g %g %g]", m3[3], m3[4], m3[5]);
        return SHG_E_RUNTIME;
    }
    memcpy(host_mat3_9, m3, sizeof(m3));
    host_inv4[0] = inv[0][0]; host_inv4[1] = inv[0][1]; host_inv4[2] = inv[1][0]; host_inv4[3] = inv[1][1];
    host_origin2[0] = xmin; host_origin2[1] = ymin;
    *det_out = mat[0][0] * mat[1][1] - mat[0][1] * mat[1][0];
    *out_h = (int64_t)ceil(new_h);
    *out_w = (int64_t)ceil(new_w);
    return 0;
}

// ---- a9: chord bounds of the transversalium rows (solex_util.py:384-391) ---------------------------------
// xa, xb [max(y2-y1,1)] (entry 0 unused): NumPy-normalised slice [a, b) of row y1+i.
extern "C" int shg_host_chord_bounds(double cx, double cy, double r, double b0, double b2, int64_t y1, int64_t y2,
                                     int64_t w, int32_t* host_xa, int32_t* host_xb) {
    SHG_REQUIRE(host_xa && host_xb && w > 0, SHG_E_ARG, "shg_host_chord_bounds: bad argument");
    const int64_t count = std::max<int64_t>(y2 - y1, 1);
    for (int64_t i = 0; i < count; ++i) host_xa[i] = host_xb[i] = 0;
    for (int64_t y = y1 + 1; y < y2; ++y) {
        const double v = r * r - ((double)y - cy) * ((double)y - cy);
        if (v < 0) { shg::set_error("transversalium: row outside the disk circle (complex chord length)"); return SHG_E_TYPE; }
        const double dx = floor(pow(v, 0.5));                         // math.floor((r**2 - (y-cy)**2) ** 0.5)
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
