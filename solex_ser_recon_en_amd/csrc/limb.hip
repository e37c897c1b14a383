// Limb detection on the 1/16-size float64 image: the per-pixel half of the reference's
// get_flood_image / get_edge_list (ellipse_to_circle.py:148-250):
//   * cv2.blur on float64 (k x k box filter)                       -> shg_box_blur_f64
//   * skimage.feature.canny up to the hysteresis masks            -> shg_canny_masks_f64
//     (threshold into the 0 / 65000 "flood" image, Gaussian smoothing with mask
//      normalisation, Sobel, hypot, 4-sector interpolated non-maximum suppression,
//      low / high thresholds), scikit-image 0.18.3 + SciPy ndimage semantics.
// The image is ~1 MB: these kernels are launch-latency bound; they exist to take ~15 ms of
// NumPy/SciPy array passes off the serial host tail.  Float64, unfused, in the summation
// order of SciPy's NI_Correlate1D so that the masks equal the host libraries' bit for bit:
//   symmetric kernel  : t = x[0]*w[0]; for j = -R..-1: t += (x[j] + x[-j]) * w[j]
//   antisymmetric     : t = x[0]*w[0]; for j = -R..-1: t += (x[j] - x[-j]) * w[j]
#include <math.h>
#include "shg_common.h"

namespace {

constexpr int MAXR = 16;                     // Gaussian radius: int(4 * sigma + 0.5), sigma <= 4
struct GaussW { double w[2 * MAXR + 1]; int radius; };

// cv2.blur(float64): horizontal sums (left to right), then vertical sums (top to bottom), times 1/(k*k)
__global__ __launch_bounds__(256) void k_boxf_rows(const double* __restrict__ src, int h, int w, int k, double* __restrict__ tmp) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= h * w) return;
    const int y = i / w, x = i - y * w;
    const double* row = src + (int64_t)y * w;
    const int xa = x - k / 2;
    double s = 0.0;
    for (int j = 0; j < k; ++j) s += row[shg::reflect101(xa + j, w)];
    tmp[i] = s;
}

__global__ __launch_bounds__(256) void k_boxf_cols(const double* __restrict__ tmp, int h, int w, int k, double scale,
                                                   double* __restrict__ dst) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= h * w) return;
    const int y = i / w, x = i - y * w;
    const int ya = y - k / 2;
    double s = 0.0;
    for (int j = 0; j < k; ++j) s += tmp[(int64_t)shg::reflect101(ya + j, h) * w + x];
    dst[i] = s * scale;
}

// Gaussian along axis 0 (mode 'constant', cval 0) of (a) the flooded image and (b) an all-ones mask
__global__ __launch_bounds__(256) void k_gauss_v2(const double* __restrict__ blurred, int h, int w, double flood_thresh,
                                                  GaussW g, double* __restrict__ img_v, double* __restrict__ one_v) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= h * w) return;
    const int y = i / w, x = i - y * w;
    auto px = [&](int yy) -> double {          // img_blurred[< thresh3] = 0; [>= thresh3] = 65000 (:226-227)
        if (yy < 0 || yy >= h) return 0.0;
        return blurred[(int64_t)yy * w + x] < flood_thresh ? 0.0 : 65000.0;
    };
    auto one = [&](int yy) -> double { return (yy < 0 || yy >= h) ? 0.0 : 1.0; };
    const int R = g.radius;
    double t = px(y) * g.w[R];
    double u = one(y) * g.w[R];
    for (int j = -R; j < 0; ++j) {
        t += (px(y + j) + px(y - j)) * g.w[R + j];
        u += (one(y + j) + one(y - j)) * g.w[R + j];
    }
    img_v[i] = t;
    one_v[i] = u;
}

// Gaussian along axis 1 of both planes, then smoothed = image / (bleed_over + eps)
__global__ __launch_bounds__(256) void k_gauss_h2_div(const double* __restrict__ img_v, const double* __restrict__ one_v, int h,
                                                      int w, GaussW g, double* __restrict__ smoothed) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= h * w) return;
    const int y = i / w, x = i - y * w;
    const double* a = img_v + (int64_t)y * w;
    const double* b = one_v + (int64_t)y * w;
    auto at = [&](const double* p, int xx) -> double { return (xx < 0 || xx >= w) ? 0.0 : p[xx]; };
    const int R = g.radius;
    double t = a[x] * g.w[R];
    double u = b[x] * g.w[R];
    for (int j = -R; j < 0; ++j) {
        t += (at(a, x + j) + at(a, x - j)) * g.w[R + j];
        u += (at(b, x + j) + at(b, x - j)) * g.w[R + j];
    }
    smoothed[i] = t / (u + 2.220446049250313e-16);
}

// glibc 2.35 hypot (sysdeps/ieee754/dbl-64/e_hypot.c, the non-FMA kernel), for finite normal-range inputs:
// this is what np.hypot evaluates on the reference's x86-64 hosts.
__device__ __forceinline__ double hypot_glibc(double x, double y) {
    x = fabs(x);
    y = fabs(y);
    const double ax = x < y ? y : x;
    const double ay = x < y ? x : y;
    if (ax >= ay / 0x1p-54) return ax + ay;
    double h = sqrt(ax * ax + ay * ay);
    double t1, t2;
    if (h <= 2.0 * ay) {
        const double delta = h - ay;
        t1 = ax * (2.0 * delta - ax);
        t2 = (delta - 2.0 * (ax - ay)) * delta;
    } else {
        const double delta = h - ax;
        t1 = 2.0 * delta * (ax - 2.0 * ay);
        t2 = (4.0 * delta - ay) * ay + delta * delta;
    }
    h -= (t1 + t2) / (2.0 * h);
    return h;
}

__device__ __forceinline__ int refl(int i, int n) {       // scipy mode 'reflect': d c b a | a b c d | d c b a
    return i < 0 ? -i - 1 : (i >= n ? 2 * n - 1 - i : i);
}

// ndi.sobel(axis=0) and (axis=1): derivative [-1,0,1] along the axis, then [1,2,1] along the other
__global__ __launch_bounds__(256) void k_sobel_mag(const double* __restrict__ s, int h, int w, double* __restrict__ isob,
                                                   double* __restrict__ jsob, double* __restrict__ mag) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= h * w) return;
    const int y = i / w, x = i - y * w;
    const int ym = refl(y - 1, h), yp = refl(y + 1, h), xm = refl(x - 1, w), xp = refl(x + 1, w);
    auto S = [&](int yy, int xx) -> double { return s[(int64_t)yy * w + xx]; };
    // correlate1d([-1,0,1]): t = x[0]*0 + (x[-1] - x[+1]) * (-1)
    auto dy = [&](int xx) -> double { double t = S(y, xx) * 0.0; t += (S(ym, xx) - S(yp, xx)) * -1.0; return t; };
    auto dx = [&](int yy) -> double { double t = S(yy, x) * 0.0; t += (S(yy, xm) - S(yy, xp)) * -1.0; return t; };
    // correlate1d([1,2,1]): t = x[0]*2 + (x[-1] + x[+1]) * 1
    double iv = dy(x) * 2.0;
    iv += (dy(xm) + dy(xp)) * 1.0;
    double jv = dx(y) * 2.0;
    jv += (dx(ym) + dx(yp)) * 1.0;
    isob[i] = iv;
    jsob[i] = jv;
    mag[i] = hypot_glibc(iv, jv);
}

// 4-sector non-maximum suppression with interpolation, then the two thresholds
__global__ __launch_bounds__(256) void k_nms(const double* __restrict__ isob, const double* __restrict__ jsob,
                                             const double* __restrict__ mag, int h, int w, double low, double high,
                                             uint8_t* __restrict__ low_mask, uint8_t* __restrict__ high_mask) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= h * w) return;
    const int y = i / w, x = i - y * w;
    bool local = false;
    const double m = mag[i];
    // eroded all-ones mask (border_value 0) & magnitude > 0
    if (y > 0 && y < h - 1 && x > 0 && x < w - 1 && m > 0.0) {
        const double is = isob[i], js = jsob[i];
        const double ai = fabs(is), aj = fabs(js);
        auto M = [&](int dy, int dx) -> double { return mag[(int64_t)(y + dy) * w + (x + dx)]; };
        const bool same = (is >= 0 && js >= 0) || (is <= 0 && js <= 0);
        const bool opp = (is <= 0 && js >= 0) || (is >= 0 && js <= 0);
        auto test = [&](double wgt, double p1, double p2, double m1, double m2) -> bool {
            const bool c_plus = p2 * wgt + p1 * (1 - wgt) <= m;
            const bool c_minus = m2 * wgt + m1 * (1 - wgt) <= m;
            return c_plus && c_minus;
        };
        // later sectors overwrite earlier ones, as the sequential assignments in skimage do
        if (same && ai >= aj) local = test(aj / ai, M(1, 0), M(1, 1), M(-1, 0), M(-1, -1));
        if (same && ai <= aj) local = test(ai / aj, M(0, 1), M(1, 1), M(0, -1), M(-1, -1));
        if (opp && ai <= aj) local = test(ai / aj, M(0, 1), M(-1, 1), M(0, -1), M(1, -1));
        if (opp && ai >= aj) local = test(aj / ai, M(-1, 0), M(-1, 1), M(1, 0), M(1, -1));
    }
    low_mask[i] = (local && m >= low) ? 1 : 0;
    high_mask[i] = (local && m >= high) ? 1 : 0;
}

}  // namespace

extern "C" int shg_box_blur_f64(const double* src, int64_t h, int64_t w, int k, double* dst, double* tmp, shg_stream_t stream) {
    SHG_REQUIRE(src && dst && tmp, SHG_E_ARG, "shg_box_blur_f64: null pointer");
    SHG_REQUIRE(h > 0 && w > 0 && h * w < (1ll << 30), SHG_E_ARG, "shg_box_blur_f64: bad image size");
    SHG_REQUIRE(k > 0, SHG_E_ARG, "shg_box_blur_f64: kernel %d must be positive", k);      // cv2.blur raises as well
    hipStream_t st = shg::as_stream(stream);
    const unsigned blocks = (unsigned)((h * w + 255) / 256);
    { SHG_PROF("box_blur_f64", st); k_boxf_rows<<<blocks, 256, 0, st>>>(src, (int)h, (int)w, k, tmp); }
    if (int e = shg::check_launch("k_boxf_rows")) return e;
    { SHG_PROF("box_blur_f64", st); k_boxf_cols<<<blocks, 256, 0, st>>>(tmp, (int)h, (int)w, k, 1.0 / ((double)k * (double)k), dst); }
    return shg::check_launch("k_boxf_cols");
}

extern "C" size_t shg_canny_workspace_bytes(int64_t h, int64_t w) {
    if (h <= 0 || w <= 0) return 0;
    return (size_t)6 * (size_t)h * (size_t)w * sizeof(double);
}

extern "C" int shg_canny_masks_f64(const double* blurred, int64_t h, int64_t w, double flood_thresh, const double* host_gauss_weights,
                                   int radius, double low, double high, uint8_t* low_mask, uint8_t* high_mask,
                                   void* workspace, size_t workspace_bytes, shg_stream_t stream) {
    SHG_REQUIRE(blurred && host_gauss_weights && low_mask && high_mask && workspace, SHG_E_ARG, "shg_canny_masks_f64: null pointer");
    SHG_REQUIRE(h > 2 && w > 2 && h * w < (1ll << 30), SHG_E_ARG, "shg_canny_masks_f64: bad image size");
    SHG_REQUIRE(radius >= 0 && radius <= MAXR, SHG_E_UNSUPPORTED, "shg_canny_masks_f64: Gaussian radius %d > %d", radius, MAXR);
    SHG_REQUIRE(workspace_bytes >= shg_canny_workspace_bytes(h, w), SHG_E_WORKSPACE, "shg_canny_masks_f64: workspace too small");
    GaussW g;
    g.radius = radius;
    for (int i = 0; i < 2 * radius + 1; ++i) g.w[i] = host_gauss_weights[i];
    const size_t n = (size_t)h * (size_t)w;
    double* p = static_cast<double*>(workspace);
    double *img_v = p, *one_v = p + n, *smoothed = p + 2 * n, *isob = p + 3 * n, *jsob = p + 4 * n, *mag = p + 5 * n;
    hipStream_t st = shg::as_stream(stream);
    const unsigned blocks = (unsigned)((n + 255) / 256);
    { SHG_PROF("canny", st); k_gauss_v2<<<blocks, 256, 0, st>>>(blurred, (int)h, (int)w, flood_thresh, g, img_v, one_v); }
    if (int e = shg::check_launch("k_gauss_v2")) return e;
    { SHG_PROF("canny", st); k_gauss_h2_div<<<blocks, 256, 0, st>>>(img_v, one_v, (int)h, (int)w, g, smoothed); }
    if (int e = shg::check_launch("k_gauss_h2_div")) return e;
    { SHG_PROF("canny", st); k_sobel_mag<<<blocks, 256, 0, st>>>(smoothed, (int)h, (int)w, isob, jsob, mag); }
    if (int e = shg::check_launch("k_sobel_mag")) return e;
    { SHG_PROF("canny", st); k_nms<<<blocks, 256, 0, st>>>(isob, jsob, mag, (int)h, (int)w, low, high, low_mask, high_mask); }
    return shg::check_launch("k_nms");
}
