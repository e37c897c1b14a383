// Transversalium (row-defect) correction: device parts.
// Reference: correct_transversalium2 (solex_util.py:383-395, 489, 515-516) and
// reject_outliers (solex_util.py:76-86).
//
// k_rowpair_stats: one workgroup per row pair.  The log-ratios of the chord are held
// in LDS, sorted with a bitonic network to read the median, their absolute deviations
// are sorted the same way for the MAD, and the 2-MAD inliers are averaged.  All float64.
// The reference sums the inliers in image order with NumPy's pairwise scheme; here they
// are summed by a fixed-shape tree, so the mean can differ in the last bits (as NumPy's
// own log already does between CPUs); the correction factors agree to ~1e-15 relative.
//
// k_scale_rows: img * c[y], saturate, truncate -- a pure streaming pass.
#include <math.h>
#include "shg_common.h"

namespace {

constexpr int MAXN = SHG_TRANSV_MAX_COLS;   // 8192 doubles = 64 KiB per array
constexpr int NT = 256;

__device__ __forceinline__ void bitonic_sort(double* a, int n2) {
    // n2 = power of two >= element count; padding holds +inf
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n2; i += NT) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const double x = a[i], y = a[ixj];
                    const bool up = (i & k) == 0;
                    // NaN never compares: rows holding a NaN are detected before sorting
                    if ((x > y) == up) { a[i] = y; a[ixj] = x; }
                }
            }
            __syncthreads();
        }
    }
}

__device__ __forceinline__ double median_sorted(const double* a, int n) {
    // np.median: mean of the two middle order statistics for even n
    if (n & 1) return a[n >> 1];
    return (a[(n >> 1) - 1] + a[n >> 1]) / 2.0;
}

__device__ __forceinline__ double block_sum(double v, double* red) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(NT) void k_rowpair_stats(const uint16_t* __restrict__ img, int64_t pitch, int64_t y1,
                                                      const int32_t* __restrict__ xa, const int32_t* __restrict__ xb,
                                                      double* __restrict__ out) {
    extern __shared__ double lds[];      // [2][n2]
    __shared__ double red[4];
    __shared__ int bad;
    const int t = blockIdx.x + 1;        // out[0] stays 0 (solex_util.py:386)
    const int64_t y = y1 + t;
    const int a = xa[t], b = xb[t];
    const int n = b - a;
    if (n <= 0) {                        // np.mean of an empty slice
        if (threadIdx.x == 0) out[t] = __builtin_nan("");
        return;
    }
    int n2 = 1;
    while (n2 < n) n2 <<= 1;
    double* v = lds;
    double* d = lds + n2;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    const uint16_t* r1 = img + y * pitch + a;
    const uint16_t* r0 = img + (y - 1) * pitch + a;
    // Zero pixels give -inf / +inf / NaN ratios.  NumPy keeps infinities as ordinary (sortable) values and
    // lets any NaN poison the row statistic (np.median -> nan -> empty inlier set -> nan); same here.
    for (int i = threadIdx.x; i < n2; i += NT) {
        double x = __builtin_inf();
        if (i < n) {
            x = log((double)r1[i] / (double)r0[i]);        // np.log(strip1 / strip0)
            if (x != x) bad = 1;
        }
        v[i] = x;
    }
    __syncthreads();
    if (bad) {
        if (threadIdx.x == 0) out[t] = __builtin_nan("");
        return;
    }
    bitonic_sort(v, n2);
    const double med = median_sorted(v, n);
    for (int i = threadIdx.x; i < n2; i += NT) {
        double dv = __builtin_inf();
        if (i < n) {
            dv = fabs(v[i] - med);                         // inf - inf or a NaN median -> NaN
            if (dv != dv) bad = 1;
        }
        d[i] = dv;
    }
    __syncthreads();
    if (bad) {
        if (threadIdx.x == 0) out[t] = __builtin_nan("");
        return;
    }
    bitonic_sort(d, n2);
    const double mdev = median_sorted(d, n);
    double s = 0.0, cnt = 0.0;
    for (int i = threadIdx.x; i < n; i += NT) {
        const double x = v[i];
        const double dev = fabs(x - med);
        const bool keep = (mdev != 0.0) ? (dev / mdev < 2.0) : true;   // s = d/mdev if mdev else zeros; data[s < m] (NaN < 2 is false)
        if (keep) { s += x; cnt += 1.0; }
    }
    s = block_sum(s, red);
    cnt = block_sum(cnt, red);
    if (threadIdx.x == 0) out[t] = s / cnt;
}

__global__ __launch_bounds__(256) void k_scale_rows(const uint16_t* __restrict__ img, int64_t w, int64_t pitch,
                                                    const double* __restrict__ c, uint16_t* __restrict__ dst,
                                                    int64_t dst_pitch) {
    const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t y = blockIdx.y;
    if (x >= w) return;
    double v = (double)img[y * pitch + x] * c[y];
    v = v > 65535.0 ? 65535.0 : v;
    dst[y * dst_pitch + x] = (uint16_t)(int)v;
}

}  // namespace

extern "C" int shg_rowpair_logratio_stats(const uint16_t* img, int64_t h, int64_t w, int64_t pitch, int64_t y1, int64_t y2,
                                          const int32_t* xa, const int32_t* xb, double* out, shg_stream_t stream) {
    SHG_REQUIRE(img && xa && xb && out, SHG_E_ARG, "shg_rowpair_logratio_stats: null pointer");
    SHG_REQUIRE(h > 0 && w > 0 && pitch >= w, SHG_E_ARG, "shg_rowpair_logratio_stats: bad image size");
    SHG_REQUIRE(y1 >= 0 && y2 <= h && y2 > y1, SHG_E_ARG, "shg_rowpair_logratio_stats: rows [%lld, %lld) outside the image",
                (long long)y1, (long long)y2);
    SHG_REQUIRE(w <= MAXN, SHG_E_UNSUPPORTED, "shg_rowpair_logratio_stats: width %lld > %d", (long long)w, MAXN);
    hipStream_t st = shg::as_stream(stream);
    if (hipError_t e = hipMemsetAsync(out, 0, sizeof(double), st)) {
        shg::set_error("shg_rowpair_logratio_stats: memset: %s", hipGetErrorString(e));
        return (int)e;
    }
    const int64_t rows = y2 - y1 - 1;
    if (rows <= 0) return 0;
    int n2 = 1;
    while (n2 < w) n2 <<= 1;
    const size_t lds_bytes = (size_t)2 * n2 * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_rowpair_stats), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * MAXN * 8);
        attr_set = true;
    }
    { SHG_PROF("rowpair_stats", st); k_rowpair_stats<<<(unsigned)rows, NT, lds_bytes, st>>>(img, pitch, y1, xa, xb, out); }
    return shg::check_launch("k_rowpair_stats");
}

extern "C" int shg_scale_rows_u16(const uint16_t* img, int64_t h, int64_t w, int64_t pitch, const double* c, uint16_t* dst,
                                  int64_t dst_pitch, shg_stream_t stream) {
    SHG_REQUIRE(img && c && dst, SHG_E_ARG, "shg_scale_rows_u16: null pointer");
    SHG_REQUIRE(h > 0 && w > 0 && pitch >= w && dst_pitch >= w, SHG_E_ARG, "shg_scale_rows_u16: bad image size");
    SHG_REQUIRE(h < 65536, SHG_E_UNSUPPORTED, "shg_scale_rows_u16: more than 65535 rows");
    dim3 grid((unsigned)((w + 255) / 256), (unsigned)h);
    { SHG_PROF("scale_rows", shg::as_stream(stream)); k_scale_rows<<<grid, 256, 0, shg::as_stream(stream)>>>(img, w, pitch, c, dst, dst_pitch); }
    return shg::check_launch("k_scale_rows");
}
