// Device parts of the spectral-line detection on the mean / max image:
// box blur (cv2.blur), row arg-minimum, row mean.  Reference call sites:
// solex_util.py:165-172 (detect_bord), 228-231, 242 (compute_mean_return_fit).
// The images are one frame in size (<= ~1 MB): these kernels are latency-, not
// bandwidth-bound; they exist so that the mean/max images never leave HBM.
#include "shg_common.h"

namespace {

// horizontal window sums with BORDER_REFLECT_101, anchor kw/2
__global__ __launch_bounds__(256) void k_box_rows(const uint16_t* __restrict__ src, int64_t h, int64_t w, int kw,
                                                  uint32_t* __restrict__ tmp) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= h * w) return;
    const int64_t y = i / w, x = i - y * w;
    const uint16_t* row = src + y * w;
    const int64_t xa = x - kw / 2;
    uint32_t s = 0;
    if (xa >= 0 && xa + kw <= w) {
        for (int j = 0; j < kw; ++j) s += row[xa + j];
    } else {
        for (int j = 0; j < kw; ++j) s += row[shg::reflect101(xa + j, w)];
    }
    tmp[i] = s;
}

// vertical window sums, then the OpenCV ColumnSum<int, ushort> scaling: float32 multiply for
// the SIMD lanes (columns below the last multiple of 8), double for the scalar tail; both
// round half to even and saturate.
__global__ __launch_bounds__(256) void k_box_cols(const uint32_t* __restrict__ tmp, int64_t h, int64_t w, int kh,
                                                  double scale, uint16_t* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= h * w) return;
    const int64_t y = i / w, x = i - y * w;
    const int64_t ya = y - kh / 2;
    uint32_t s = 0;
    for (int j = 0; j < kh; ++j) s += tmp[shg::reflect101(ya + j, h) * w + x];
    int r;
    if (x < (w / 8) * 8) r = __float2int_rn(__int2float_rn((int)s) * (float)scale);
    else r = __double2int_rn((double)(int)s * scale);
    dst[i] = (uint16_t)(r < 0 ? 0 : (r > 65535 ? 65535 : r));
}

// one wave per row; first occurrence of the minimum over [x0, x1)
__global__ __launch_bounds__(256) void k_row_argmin(const uint16_t* __restrict__ img, int64_t h, int64_t w, int64_t x0,
                                                    int64_t x1, int32_t* __restrict__ out) {
    const int64_t y = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (y >= h) return;
    const int lane = threadIdx.x & 63;
    const uint16_t* row = img + y * w;
    uint32_t best = 0xffffffffu;           // (value << 16 | index-in-chunk) does not fit: keep key = value, idx separate
    int32_t best_i = 0x7fffffff;
    for (int64_t x = x0 + lane; x < x1; x += 64) {
        const uint32_t v = row[x];
        if (v < best) { best = v; best_i = (int32_t)(x - x0); }
    }
    // (the smallest value, the lowest index among equals: the minimum of value << 32 | index)
    best_i = (int32_t)(uint32_t)shg::wave_min(((uint64_t)best << 32) | (uint32_t)best_i);
    if (lane == 0) out[y] = best_i;
}

// np.mean(axis=1): exact integer row sum, one correctly rounded float64 division
__global__ __launch_bounds__(256) void k_row_mean(const uint16_t* __restrict__ img, int64_t h, int64_t w,
                                                  double* __restrict__ out) {
    const int64_t y = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (y >= h) return;
    const int lane = threadIdx.x & 63;
    const uint16_t* row = img + y * w;
    uint64_t s = 0;
    for (int64_t x = lane; x < w; x += 64) s += row[x];
    s = shg::wave_sum(s);
    if (lane == 0) out[y] = (double)s / (double)w;
}

}  // namespace

extern "C" int shg_box_blur_u16(const uint16_t* src, int64_t h, int64_t w, int kw, int kh, uint16_t* dst, uint32_t* tmp,
                                shg_stream_t stream) {
    SHG_REQUIRE(src && dst && tmp, SHG_E_ARG, "shg_box_blur_u16: null pointer");
    SHG_REQUIRE(h > 0 && w > 0, SHG_E_ARG, "shg_box_blur_u16: empty image");
    // cv2.blur raises for a zero-sized kernel (the reference's hidden precondition at solex_util.py:229-230)
    SHG_REQUIRE(kw > 0 && kh > 0, SHG_E_ARG, "shg_box_blur_u16: kernel %d x %d must be positive", kw, kh);
    SHG_REQUIRE((int64_t)kw * kh <= 32768, SHG_E_UNSUPPORTED, "shg_box_blur_u16: window %d x %d overflows int32 sums", kw, kh);
    hipStream_t st = shg::as_stream(stream);
    const unsigned blocks = (unsigned)((h * w + 255) / 256);
    { SHG_PROF("box_blur", st); k_box_rows<<<blocks, 256, 0, st>>>(src, h, w, kw, tmp); }
    if (int e = shg::check_launch("k_box_rows")) return e;
    { SHG_PROF("box_blur", st); k_box_cols<<<blocks, 256, 0, st>>>(tmp, h, w, kh, 1.0 / ((double)kw * (double)kh), dst); }
    return shg::check_launch("k_box_cols");
}

extern "C" int shg_row_argmin_u16(const uint16_t* img, int64_t h, int64_t w, int64_t x0, int64_t x1, int32_t* out,
                                  shg_stream_t stream) {
    SHG_REQUIRE(img && out, SHG_E_ARG, "shg_row_argmin_u16: null pointer");
    SHG_REQUIRE(h > 0 && w > 0 && x0 >= 0 && x1 <= w && x0 < x1, SHG_E_ARG,
                "shg_row_argmin_u16: empty column range [%lld, %lld) of %lld", (long long)x0, (long long)x1, (long long)w);
    { SHG_PROF("row_argmin", shg::as_stream(stream)); k_row_argmin<<<(unsigned)((h + 3) / 4), 256, 0, shg::as_stream(stream)>>>(img, h, w, x0, x1, out); }
    return shg::check_launch("k_row_argmin");
}

extern "C" int shg_row_mean_u16(const uint16_t* img, int64_t h, int64_t w, double* out, shg_stream_t stream) {
    SHG_REQUIRE(img && out, SHG_E_ARG, "shg_row_mean_u16: null pointer");
    SHG_REQUIRE(h > 0 && w > 0, SHG_E_ARG, "shg_row_mean_u16: empty image");
    { SHG_PROF("row_mean", shg::as_stream(stream)); k_row_mean<<<(unsigned)((h + 3) / 4), 256, 0, shg::as_stream(stream)>>>(img, h, w, out); }
    return shg::check_launch("k_row_mean");
}

// ---- fused forms: the blur never leaves the workgroup ---------------------------------------------------------
// detect_bord needs only the row means of blur(max, 5x5); the line trace only the row arg-minima of blur(mean, 25 x
// kh) (and of the mean image itself).  One workgroup owns FROWS output rows: the horizontal window sums of the rows it
// needs (FROWS + kh - 1, reflected) are staged in LDS, summed down the columns, then along the rows; OpenCV's scaling /
// rounding and the row reduction follow in registers.  Same integer sums and the same float32 / double scaling per column as k_box_rows / k_box_cols,
// so the results are identical; what disappears is two image-sized round trips through HBM and two launches.
namespace {
constexpr int FROWS = 4;

__device__ __forceinline__ int blur_scaled(uint32_t s, int x, int w, double scale) {
    int r;
    if (x < (w & ~7)) r = __float2int_rn(__int2float_rn((int)s) * (float)scale);
    else r = __double2int_rn((double)(int)s * scale);
    return r < 0 ? 0 : (r > 65535 ? 65535 : r);
}

// MODE 0: row means of the blurred image (float64).  MODE 1: first arg-minimum over [x0, x1) of the blurred row
// (relative to x0) and first arg-minimum of the unblurred row over all columns.
// reflect101 without the 64-bit remainder for the indices a window can reach (one reflection at either end)
__device__ __forceinline__ int reflect_near(int i, int n) {
    int r = i < 0 ? -i : i;
    r = r >= n ? 2 * n - 2 - r : r;
    return (unsigned)r < (unsigned)n ? r : (int)shg::reflect101(i, n);
}

struct BlurReduceArgs {
    const uint16_t* src;
    int h, w, kw, kh;
    double scale;
    int x0, x1;
    double* means;
    int32_t *arg_blur, *arg_sharp;
};

template <int MODE> __global__ __launch_bounds__(256) void k_blur_reduce(const BlurReduceArgs kargs) {
    const uint16_t* __restrict__ src = kargs.src;
    const int h = kargs.h, w = kargs.w, kw = kargs.kw, kh = kargs.kh, x0 = kargs.x0, x1 = kargs.x1;
    const double scale = kargs.scale;
    double* __restrict__ means = kargs.means;
    int32_t* __restrict__ arg_blur = kargs.arg_blur;
    int32_t* __restrict__ arg_sharp = kargs.arg_sharp;
    // LDS: vs[FROWS][w] uint32 vertical sums, then raw[FROWS + kh - 1][w] uint16 (the rows this workgroup needs,
    // reflected).  Box sums are integers, so summing down the columns first and along the rows second gives the very
    // sums of k_box_rows + k_box_cols.  All index arithmetic is 32-bit and division-free: with one workgroup per CU the
    // kernel's time is its longest dependent chain, and a 64-bit i / w in each loop was most of it (17-20 us -> see
    // profiles/).
    extern __shared__ uint32_t blur_lds[];
    const int nrows = FROWS + kh - 1;
    uint32_t* vs = blur_lds;                                               // [FROWS][w]
    uint16_t* raw = reinterpret_cast<uint16_t*>(blur_lds + FROWS * w);     // [nrows][w]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = blockIdx.x * FROWS;
    const int ya = r0 - kh / 2;
    if ((w & 1) == 0) {
        // even width: every row starts on a 4-byte boundary, the tile is nrows x w/2 dwords.  Four loads in flight per
        // lane before the first LDS store (a load -> store loop waits for each load in turn: ~10 round trips to memory
        // were most of this kernel's time)
        const int w2 = w >> 1, total = nrows * w2;
        const uint32_t* s2 = reinterpret_cast<const uint32_t*>(src);
        uint32_t* d2 = reinterpret_cast<uint32_t*>(raw);
        for (int base = threadIdx.x; base < total; base += 4 * 256) {
            uint32_t v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = base + 256 * u < total ? base + 256 * u : base;
                const int j = i / w2, x = i - j * w2;
                v[u] = s2[(int64_t)reflect_near(ya + j, h) * w2 + x];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (base + 256 * u < total) d2[base + 256 * u] = v[u];
        }
    } else {
        for (int j = wave; j < nrows; j += 4) {            // one wave per staged row
            const uint16_t* srow = src + (int64_t)reflect_near(ya + j, h) * w;
            uint16_t* drow = raw + j * w;
            for (int x = lane; x < w; x += 64) drow[x] = srow[x];
        }
    }
    __syncthreads();
    for (int x = threadIdx.x; x < w; x += 256) {           // a running sum down each column
        uint32_t s = 0;
        for (int j = 0; j < kh; ++j) s += raw[j * w + x];
        vs[x] = s;
        for (int rr = 1; rr < FROWS; ++rr) {
            s += raw[(rr + kh - 1) * w + x];
            s -= raw[(rr - 1) * w + x];
            vs[rr * w + x] = s;
        }
    }
    __syncthreads();
    for (int rr = wave; rr < FROWS; rr += 4) {             // one wave per output row
        const int y = r0 + rr;
        if (y >= h) break;
        const uint32_t* vrow = vs + rr * w;
        auto blurred = [&](int x) {
            const int xa = x - kw / 2;
            uint32_t s = 0;
            if (xa >= 0 && xa + kw <= w) {
                for (int t = 0; t < kw; ++t) s += vrow[xa + t];
            } else {
                for (int t = 0; t < kw; ++t) s += vrow[reflect_near(xa + t, w)];
            }
            return (uint32_t)blur_scaled(s, x, w, scale);
        };
        if (MODE == 0) {
            uint64_t acc = 0;
            for (int x = lane; x < w; x += 64) acc += (uint64_t)blurred(x);
            acc = shg::wave_sum(acc);
            if (lane == 0) means[y] = (double)acc / (double)w;
        } else {
            uint32_t best = 0xffffffffu, sbest = 0xffffffffu;
            int32_t best_i = 0x7fffffff, sbest_i = 0x7fffffff;
            const uint16_t* row = raw + (rr + kh / 2) * w;                  // the unblurred row y itself
            for (int x = lane; x < w; x += 64) {
                const uint32_t sv = row[x];
                if (sv < sbest) { sbest = sv; sbest_i = x; }
                if (x >= x0 && x < x1) {
                    const uint32_t v = blurred(x);
                    if (v < best) { best = v; best_i = x - x0; }
                }
            }
            // (the smallest value, the lowest index among equals: the minimum of value << 32 | index)
            best_i = (int32_t)(uint32_t)shg::wave_min(((uint64_t)best << 32) | (uint32_t)best_i);
            sbest_i = (int32_t)(uint32_t)shg::wave_min(((uint64_t)sbest << 32) | (uint32_t)sbest_i);
            if (lane == 0) { arg_blur[y] = best_i; arg_sharp[y] = sbest_i; }
        }
    }
}

constexpr size_t kBlurReduceMaxLds = 96 * 1024;
inline size_t blur_reduce_lds(int64_t w, int kh) { return (size_t)FROWS * (size_t)w * sizeof(uint32_t) + ((size_t)(FROWS + kh - 1) * (size_t)w * sizeof(uint16_t) + 3) / 4 * 4; }
}  // namespace

extern "C" int shg_blur_fits_fused(int64_t w, int kh) { return kh > 0 && w > 0 && blur_reduce_lds(w, kh) <= kBlurReduceMaxLds; }

extern "C" int shg_blur_row_mean_u16(const uint16_t* src, int64_t h, int64_t w, int kw, int kh, double* out, shg_stream_t stream) {
    SHG_REQUIRE(src && out, SHG_E_ARG, "shg_blur_row_mean_u16: null pointer");
    SHG_REQUIRE(h > 0 && w > 0, SHG_E_ARG, "shg_blur_row_mean_u16: empty image");
    SHG_REQUIRE(kw > 0 && kh > 0, SHG_E_ARG, "shg_box_blur_u16: kernel %d x %d must be positive", kw, kh);
    SHG_REQUIRE((int64_t)kw * kh <= 32768, SHG_E_UNSUPPORTED, "shg_box_blur_u16: window %d x %d overflows int32 sums", kw, kh);
    SHG_REQUIRE(h < (1ll << 30), SHG_E_UNSUPPORTED, "shg_blur_row_mean_u16: %lld rows", (long long)h);
    SHG_REQUIRE(shg_blur_fits_fused(w, kh), SHG_E_UNSUPPORTED, "shg_blur_row_mean_u16: %lld columns x %d rows do not fit the LDS tile", (long long)w, kh);
    hipStream_t st = shg::as_stream(stream);
    static const bool attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k_blur_reduce<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBlurReduceMaxLds) == hipSuccess;
    (void)attr;
    SHG_PROF("blur_row_mean", st);
    return shg::launch(k_blur_reduce<0>, dim3((unsigned)((h + FROWS - 1) / FROWS)), dim3(256), blur_reduce_lds(w, kh), st,
                        BlurReduceArgs{src, (int)h, (int)w, kw, kh, 1.0 / ((double)kw * (double)kh), 0, (int)w, out, nullptr, nullptr}, "k_blur_reduce");
}

extern "C" int shg_blur_argmin_u16(const uint16_t* src, int64_t h, int64_t w, int kw, int kh, int64_t x0, int64_t x1,
                                   int32_t* out_blur, int32_t* out_sharp, shg_stream_t stream) {
    SHG_REQUIRE(src && out_blur && out_sharp, SHG_E_ARG, "shg_blur_argmin_u16: null pointer");
    SHG_REQUIRE(h > 0 && w > 0, SHG_E_ARG, "shg_blur_argmin_u16: empty image");
    SHG_REQUIRE(kw > 0 && kh > 0, SHG_E_ARG, "shg_box_blur_u16: kernel %d x %d must be positive", kw, kh);
    SHG_REQUIRE((int64_t)kw * kh <= 32768, SHG_E_UNSUPPORTED, "shg_box_blur_u16: window %d x %d overflows int32 sums", kw, kh);
    SHG_REQUIRE(x0 >= 0 && x1 <= w && x0 < x1, SHG_E_ARG, "shg_row_argmin_u16: empty column range [%lld, %lld) of %lld", (long long)x0, (long long)x1, (long long)w);
    SHG_REQUIRE(h < (1ll << 30), SHG_E_UNSUPPORTED, "shg_blur_argmin_u16: %lld rows", (long long)h);
    SHG_REQUIRE(shg_blur_fits_fused(w, kh), SHG_E_UNSUPPORTED, "shg_blur_argmin_u16: %lld columns x %d rows do not fit the LDS tile", (long long)w, kh);
    hipStream_t st = shg::as_stream(stream);
    static const bool attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k_blur_reduce<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBlurReduceMaxLds) == hipSuccess;
    (void)attr;
    SHG_PROF("blur_argmin", st);
    return shg::launch(k_blur_reduce<1>, dim3((unsigned)((h + FROWS - 1) / FROWS)), dim3(256), blur_reduce_lds(w, kh), st,
                        BlurReduceArgs{src, (int)h, (int)w, kw, kh, 1.0 / ((double)kw * (double)kh), (int)x0, (int)x1, nullptr, out_blur, out_sharp}, "k_blur_reduce");
}
