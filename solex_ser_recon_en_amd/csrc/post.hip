// Small per-disk image passes: crop/pad, brightness rescale, protuberance disc,
// 4x4 block mean.  Reference: Solex_recon.py:155-171 (crop), solex_util.py:519-525
// (rescale_brightness), solex_util.py:542-547 (cv2.circle), ellipse_to_circle.py:299-302
// (downscale_local_mean).  All are single streaming passes over a few-MB image.
#include "shg_common.h"

namespace {

__global__ __launch_bounds__(256) void k_crop_pad(const uint16_t* __restrict__ src, int64_t pitch, uint16_t* __restrict__ dst,
                                                  int64_t nw, int64_t dst_pitch, int64_t sx0, int64_t dx0, int64_t n,
                                                  uint16_t fill, int fill_is_src00) {
    const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t y = blockIdx.y;
    if (x >= nw) return;
    if (fill_is_src00) fill = src[0];                   // np.full(..., img[0, 0]) without a host round trip
    const int64_t j = x - dx0;
    dst[y * dst_pitch + x] = (j >= 0 && j < n) ? src[y * pitch + sx0 + j] : fill;
}

__global__ __launch_bounds__(256) void k_rescale(const uint16_t* __restrict__ img, int64_t w, int64_t pitch, double a,
                                                 double lo, double span, uint16_t* __restrict__ dst, int64_t dst_pitch) {
    const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t y = blockIdx.y;
    if (x >= w) return;
    // (float(sat) * alpha * (img - lo)) / (hi - lo): left to right, float64
    double v = a * ((double)img[y * pitch + x] - lo) / span;
    v = v < 0.0 ? 0.0 : v;
    v = v > 65535.0 ? 65535.0 : v;
    dst[y * dst_pitch + x] = (uint16_t)(int)v;
}

// the same for 8-bit images (clahe_apply.py:251 stretches 8-bit PNGs too): sat = 255
__global__ __launch_bounds__(256) void k_rescale_u8(const uint8_t* __restrict__ img, int64_t w, int64_t pitch, double a, double lo,
                                                    double span, uint8_t* __restrict__ dst, int64_t dst_pitch) {
    const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t y = blockIdx.y;
    if (x >= w) return;
    double v = a * ((double)img[y * pitch + x] - lo) / span;
    v = v < 0.0 ? 0.0 : v;
    v = v > 255.0 ? 255.0 : v;
    dst[y * dst_pitch + x] = (uint8_t)(int)v;
}

// OpenCV drawing.cpp Circle(img, c, r, color, fill=true): the integer midpoint circle keeps the invariant
// err = dx^2 + dy^2 - r^2 <= 0 with dx maximal, so the span drawn on rows y0 +- j is exactly
// |x - x0| <= isqrt(r^2 - j^2) (checked against the stepwise algorithm for every r < 400 in the tests).
__device__ __forceinline__ int64_t isqrt64(int64_t v) {
    int64_t s = (int64_t)sqrt((double)v);
    while (s * s > v) --s;
    while ((s + 1) * (s + 1) <= v) ++s;
    return s;
}

__global__ __launch_bounds__(256) void k_fill_disc(uint16_t* __restrict__ img, int64_t h, int64_t w, int64_t pitch, int64_t x0,
                                                   int64_t y0, int64_t r, uint16_t value) {
    const int64_t x = x0 - r + (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t y = y0 - r + blockIdx.y;
    if (x < 0 || x >= w || y < 0 || y >= h || x > x0 + r) return;
    const int64_t ady = y > y0 ? y - y0 : y0 - y;
    const int64_t adx = x > x0 ? x - x0 : x0 - x;
    if (adx <= isqrt64(r * r - ady * ady)) img[y * pitch + x] = value;
}

__global__ __launch_bounds__(256) void k_downscale_mean(const uint16_t* __restrict__ img, int64_t h, int64_t w, int64_t pitch,
                                                        int f, int64_t oh, int64_t ow, double* __restrict__ dst) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= oh * ow) return;
    const int64_t oy = o / ow, ox = o - oy * ow;
    uint64_t s = 0;                     // zero padded blocks (block_reduce cval=0); the integer sum is exact
    for (int j = 0; j < f; ++j) {
        const int64_t y = oy * f + j;
        if (y >= h) break;
        for (int i = 0; i < f; ++i) {
            const int64_t x = ox * f + i;
            if (x < w) s += img[y * pitch + x];
        }
    }
    // mean of f*f float64 values k/65536: every partial sum is exactly representable
    dst[o] = ((double)s / 65536.0) / (double)(f * f);
}

}  // namespace

extern "C" int shg_crop_pad_u16(const uint16_t* src, int64_t h, int64_t w, int64_t pitch, uint16_t* dst, int64_t nw,
                                int64_t dst_pitch, int64_t sx0, int64_t dx0, int64_t n, int32_t fill, shg_stream_t stream) {
    SHG_REQUIRE(src && dst, SHG_E_ARG, "shg_crop_pad_u16: null pointer");
    SHG_REQUIRE(h > 0 && w > 0 && nw > 0 && pitch >= w && dst_pitch >= nw, SHG_E_ARG, "shg_crop_pad_u16: bad image size");
    SHG_REQUIRE(n >= 0 && sx0 >= 0 && sx0 + n <= w && dx0 >= 0 && dx0 + n <= nw, SHG_E_ARG, "shg_crop_pad_u16: copy window outside the images");
    SHG_REQUIRE(h < 65536, SHG_E_UNSUPPORTED, "shg_crop_pad_u16: more than 65535 rows");
    SHG_REQUIRE(fill <= 65535, SHG_E_ARG, "shg_crop_pad_u16: fill must be 0..65535, or negative for src[0][0]");
    dim3 grid((unsigned)((nw + 255) / 256), (unsigned)h);
    { SHG_PROF("crop_pad", shg::as_stream(stream)); k_crop_pad<<<grid, 256, 0, shg::as_stream(stream)>>>(src, pitch, dst, nw, dst_pitch, sx0, dx0, n, (uint16_t)(fill < 0 ? 0 : fill), fill < 0 ? 1 : 0); }
    return shg::check_launch("k_crop_pad");
}

extern "C" int shg_rescale_u16(const uint16_t* img, int64_t h, int64_t w, int64_t pitch, double lo, double hi, double alpha,
                               uint16_t* dst, int64_t dst_pitch, shg_stream_t stream) {
    SHG_REQUIRE(img && dst, SHG_E_ARG, "shg_rescale_u16: null pointer");
    SHG_REQUIRE(h > 0 && w > 0 && pitch >= w && dst_pitch >= w, SHG_E_ARG, "shg_rescale_u16: bad image size");
    SHG_REQUIRE(65535.0 >= hi && hi > lo, SHG_E_ARG, "shg_rescale_u16: need sat >= hi > lo (got lo=%g hi=%g)", lo, hi);   // assert, solex_util.py:521
    SHG_REQUIRE(h < 65536, SHG_E_UNSUPPORTED, "shg_rescale_u16: more than 65535 rows");
    dim3 grid((unsigned)((w + 255) / 256), (unsigned)h);
    { SHG_PROF("rescale", shg::as_stream(stream)); k_rescale<<<grid, 256, 0, shg::as_stream(stream)>>>(img, w, pitch, 65535.0 * alpha, lo, hi - lo, dst, dst_pitch); }
    return shg::check_launch("k_rescale");
}

extern "C" int shg_rescale_u8(const uint8_t* img, int64_t h, int64_t w, int64_t pitch, double lo, double hi, double alpha,
                              uint8_t* dst, int64_t dst_pitch, shg_stream_t stream) {
    SHG_REQUIRE(img && dst, SHG_E_ARG, "shg_rescale_u8: null pointer");
    SHG_REQUIRE(h > 0 && w > 0 && pitch >= w && dst_pitch >= w, SHG_E_ARG, "shg_rescale_u8: bad image size");
    SHG_REQUIRE(255.0 >= hi && hi > lo, SHG_E_ARG, "shg_rescale_u8: need sat >= hi > lo (got lo=%g hi=%g)", lo, hi);
    SHG_REQUIRE(h < 65536, SHG_E_UNSUPPORTED, "shg_rescale_u8: more than 65535 rows");
    dim3 grid((unsigned)((w + 255) / 256), (unsigned)h);
    { SHG_PROF("rescale", shg::as_stream(stream)); k_rescale_u8<<<grid, 256, 0, shg::as_stream(stream)>>>(img, w, pitch, 255.0 * alpha, lo, hi - lo, dst, dst_pitch); }
    return shg::check_launch("k_rescale_u8");
}

extern "C" int shg_fill_disc_u16(uint16_t* img, int64_t h, int64_t w, int64_t pitch, int64_t x0, int64_t y0, int64_t r,
                                 uint16_t value, int32_t* scratch, shg_stream_t stream) {
    SHG_REQUIRE(img, SHG_E_ARG, "shg_fill_disc_u16: null pointer");
    (void)scratch;   // kept in the ABI; the span table it once held is now a closed form
    SHG_REQUIRE(h > 0 && w > 0 && pitch >= w, SHG_E_ARG, "shg_fill_disc_u16: bad image size");
    SHG_REQUIRE(r >= 0 && r < 32768, SHG_E_UNSUPPORTED, "shg_fill_disc_u16: radius %lld out of range", (long long)r);
    hipStream_t st = shg::as_stream(stream);
    dim3 grid((unsigned)((2 * r + 1 + 255) / 256), (unsigned)(2 * r + 1));
    { SHG_PROF("fill_disc", st); k_fill_disc<<<grid, 256, 0, st>>>(img, h, w, pitch, x0, y0, r, value); }
    return shg::check_launch("k_fill_disc");
}

extern "C" int shg_downscale_mean_u16(const uint16_t* img, int64_t h, int64_t w, int64_t pitch, int factor, double* dst,
                                      shg_stream_t stream) {
    SHG_REQUIRE(img && dst, SHG_E_ARG, "shg_downscale_mean_u16: null pointer");
    SHG_REQUIRE(h > 0 && w > 0 && pitch >= w && factor >= 1 && factor <= 64, SHG_E_ARG, "shg_downscale_mean_u16: bad size");
    const int64_t oh = (h + factor - 1) / factor, ow = (w + factor - 1) / factor;
    { SHG_PROF("downscale", shg::as_stream(stream)); k_downscale_mean<<<(unsigned)((oh * ow + 255) / 256), 256, 0, shg::as_stream(stream)>>>(img, h, w, pitch, factor, oh, ow, dst); }
    return shg::check_launch("k_downscale_mean");
}

// The three rescale_brightness calls of image_process and the protuberance disc (solex_util.py:539-547) in one call and
// ONE pass: frame and cl1 are read once, the three products written once (the separate calls read the frame twice and
// went back over protus for the disc).  Same arithmetic per pixel as k_rescale / k_fill_disc.
namespace {
struct Bounds6 { double lo[3], span[3]; };

__device__ __forceinline__ uint16_t rescale1(double px, double lo, double span) {
    double v = 65535.0 * (px - lo) / span;             // (float(sat) * alpha * (img - lo)) / (hi - lo), alpha = 1
    v = v < 0.0 ? 0.0 : v;
    v = v > 65535.0 ? 65535.0 : v;
    return (uint16_t)(int)v;
}

__global__ __launch_bounds__(256) void k_products(const uint16_t* __restrict__ frame, int64_t frame_pitch,
                                                  const uint16_t* __restrict__ cl1, int64_t cl1_pitch, int64_t w, Bounds6 b,
                                                  uint16_t* __restrict__ hc, uint16_t* __restrict__ protus, uint16_t* __restrict__ cc,
                                                  int64_t dst_pitch, int64_t x0, int64_t y0, int64_t r) {
    const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t y = blockIdx.y;
    if (x >= w) return;
    const double f = (double)frame[y * frame_pitch + x];
    const double c = (double)cl1[y * cl1_pitch + x];
    hc[y * dst_pitch + x] = rescale1(f, b.lo[0], b.span[0]);
    uint16_t p = rescale1(f, b.lo[1], b.span[1]);
    if (r > 0) {                                            // cv2.circle(frame_protus, (x0, y0), r, 80, -1)
        const int64_t ady = y > y0 ? y - y0 : y0 - y;
        const int64_t adx = x > x0 ? x - x0 : x0 - x;
        if (ady <= r && adx <= r && adx <= isqrt64(r * r - ady * ady)) p = 80;
    }
    protus[y * dst_pitch + x] = p;
    cc[y * dst_pitch + x] = rescale1(c, b.lo[2], b.span[2]);
}
}  // namespace

extern "C" int shg_contrast_products_u16(const uint16_t* frame, int64_t frame_pitch, const uint16_t* cl1, int64_t cl1_pitch, int64_t h,
                                         int64_t w, const double* lo_hi6, uint16_t* high_contrast, uint16_t* protus, uint16_t* cc,
                                         int64_t dst_pitch, int64_t disc_x0, int64_t disc_y0, int64_t disc_r, shg_stream_t stream) {
    SHG_REQUIRE(frame && cl1 && lo_hi6 && high_contrast && protus && cc, SHG_E_ARG, "shg_contrast_products_u16: null pointer");
    SHG_REQUIRE(h > 0 && w > 0 && frame_pitch >= w && cl1_pitch >= w && dst_pitch >= w, SHG_E_ARG, "shg_contrast_products_u16: bad image size");
    SHG_REQUIRE(h < 65536, SHG_E_UNSUPPORTED, "shg_contrast_products_u16: more than 65535 rows");
    SHG_REQUIRE(disc_r < 32768, SHG_E_UNSUPPORTED, "shg_contrast_products_u16: radius %lld out of range", (long long)disc_r);
    Bounds6 b;
    for (int i = 0; i < 3; ++i) {
        const double lo = lo_hi6[2 * i], hi = lo_hi6[2 * i + 1];
        SHG_REQUIRE(65535.0 >= hi && hi > lo, SHG_E_ARG, "shg_contrast_products_u16: need sat >= hi > lo (got lo=%g hi=%g)", lo, hi);   // assert, solex_util.py:521
        b.lo[i] = lo;
        b.span[i] = hi - lo;
    }
    hipStream_t st = shg::as_stream(stream);
    dim3 grid((unsigned)((w + 255) / 256), (unsigned)h);
    { SHG_PROF("products", st); k_products<<<grid, 256, 0, st>>>(frame, frame_pitch, cl1, cl1_pitch, w, b, high_contrast, protus, cc, dst_pitch,
                                                                  disc_x0, disc_y0, disc_r > 0 ? disc_r : 0); }
    return shg::check_launch("k_products");
}
