// Small per-disk image passes: crop/pad, brightness rescale, protuberance disc,
// 4x4 block mean.  Reference: Solex_recon.py:155-171 (crop), solex_util.py:519-525
// (rescale_brightness), solex_util.py:542-547 (cv2.circle), ellipse_to_circle.py:299-302
// (downscale_local_mean).  All are single streaming passes over a few-MB image.
#include <algorithm>
#include "shg_common.h"

namespace {

// grid (x, rows, disks)
__global__ __launch_bounds__(256) void k_crop_pad(shg::PtrBatch srcs, int64_t pitch, shg::PtrBatch dsts,
                                                  int64_t nw, int64_t dst_pitch, int64_t sx0, int64_t dx0, int64_t n,
                                                  uint16_t fill, int fill_is_src00) {
    const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t y = blockIdx.y;
    if (x >= nw) return;
    const uint16_t* __restrict__ src = srcs.at<const uint16_t>(blockIdx.z);
    uint16_t* __restrict__ dst = dsts.at<uint16_t>(blockIdx.z);
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
                                                        int f, int64_t oh, int64_t ow, int vec4, double* __restrict__ dst) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= oh * ow) return;
    const int64_t oy = (int64_t)((uint32_t)o / (uint32_t)ow), ox = o - oy * ow;      // (oh * ow < 2^31: checked by the entry point)
    uint64_t s = 0;                     // zero padded blocks (block_reduce cval=0); the integer sum is exact
    if (f == 4 && vec4 && ox * 4 + 4 <= w && oy * 4 + 4 <= h) {
        // the usual block (ellipse_to_circle.py:299): four 8-byte loads in flight instead of sixteen 2-byte ones
        uint2 q[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) q[j] = *reinterpret_cast<const uint2*>(img + (oy * 4 + j) * pitch + ox * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) s += (uint64_t)((q[j].x & 0xffffu) + (q[j].x >> 16) + (q[j].y & 0xffffu) + (q[j].y >> 16));
        dst[o] = ((double)s / 65536.0) / 16.0;
        return;
    }
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
    return shg::crop_pad_batch(&src, 1, h, w, pitch, &dst, nw, dst_pitch, sx0, dx0, n, fill, stream);
}

int shg::crop_pad_batch(const uint16_t* const* host_srcs, int64_t k, int64_t h, int64_t w, int64_t pitch, uint16_t* const* host_dsts, int64_t nw,
                        int64_t dst_pitch, int64_t sx0, int64_t dx0, int64_t n, int32_t fill, shg_stream_t stream) {
    SHG_REQUIRE(host_srcs && host_dsts && k > 0, SHG_E_ARG, "shg_crop_pad_u16: null pointer");
    SHG_REQUIRE(h > 0 && w > 0 && nw > 0 && pitch >= w && dst_pitch >= nw, SHG_E_ARG, "shg_crop_pad_u16: bad image size");
    SHG_REQUIRE(n >= 0 && sx0 >= 0 && sx0 + n <= w && dx0 >= 0 && dx0 + n <= nw, SHG_E_ARG, "shg_crop_pad_u16: copy window outside the images");
    SHG_REQUIRE(h < 65536, SHG_E_UNSUPPORTED, "shg_crop_pad_u16: more than 65535 rows");
    SHG_REQUIRE(fill <= 65535, SHG_E_ARG, "shg_crop_pad_u16: fill must be 0..65535, or negative for src[0][0]");
    for (int64_t i = 0; i < k; ++i) SHG_REQUIRE(host_srcs[i] && host_dsts[i], SHG_E_ARG, "shg_crop_pad_u16: null image");
    hipStream_t st = shg::as_stream(stream);
    SHG_PROF("crop_pad", st);
    for (int64_t i0 = 0; i0 < k; i0 += shg::kMaxBatch) {
        const int m = (int)std::min<int64_t>(shg::kMaxBatch, k - i0);
        dim3 grid((unsigned)((nw + 255) / 256), (unsigned)h, (unsigned)m);
        k_crop_pad<<<grid, 256, 0, st>>>(shg::make_batch(host_srcs, (int)i0, m), pitch, shg::make_batch(host_dsts, (int)i0, m), nw, dst_pitch, sx0, dx0, n,
                                         (uint16_t)(fill < 0 ? 0 : fill), fill < 0 ? 1 : 0);
        if (int e = shg::check_launch("k_crop_pad")) return e;
    }
    return 0;
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
    SHG_REQUIRE(oh * ow < (1ll << 31), SHG_E_UNSUPPORTED, "shg_downscale_mean_u16: image too large");
    { SHG_PROF("downscale", shg::as_stream(stream)); k_downscale_mean<<<(unsigned)((oh * ow + 255) / 256), 256, 0, shg::as_stream(stream)>>>(
          img, h, w, pitch, factor, oh, ow, (reinterpret_cast<uintptr_t>(img) & 7) == 0 && pitch % 4 == 0, dst); }
    return shg::check_launch("k_downscale_mean");
}

// The three rescale_brightness calls of image_process and the protuberance disc (solex_util.py:539-547) in one call and
// ONE pass: frame and cl1 are read once, the three products written once (the separate calls read the frame twice and
// went back over protus for the disc).  Same arithmetic per pixel as k_rescale / k_fill_disc.
namespace {
struct Bounds6 { double lo[3], span[3]; };
constexpr int kProductsBatch = 24;                  // disks per launch of the products kernels (their bounds and five pointer tables travel by value: 2.2 KB of the 4 KB of kernel arguments; a 21-disk stack is one launch -- three launches of eight had three tails)
using ProdPtrs = shg::PtrBatchN<kProductsBatch>;
struct BoundsBatch { Bounds6 v[kProductsBatch]; };

__device__ __forceinline__ uint16_t rescale1(double px, double lo, double span) {
    double v = 65535.0 * (px - lo) / span;             // (float(sat) * alpha * (img - lo)) / (hi - lo), alpha = 1
    v = v < 0.0 ? 0.0 : v;
    v = v > 65535.0 ? 65535.0 : v;
    return (uint16_t)(int)v;
}

// The six rescale bounds of image_process (solex_util.py:535-546) from the five order statistics the contrast stage left on
// the device, formed where they are used -- the products kernel no longer waits for the host to do this arithmetic (a stream
// synchronisation, a wake-up and a launch in the middle of every scan).  s = {frame: two order statistics of the 99.9999th
// percentile, cl1: two of the 10th percentile, cl1 max}; NumPy's _lerp.  -> false where rescale_brightness's assert fails
// (the kernel then writes nothing and the host, which sees the same numbers, reports it).
struct StatsSource { const double* stats5; double g_bright, g_dark; double* mirror5; };
__device__ __forceinline__ double lerp_np(double a, double b, double gamma) {
    const double diff = b - a;
    return gamma >= 0.5 ? b - diff * (1 - gamma) : a + diff * gamma;
}
__device__ __forceinline__ bool bounds_from_stats(const StatsSource& src, int disk, Bounds6& b) {
    const double* s = src.stats5 + 5 * disk;
    const double bright = lerp_np(s[0], s[1], src.g_bright);                 // basically the same as max
    const double dark_clahe = lerp_np(s[2], s[3], src.g_dark);
    const double bright_clahe = (double)(int64_t)s[4];
    b.lo[0] = bright * 0.25; b.span[0] = bright - b.lo[0];
    b.lo[1] = 0.0;           b.span[1] = bright * 0.18 - 0.0;
    b.lo[2] = dark_clahe;    b.span[2] = bright_clahe - dark_clahe;
    return 65535 >= bright && bright > bright * 0.25 && 65535 >= bright * 0.18 && bright * 0.18 > 0 && 65535 >= bright_clahe &&
           bright_clahe > dark_clahe;
}

// Eight pixels per lane: 16-byte loads of frame and cl1, 16-byte stores of the three products (rows 16-byte aligned,
// pitches multiples of 8; a row's last, partial vector goes pixel by pixel).  Lanes are dealt (row, 8-column vector) pairs in one flat
// sequence -- a width just past a multiple of 2048 pixels (2096 at C2) would leave every second workgroup of a (x, y) grid with half a
// dozen lanes to do.
// grid (ceil(vectors / 256), 1, disks)
//
// Round 6, second half: the kernel issued 70 vector instructions a pixel (96 M wave instructions a C4 launch, profiles/
// r06_sq_k_products8.json of the round's fourth collection), a third of them per THREAD, not per pixel: the bounds from the statistics, three IEEE
// divisions for 1 / span, the disc's half width by a float64 square root and 64-bit fix-up loops, 64-bit addresses -- for eight
// pixels.  Now: 1 / span is the hardware's reciprocal and two Newton steps (its error, 2^-52, moves q by 1e-11: the near-whole test
// below allows 1e-7); the half width is a float square root and two 32-bit fix-ups (radii below 32768: the host checks); row and
// column by a multiply-high; 32-bit byte offsets off the scalar image bases.  Per pixel: the three rescales without a branch each --
// every pixel takes the short way, the lane notes whether any of its 24 quotients sat within 1e-7 of a whole number, and only such a
// lane (one in ~10^5) goes over its eight pixels again with the exact division.  70 -> 45 instructions a pixel; a single disk (C2)
// 12.8 -> 10.4 us.  A 21-disk stack stays at 162 - 167 us: with the instructions gone it is its five streams (two read, three
// written, 718 MB at 4.4 TB/s) that set the pace -- a plain copy of that size runs at 5.4 TB/s on this part (tools/probes/
// write_rate.py), four vectors a thread measured 163 and 189 us on two boxes (profiles/r06_sweeps.txt).
struct ProductsArgs {
    ProdPtrs frames;
    int64_t frame_pitch;
    ProdPtrs cl1s;
    int64_t cl1_pitch, h, w;
    BoundsBatch bb;
    ProdPtrs hcs, protuss, ccs;
    int64_t dst_pitch, x0, y0, r;
    StatsSource stats;
    uint32_t nv_magic, nv_shift;         // flat / (vectors a row) = __umulhi(flat, nv_magic) >> nv_shift (see k_warp_rows8's host side)
};

// rescale1 without the division for all but a handful of pixels.  t = 65535 * (px - lo) is rounded as the reference rounds it;
// q = t * (1 / span) is within 2^-50 * 65535 < 1e-10 of the correctly rounded t / span, so unless q sits that close to a whole number
// both truncate to the same integer (and clamp alike beyond 0 / 65535); a pixel near a whole number takes the exact division (three
// float64 divisions per pixel made the first k_products ALU bound).  This is the short way, and whether the exact division has to
// decide: r = trunc(q), the fraction fr = q - r is >= 0 here, "within 1e-7 of a whole number" is fr's high word outside [A, B) = [that
// of 1e-7, that of 1 - 1e-7) (a shade wider: more pixels in doubt, none fewer), i.e. (hi - A) >= B - A unsigned -- and a lane only has
// to know whether ANY of its 24 quotients is: the largest hi - A, one v_max_u32 a quotient instead of a compare and a scalar or (and 24
// flags kept in registers until the end: 134 of them, three waves a SIMD).  A quotient beyond int32 or infinite has a "fraction" >= 1: in doubt.
constexpr uint32_t kFracA = 0x3E7AD7F3u, kFracB = 0x3FEFFFFFu;
__device__ __forceinline__ uint32_t rescale1_quick(double px, double lo, double inv_span, uint32_t& worst) {
    // (a quotient at or below 0 -- a black pixel under lo = 0, a pixel equal to an integral lo: every such one exactly 0, i.e. "whole" --
    // gives 0 whichever way it is computed: lifted to 0.25 it does not send its lane the long way; a NaN -- 0 x inf -- likewise)
    const double q = fmax((65535.0 * (px - lo)) * inv_span, 0.25);
    const int r = (int)q;                                                  // saturating
    const double fr = q - (double)r;
    worst = max(worst, (uint32_t)__double2hiint(fr) - kFracA);
    return (uint32_t)min(r, 65535);
}
__device__ __forceinline__ double recip_newton(double d) {                // 1 / d to an ulp or so (0, inf, NaN: a NaN -- every pixel in doubt)
    double y = __builtin_amdgcn_rcp(d);
    y = fma(y, fma(-d, y, 1.0), y);
    return fma(y, fma(-d, y, 1.0), y);
}
// floor(sqrt(v)) for v < 2^31
__device__ __forceinline__ uint32_t isqrt31(uint32_t v) {
    uint32_t s = (uint32_t)__builtin_amdgcn_sqrtf((float)v);             // within 1 of the root (v < 2^31, the root < 46341: float's 2^-22 is 0.01)
    s = s > 46340u ? 46340u : s;
#pragma unroll
    for (int i = 0; i < 2; ++i) s -= (s * s > v) ? 1u : 0u;
#pragma unroll
    for (int i = 0; i < 2; ++i) s += ((s + 1u) * (s + 1u) <= v) ? 1u : 0u;
    return s;
}

__global__ __launch_bounds__(256) void k_products8(const ProductsArgs kargs) {
    const uint32_t h = (uint32_t)kargs.h, w = (uint32_t)kargs.w;
    const uint32_t frame_pitch = (uint32_t)kargs.frame_pitch, cl1_pitch = (uint32_t)kargs.cl1_pitch, dst_pitch = (uint32_t)kargs.dst_pitch;
    const int64_t x0 = kargs.x0, y0 = kargs.y0;
    const int r = (int)kargs.r;
    const StatsSource& stats = kargs.stats;
    const char* __restrict__ frame = kargs.frames.at<const char>(blockIdx.z);
    const char* __restrict__ cl1 = kargs.cl1s.at<const char>(blockIdx.z);
    char* __restrict__ hc = kargs.hcs.at<char>(blockIdx.z);
    char* __restrict__ protus = kargs.protuss.at<char>(blockIdx.z);
    char* __restrict__ cc = kargs.ccs.at<char>(blockIdx.z);
    Bounds6 b = kargs.bb.v[blockIdx.z];
    if (stats.stats5) {
        const bool ok = bounds_from_stats(stats, blockIdx.z, b);
        if (stats.mirror5 && blockIdx.x == 0 && threadIdx.x < 5) stats.mirror5[5 * blockIdx.z + threadIdx.x] = stats.stats5[5 * blockIdx.z + threadIdx.x];
        if (!ok) return;
    }
    const double i0 = recip_newton(b.span[0]), i1 = recip_newton(b.span[1]), i2 = recip_newton(b.span[2]);
    const uint32_t nv = (w + 7u) / 8u, total = nv * h;
    typedef unsigned int __attribute__((ext_vector_type(4))) u32x4;
    {
        const uint32_t flat = (uint32_t)blockIdx.x * 256u + threadIdx.x;
        if (flat >= total) return;
        const uint32_t y = __umulhi(flat, kargs.nv_magic) >> kargs.nv_shift;
        const uint32_t x = (flat - y * nv) * 8u;
        // the disc's span on this row: [x0 - half, x0 + half] (cv2.circle(frame_protus, (x0, y0), r, 80, -1)), relative to x
        int d_lo = 1, d_hi = 0;
        if (r > 0) {
            const int64_t ady = (int64_t)y > y0 ? (int64_t)y - y0 : y0 - (int64_t)y;
            if (ady <= (int64_t)r) {
                const int64_t half = (int64_t)isqrt31((uint32_t)(r * r) - (uint32_t)((int)ady * (int)ady));
                const int64_t lim = 1ll << 30;                             // (columns are below 2^30: clamping there keeps every compare)
                d_lo = (int)(std::min(std::max(x0 - half, -lim), lim) - (int64_t)x);
                d_hi = (int)(std::min(std::max(x0 + half, -lim), lim) - (int64_t)x);
            }
        }
        if (x + 8u <= w) {
            const u32x4 qf = *reinterpret_cast<const u32x4*>(frame + ((__umul24(y, frame_pitch) + x) << 1));
            const u32x4 qc = *reinterpret_cast<const u32x4*>(cl1 + ((__umul24(y, cl1_pitch) + x) << 1));
            const uint32_t fw[4] = {qf.x, qf.y, qf.z, qf.w}, cw[4] = {qc.x, qc.y, qc.z, qc.w};
            uint32_t oh[4], op[4], oc[4];
            uint32_t worst = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t h2[2], p2[2], c2[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const double f = (double)((fw[j] >> (16 * e)) & 0xffffu);
                    const double c = (double)((cw[j] >> (16 * e)) & 0xffffu);
                    const int k = 2 * j + e;
                    h2[e] = rescale1_quick(f, b.lo[0], i0, worst);
                    const uint32_t pq = rescale1_quick(f, b.lo[1], i1, worst);
                    p2[e] = (k >= d_lo && k <= d_hi) ? 80u : pq;
                    c2[e] = rescale1_quick(c, b.lo[2], i2, worst);
                }
                oh[j] = h2[0] | (h2[1] << 16);
                op[j] = p2[0] | (p2[1] << 16);
                oc[j] = c2[0] | (c2[1] << 16);
            }
            if (worst >= kFracB - kFracA) {                                // rare: the reference's own arithmetic for the lane's eight pixels
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint32_t h2[2], p2[2], c2[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const double f = (double)((fw[j] >> (16 * e)) & 0xffffu);
                        const double c = (double)((cw[j] >> (16 * e)) & 0xffffu);
                        const int k = 2 * j + e;
                        h2[e] = rescale1(f, b.lo[0], b.span[0]);
                        p2[e] = (k >= d_lo && k <= d_hi) ? 80u : (uint32_t)rescale1(f, b.lo[1], b.span[1]);
                        c2[e] = rescale1(c, b.lo[2], b.span[2]);
                    }
                    oh[j] = h2[0] | (h2[1] << 16);
                    op[j] = p2[0] | (p2[1] << 16);
                    oc[j] = c2[0] | (c2[1] << 16);
                }
            }
            // the products are final: nobody on the GPU reads them again, so they bypass the caches (frame and cl1 stay)
            const uint32_t o = (__umul24(y, dst_pitch) + x) << 1;
            __builtin_nontemporal_store((u32x4){oh[0], oh[1], oh[2], oh[3]}, reinterpret_cast<u32x4*>(hc + o));
            __builtin_nontemporal_store((u32x4){op[0], op[1], op[2], op[3]}, reinterpret_cast<u32x4*>(protus + o));
            __builtin_nontemporal_store((u32x4){oc[0], oc[1], oc[2], oc[3]}, reinterpret_cast<u32x4*>(cc + o));
        } else {
            for (uint32_t xi = x; xi < w; ++xi) {
                const double f = (double)*reinterpret_cast<const uint16_t*>(frame + ((__umul24(y, frame_pitch) + xi) << 1));
                const double c = (double)*reinterpret_cast<const uint16_t*>(cl1 + ((__umul24(y, cl1_pitch) + xi) << 1));
                const uint32_t o = (__umul24(y, dst_pitch) + xi) << 1;
                const int k = (int)(xi - x);
                *reinterpret_cast<uint16_t*>(hc + o) = rescale1(f, b.lo[0], b.span[0]);
                *reinterpret_cast<uint16_t*>(protus + o) = (k >= d_lo && k <= d_hi) ? (uint16_t)80 : rescale1(f, b.lo[1], b.span[1]);
                *reinterpret_cast<uint16_t*>(cc + o) = rescale1(c, b.lo[2], b.span[2]);
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_products(ProdPtrs frames, int64_t frame_pitch,
                                                  ProdPtrs cl1s, int64_t cl1_pitch, int64_t w, BoundsBatch bb,
                                                  ProdPtrs hcs, ProdPtrs protuss, ProdPtrs ccs,
                                                  int64_t dst_pitch, int64_t x0, int64_t y0, int64_t r, StatsSource stats) {
    const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t y = blockIdx.y;
    if (x >= w) return;
    const uint16_t* __restrict__ frame = frames.at<const uint16_t>(blockIdx.z);
    const uint16_t* __restrict__ cl1 = cl1s.at<const uint16_t>(blockIdx.z);
    uint16_t* __restrict__ hc = hcs.at<uint16_t>(blockIdx.z);
    uint16_t* __restrict__ protus = protuss.at<uint16_t>(blockIdx.z);
    uint16_t* __restrict__ cc = ccs.at<uint16_t>(blockIdx.z);
    Bounds6 b = bb.v[blockIdx.z];
    if (stats.stats5) {
        const bool ok = bounds_from_stats(stats, blockIdx.z, b);
        if (stats.mirror5 && blockIdx.x == 0 && threadIdx.x < 5) stats.mirror5[5 * blockIdx.z + threadIdx.x] = stats.stats5[5 * blockIdx.z + threadIdx.x];
        if (!ok) return;
    }
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
    return shg::contrast_products_batch(&frame, frame_pitch, &cl1, cl1_pitch, 1, h, w, lo_hi6, &high_contrast, &protus, &cc, dst_pitch, disc_x0,
                                        disc_y0, disc_r, stream);
}

// k disks of one shape in one launch (per kProductsBatch): host_lo_hi6 is [k][6] -- or NULL, and the bounds are formed on the
// device from stats5 [k][5] (device; the order statistics shg_contrast_stats_u16 leaves) with the _lerp weights g_bright /
// g_dark; mirror5 (may be NULL, e.g. GPU-mapped host memory): where the kernel also leaves those statistics for the host.
int shg::contrast_products_batch(const uint16_t* const* host_frames, int64_t frame_pitch, const uint16_t* const* host_cl1, int64_t cl1_pitch,
                                 int64_t k, int64_t h, int64_t w, const double* host_lo_hi6, uint16_t* const* host_hc,
                                 uint16_t* const* host_protus, uint16_t* const* host_cc, int64_t dst_pitch, int64_t disc_x0, int64_t disc_y0,
                                 int64_t disc_r, shg_stream_t stream, const double* stats5, double g_bright, double g_dark, double* mirror5) {
    SHG_REQUIRE(host_frames && host_cl1 && (host_lo_hi6 || stats5) && host_hc && host_protus && host_cc && k > 0, SHG_E_ARG, "shg_contrast_products_u16: null pointer");
    SHG_REQUIRE(h > 0 && w > 0 && frame_pitch >= w && cl1_pitch >= w && dst_pitch >= w, SHG_E_ARG, "shg_contrast_products_u16: bad image size");
    SHG_REQUIRE(h < 65536, SHG_E_UNSUPPORTED, "shg_contrast_products_u16: more than 65535 rows");
    SHG_REQUIRE(disc_r < 32768, SHG_E_UNSUPPORTED, "shg_contrast_products_u16: radius %lld out of range", (long long)disc_r);
    uintptr_t ptrs = 0;
    for (int64_t d = 0; d < k; ++d) {
        SHG_REQUIRE(host_frames[d] && host_cl1[d] && host_hc[d] && host_protus[d] && host_cc[d], SHG_E_ARG, "shg_contrast_products_u16: null image");
        ptrs |= reinterpret_cast<uintptr_t>(host_frames[d]) | reinterpret_cast<uintptr_t>(host_cl1[d]) | reinterpret_cast<uintptr_t>(host_hc[d]) |
                reinterpret_cast<uintptr_t>(host_protus[d]) | reinterpret_cast<uintptr_t>(host_cc[d]);
        for (int i = 0; i < 3 && host_lo_hi6; ++i) {
            const double lo = host_lo_hi6[6 * d + 2 * i], hi = host_lo_hi6[6 * d + 2 * i + 1];
            SHG_REQUIRE(65535.0 >= hi && hi > lo, SHG_E_ARG, "shg_contrast_products_u16: need sat >= hi > lo (got lo=%g hi=%g)", lo, hi);   // assert, solex_util.py:521
        }
    }
    hipStream_t st = shg::as_stream(stream);
    // (the vector kernel's 32-bit offsets and multiply-high: every image below 4 GiB, pitches in 24 bits, at least two vectors a row)
    const bool vec = (ptrs & 15) == 0 && frame_pitch % 8 == 0 && cl1_pitch % 8 == 0 && dst_pitch % 8 == 0 && w >= 9 && w < (1ll << 30) &&
                     std::max(frame_pitch, std::max(cl1_pitch, dst_pitch)) < (1ll << 24) && h * std::max(frame_pitch, std::max(cl1_pitch, dst_pitch)) < (1ll << 31) &&
                     ((w + 7) / 8) * h < (1ll << 31);
    SHG_PROF("products", st);
    for (int64_t i0 = 0; i0 < k; i0 += kProductsBatch) {
        const int m = (int)std::min<int64_t>(kProductsBatch, k - i0);
        BoundsBatch bb = {};
        for (int d = 0; d < m && host_lo_hi6; ++d)
            for (int i = 0; i < 3; ++i) {
                bb.v[d].lo[i] = host_lo_hi6[6 * (i0 + d) + 2 * i];
                bb.v[d].span[i] = host_lo_hi6[6 * (i0 + d) + 2 * i + 1] - bb.v[d].lo[i];
            }
        const StatsSource src = {stats5 ? stats5 + 5 * i0 : nullptr, g_bright, g_dark, mirror5 ? mirror5 + 5 * i0 : nullptr};
        const ProdPtrs f = shg::make_batch_n<kProductsBatch>(host_frames, (int)i0, m), c = shg::make_batch_n<kProductsBatch>(host_cl1, (int)i0, m),
                       hc = shg::make_batch_n<kProductsBatch>(host_hc, (int)i0, m), pr = shg::make_batch_n<kProductsBatch>(host_protus, (int)i0, m),
                       cc = shg::make_batch_n<kProductsBatch>(host_cc, (int)i0, m);
        if (vec) {
            const int64_t nv = (w + 7) / 8, lanes = nv * h;
            uint32_t nv_shift = 0;
            while ((2ull << nv_shift) < (uint64_t)nv) ++nv_shift;
            const uint32_t nv_magic = (uint32_t)(((1ull << (32 + nv_shift)) + (uint64_t)nv - 1) / (uint64_t)nv);
            const ProductsArgs args{f, frame_pitch, c, cl1_pitch, h, w, bb, hc, pr, cc, dst_pitch, disc_x0, disc_y0, disc_r > 0 ? disc_r : 0, src, nv_magic, nv_shift};
            dim3 grid((unsigned)((lanes + 255) / 256), 1u, (unsigned)m);
            if (int e = shg::launch(k_products8, grid, dim3(256), 0, st, args, "k_products8")) return e;
        } else {
            dim3 grid((unsigned)((w + 255) / 256), (unsigned)h, (unsigned)m);
            k_products<<<grid, 256, 0, st>>>(f, frame_pitch, c, cl1_pitch, w, bb, hc, pr, cc, dst_pitch, disc_x0, disc_y0, disc_r > 0 ? disc_r : 0, src);
            if (int e = shg::check_launch("k_products")) return e;
        }
    }
    return 0;
}
