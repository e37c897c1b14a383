// "Stubborn" transversalium: the line filter of apply_lin_filter + fix_edge_effect
// (reference solex_util.py:277-375), which the reference builds from three cv2.filter2D calls
// on log(img) with 11x101 and 1x101 box kernels.
//
// cv2.filter2D is not reproducible bit for bit (OpenCV sends kernels this large through a DFT);
// what is computed here is the exact correlation with BORDER_REFLECT_101 that the DFT approximates:
// float64 sums in a fixed order, rounded once to the image's float type (float32 for a uint16 image,
// because np.log(uint16) is float32; float64 for a de-vignetted float64 image).  The oracle
// (oracle/shg_oracle.py apply_lin_filter) adds in the same order, so the two agree bit for bit up to
// the device's exp / log.
//
// The filter is separable:
//   k_lin_row_sums : HL[y][x] = sum of the linlen logs around x on row y (LDS-staged, left to right),
//                    HF the same for the image whose flagged rows are replaced by the mean of the
//                    nearest unflagged rows above and below (:306-317).  Unflagged rows: HF = HL.
//   k_lin_apply    : delta = HL/linlen - (sum of HF over the 2*half_width neighbouring rows)/(2*half_width*linlen),
//                    the limb clean-up of fix_edge_effect as a per-row column select, then
//                    img * exp(-delta * taper[y]), saturate, truncate.
// float32 logs of the 65536 possible pixel values come from a host table (NumPy's own float32 log).
#include <math.h>
#include "shg_common.h"

namespace {

constexpr int SEG = 512;            // output columns per workgroup
constexpr int MAXLIN = 512;         // longest supported horizontal window

template <bool F64>
__device__ __forceinline__ double log_px(const uint16_t* __restrict__ img, int64_t pitch, const double* __restrict__ row_factor,
                                         const float* __restrict__ lut, int64_t y, int64_t x) {
    const uint16_t v = img[y * pitch + x];
    if (F64) return log((double)v * row_factor[y]);
    return (double)lut[v];
}

template <bool F64>
__global__ __launch_bounds__(256) void k_lin_row_sums(const uint16_t* __restrict__ img, int64_t w, int64_t pitch,
                                                      const double* __restrict__ row_factor, const float* __restrict__ lut,
                                                      const uint8_t* __restrict__ flagged, const int32_t* __restrict__ up,
                                                      const int32_t* __restrict__ dn, int linlen, double* __restrict__ hl,
                                                      double* __restrict__ hf) {
    __shared__ double sl[SEG + MAXLIN];
    __shared__ double sf[SEG + MAXLIN];
    const int64_t y = blockIdx.y;
    const int64_t x0 = (int64_t)blockIdx.x * SEG;
    const int half = linlen / 2;
    const int n_out = (int)min((int64_t)SEG, w - x0);
    const int n_in = n_out + linlen - 1;
    const bool flag = flagged[y] != 0;
    const int64_t yu = flag ? up[y] : -1, yd = flag ? dn[y] : -1;
    for (int i = threadIdx.x; i < n_in; i += 256) {
        const int64_t xx = shg::reflect101(x0 - half + i, w);
        sl[i] = log_px<F64>(img, pitch, row_factor, lut, y, xx);
        if (flag) {
            if (F64) {
                const double a = yu >= 0 ? log_px<true>(img, pitch, row_factor, lut, yu, xx) : 0.0;
                const double b = yd >= 0 ? log_px<true>(img, pitch, row_factor, lut, yd, xx) : 0.0;
                double f = a / 2;
                f += b / 2;
                sf[i] = f;
            } else {
                const float a = yu >= 0 ? lut[img[yu * pitch + xx]] : 0.f;
                const float b = yd >= 0 ? lut[img[yd * pitch + xx]] : 0.f;
                float f = a / 2;
                f += b / 2;
                sf[i] = (double)f;
            }
        }
    }
    __syncthreads();
    for (int o = threadIdx.x; o < n_out; o += 256) {
        double acc = sl[o];
        for (int d = 1; d < linlen; ++d) acc += sl[o + d];
        hl[y * w + x0 + o] = acc;
        if (flag) {
            double af = sf[o];
            for (int d = 1; d < linlen; ++d) af += sf[o + d];
            hf[y * w + x0 + o] = af;
        } else {
            hf[y * w + x0 + o] = acc;
        }
    }
}

template <bool F64>
__global__ __launch_bounds__(256) void k_lin_apply(const uint16_t* __restrict__ img, int64_t h, int64_t w, int64_t pitch,
                                                   const double* __restrict__ row_factor, const double* __restrict__ hl,
                                                   const double* __restrict__ hf, int linlen, int half_width,
                                                   const double* __restrict__ taper, const int32_t* __restrict__ xa,
                                                   const int32_t* __restrict__ xb, const uint8_t* __restrict__ edge, int edge_half,
                                                   uint16_t* __restrict__ dst, int64_t dst_pitch) {
    const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t y = blockIdx.y;
    if (x >= w) return;
    double v = (double)img[y * pitch + x];
    if (F64) v = v * row_factor[y];
    const int64_t a = xa[y], b = xb[y];
    double delta = 0.0;
    if (x >= a && x < b) {
        int64_t xs = x;
        if ((edge[y] & 1) && x < a + edge_half) xs = a + edge_half;
        else if ((edge[y] & 2) && x >= b - edge_half) xs = b - edge_half - 1;
        double acc = 0.0;
        for (int d = 0; d <= 2 * half_width; ++d) {
            if (d == half_width) continue;
            acc += hf[shg::reflect101(y - half_width + d, h) * w + xs];
        }
        const double r3 = acc * (1.0 / (double)(2 * half_width * linlen));
        const double r4 = hl[y * w + xs] * (1.0 / (double)linlen);
        if (F64) {
            delta = r4 - r3;
        } else {
            const float d32 = (float)r4 - (float)r3;
            delta = (double)d32;
        }
    }
    double out = v * exp(-delta * taper[y]);
    out = out > 65535.0 ? 65535.0 : out;           // np.minimum(., 65535); NaN converts to 0 like the x86 cast
    dst[y * dst_pitch + x] = (uint16_t)(int)out;
}

}  // namespace

extern "C" int shg_lin_filter_row_sums(const uint16_t* img, int64_t h, int64_t w, int64_t pitch, const double* row_factor,
                                       const float* log_lut, const uint8_t* flagged, const int32_t* up, const int32_t* dn,
                                       int linlen, double* hl, double* hf, shg_stream_t stream) {
    SHG_REQUIRE(img && flagged && up && dn && hl && hf, SHG_E_ARG, "shg_lin_filter_row_sums: null pointer");
    SHG_REQUIRE(row_factor || log_lut, SHG_E_ARG, "shg_lin_filter_row_sums: a uint16 image needs the float32 log table");
    SHG_REQUIRE(h > 0 && w > 0 && pitch >= w && h < 65536, SHG_E_ARG, "shg_lin_filter_row_sums: bad image size");
    SHG_REQUIRE(linlen >= 1 && (linlen & 1) && linlen <= MAXLIN, SHG_E_UNSUPPORTED,
                "shg_lin_filter_row_sums: linlen must be odd and <= %d", MAXLIN);
    dim3 grid((unsigned)((w + SEG - 1) / SEG), (unsigned)h);
    hipStream_t st = shg::as_stream(stream);
    SHG_PROF("lin_row_sums", st);
    if (row_factor) k_lin_row_sums<true><<<grid, 256, 0, st>>>(img, w, pitch, row_factor, log_lut, flagged, up, dn, linlen, hl, hf);
    else k_lin_row_sums<false><<<grid, 256, 0, st>>>(img, w, pitch, row_factor, log_lut, flagged, up, dn, linlen, hl, hf);
    return shg::check_launch("k_lin_row_sums");
}

extern "C" int shg_lin_filter_apply(const uint16_t* img, int64_t h, int64_t w, int64_t pitch, const double* row_factor,
                                    const double* hl, const double* hf, int linlen, int half_width,
                                    const double* taper, const int32_t* xa, const int32_t* xb, const uint8_t* edge, int edge_half,
                                    uint16_t* dst, int64_t dst_pitch, shg_stream_t stream) {
    SHG_REQUIRE(img && hl && hf && taper && xa && xb && edge && dst, SHG_E_ARG, "shg_lin_filter_apply: null pointer");
    SHG_REQUIRE(h > 0 && w > 0 && pitch >= w && dst_pitch >= w && h < 65536, SHG_E_ARG, "shg_lin_filter_apply: bad image size");
    SHG_REQUIRE(linlen >= 1 && half_width >= 1 && edge_half >= 0, SHG_E_ARG, "shg_lin_filter_apply: bad window");
    dim3 grid((unsigned)((w + 255) / 256), (unsigned)h);
    hipStream_t st = shg::as_stream(stream);
    SHG_PROF("lin_apply", st);
    if (row_factor)
        k_lin_apply<true><<<grid, 256, 0, st>>>(img, h, w, pitch, row_factor, hl, hf, linlen, half_width, taper, xa, xb, edge, edge_half, dst, dst_pitch);
    else
        k_lin_apply<false><<<grid, 256, 0, st>>>(img, h, w, pitch, row_factor, hl, hf, linlen, half_width, taper, xa, xb, edge, edge_half, dst, dst_pitch);
    return shg::check_launch("k_lin_apply");
}
