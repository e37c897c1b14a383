// The ellipse -> circle warp.  Replaces skimage.transform.warp in correct_image
// (reference ellipse_to_circle.py:112-118).  The correction matrix is upper
// triangular with [1][1] = 1 (ellipse_to_circle.py:48-49), so every output row is a
// 1-D linear resample of the same input row: 2 reads + 1 write per output pixel,
// HBM/L2-bound.  Arithmetic follows scikit-image 0.18.3 (_warp_fast,
// bilinear_interpolation, _clip_warp_output) in float64, unfused.
#include <algorithm>
#include <cmath>
#include "shg_common.h"
#include <type_traits>

namespace {

__global__ void k_minmax_init(uint32_t* mm) {
    mm[0] = 0xffffffffu;
    mm[1] = 0u;
}

__global__ __launch_bounds__(256) void k_minmax(const uint16_t* __restrict__ src, int64_t h, int64_t w, int64_t pitch,
                                                uint32_t* __restrict__ mm) {
    uint32_t lo = 0xffffffffu, hi = 0u;
    // a workgroup owns rows blockIdx.x, + gridDim.x, ...; eight of them at a time, so that every lane has eight
    // independent loads in flight (clamped row instead of a predicate); no per-pixel division
    for (int64_t y0 = blockIdx.x; y0 < h; y0 += 8 * (int64_t)gridDim.x) {
        for (int64_t x = threadIdx.x; x < w; x += 256) {
            uint32_t v[8];
            bool ok[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t y = y0 + u * (int64_t)gridDim.x;
                ok[u] = y < h;
                v[u] = src[(ok[u] ? y : y0) * pitch + x];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (ok[u]) {
                    lo = v[u] < lo ? v[u] : lo;
                    hi = v[u] > hi ? v[u] : hi;
                }
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t ol = __shfl_xor(lo, d), oh = __shfl_xor(hi, d);
        lo = ol < lo ? ol : lo;
        hi = oh > hi ? oh : hi;
    }
    // one atomic pair per workgroup: same-address atomics serialise chip-wide
    __shared__ uint32_t wlo[4], whi[4];
    if ((threadIdx.x & 63) == 0) { wlo[threadIdx.x >> 6] = lo; whi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; ++i) { lo = wlo[i] < lo ? wlo[i] : lo; hi = whi[i] > hi ? whi[i] : hi; }
        atomicMin(&mm[0], lo);
        atomicMax(&mm[1], hi);
    }
}

constexpr int kWarpBatch = 24;                     // disks per launch (their transform rows travel by value; a 21-disk stack is one launch)
struct WarpRows { double h[kWarpBatch][3]; };
// Output rows per lane (their 2 x WARP_ROWS loads are issued before the first use).  Every wave pays for its set-up (kernel arguments,
// the transform row, the extrema) once, so more rows per lane is less work in total -- as long as there are still enough workgroups
// to fill 256 CUs several times over: 4 rows for a pair of disks, 8 for a Doppler stack (profiles/r05_sweeps.txt).

// grid (x, ceil(rows / WARP_ROWS), disks): blockIdx.z picks source, destination, transform row and extrema
using WarpPtrs = shg::PtrBatchN<kWarpBatch>;
struct WarpArgs {
    WarpPtrs srcs;
    int64_t h, w, pitch;
    WarpRows rows;
    WarpPtrs dsts;
    int64_t out_h, out_w, dst_pitch;
    WarpPtrs mms;
    uint32_t nv_magic, nv_shift;                   // k_warp_rows8: flat / nv = __umulhi(flat, nv_magic) >> nv_shift
};

template <int WARP_ROWS>
__global__ __launch_bounds__(256) void k_warp_rows(const WarpArgs kargs) {
    const WarpPtrs& srcs = kargs.srcs;
    const WarpPtrs& dsts = kargs.dsts;
    const WarpPtrs& mms = kargs.mms;
    const int64_t h = kargs.h, w = kargs.w, pitch = kargs.pitch, out_h = kargs.out_h, out_w = kargs.out_w, dst_pitch = kargs.dst_pitch;
    const WarpRows& rows = kargs.rows;
    const uint16_t* __restrict__ src = srcs.at<const uint16_t>(blockIdx.z);
    uint16_t* __restrict__ dst = dsts.at<uint16_t>(blockIdx.z);
    const uint32_t* __restrict__ mm = mms.at<const uint32_t>(blockIdx.z);
    const double h00 = rows.h[blockIdx.z][0], h01 = rows.h[blockIdx.z][1], h02 = rows.h[blockIdx.z][2];
    const int c = (int)(blockIdx.x * 256 + threadIdx.x);
    const int ra = (int)blockIdx.y * WARP_ROWS;
    if (c >= out_w) return;
    // The reference works on img / 65536 and stores (2**16 * warped).astype(uint16).  Scaling by a power of two is exact and
    // commutes with every rounding below, so the blend, the clip and the truncation run on the raw sample values: the same
    // results with three float64 multiplications less per pixel (this kernel is bound by its float64 instructions).
    // Both samples of every row are read unconditionally from clamped positions and replaced by cval afterwards where they
    // fall outside the image: no branch (and no wait) between the loads.
    // The kernel is bound by its instruction count (tools/probes/valu_rates.hip: a float64 compare costs 8.9 cycles a wave, a
    // conversion 7.9, an add 5.6), so what can be decided on integers is: x1 = ceil(x) is x0 + 1 unless x is whole, and "0 <= x0 < w"
    // is one unsigned compare of the converted x0 (the conversion saturates, so far-away positions stay outside; a transform that
    // is not finite never gets here: `sane`, uniform over the launch, keeps every sample outside as the float compares would).
    // A sample outside the image is replaced by cval = image[0, 0] BEFORE it is converted (one 32-bit select instead of two).
    double dc[WARP_ROWS];
    uint32_t v0[WARP_ROWS], v1[WARP_ROWS];
    bool in0[WARP_ROWS], in1[WARP_ROWS];
    const bool sane = fabs(h00) < 1e300 && fabs(h01) < 1e300 && fabs(h02) < 1e300;
    const uint32_t wu = (uint32_t)w;
    const double xc = h00 * (double)c;
#pragma unroll
    for (int rr = 0; rr < WARP_ROWS; ++rr) {
        const int r = ra + rr;
        const double x = xc + h01 * (double)r + h02;
        const double x0 = floor(x);
        dc[rr] = x - x0;
        const int i0 = __double2int_rz(x0);                       // exact for |x0| < 2^31, saturated beyond
        const int i1 = dc[rr] != 0.0 ? (i0 == 0x7fffffff ? i0 : i0 + 1) : i0;      // ceil(x)
        const bool row_ok = sane && r < h;                        // (r < out_h is checked at the store)
        in0[rr] = row_ok && (uint32_t)i0 < wu;
        in1[rr] = row_ok && (uint32_t)i1 < wu;
        const uint16_t* row = src + (int64_t)(row_ok ? r : 0) * pitch;
        v0[rr] = row[in0[rr] ? i0 : 0];
        v1[rr] = row[in1[rr] ? i1 : 0];
    }
    const uint32_t cval = src[0];                             // cval = image[0, 0]
    const double lo = (double)mm[0], hi = (double)mm[1];
#pragma unroll
    for (int rr = 0; rr < WARP_ROWS; ++rr) {
        const int r = ra + rr;
        if (r >= out_h) break;
        const double left = (double)(in0[rr] ? v0[rr] : cval), right = (double)(in1[rr] ? v1[rr] : cval);
        double v = (1.0 - dc[rr]) * left + dc[rr] * right;
        v = fmin(fmax(v, lo), hi);                                 // np.clip(warped, image.min(), image.max()): no NaN can get here
        dst[(int64_t)r * dst_pitch + c] = (uint16_t)(int)v;        // (2**16 * img).astype(uint16)
    }
}

// ---- eight output pixels a lane (round 6) ---------------------------------------------------------------------------------------
// k_warp_rows above issues two 2-byte loads and one 2-byte store per output pixel: 24 vector-memory instructions for a wave's 8 rows,
// each of them 128 bytes -- the kernel was bound by their issue, not by their bytes (0.37 of the HBM roofline at C4).  Here a lane owns 8
// consecutive columns of one output row: its 16 source samples lie within 24 pixels of the first one (the transform's column step is at
// most 1.75: the host checks), so it fetches three aligned pieces of the source row (16, 16 and 12 bytes), parks every PAIR of
// neighbours of the window in LDS ([pair][lane]: the bank is the lane, no conflict whatever pair a lane asks for), and reads each
// pixel's pair back with one ds_read_b32 at an offset of its own; one 16-byte store.  4 vector-memory instructions for 8 pixels a lane
// instead of 24.  Same float64 arithmetic, unfused -- but no more of it than the result needs: the column term as one fma, no test for
// whole positions, the clip on integers.
// The kernel's time follows the vector instructions it issues (57.8 M wave instructions a C4 launch, 42 a pixel: profiles/
// r06_sq_k_warp_rows8.json; 28 % fewer of them: 19 % less time), so the second half of round 6 counted them: waves whose samples all lie inside their rows skip the
// border tests (-6 a pixel), pairs instead of words in LDS (-3), flat / nv as a multiply-high (-2), 32-bit byte offsets off the
// scalar image bases (-2), the first pixel's position from the placement (-1): ~29 a pixel, 18 in the blend itself.
// C4: 115 (k_warp_rows) -> 104 -> 82 - 85 us (profiles/r06_sweeps.txt).
// Lanes are dealt (row, 8-column vector) pairs in one flat sequence, ITER of them per thread, the next one's pieces asked for before
// the current one is blended.
typedef unsigned int __attribute__((ext_vector_type(4))) u32x4_t;
typedef unsigned int __attribute__((ext_vector_type(3))) u32x3_t;
constexpr int kWarpPairs = 22;                     // LDS words per lane: the pairs (pixel p, pixel p + 1) of the window, p = -1 ... 20
template <int ITER>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7))) void k_warp_rows8(const WarpArgs kargs) {
    __shared__ uint32_t win[4][kWarpPairs][64];
    const int64_t h = kargs.h, w = kargs.w, pitch = kargs.pitch, out_h = kargs.out_h, out_w = kargs.out_w, dst_pitch = kargs.dst_pitch;
    const uint16_t* __restrict__ src = kargs.srcs.at<const uint16_t>(blockIdx.z);
    uint16_t* __restrict__ dst = kargs.dsts.at<uint16_t>(blockIdx.z);
    const uint32_t* __restrict__ mm = kargs.mms.at<const uint32_t>(blockIdx.z);
    const double h00 = kargs.rows.h[blockIdx.z][0], h01 = kargs.rows.h[blockIdx.z][1], h02 = kargs.rows.h[blockIdx.z][2];
    const uint32_t nv = (uint32_t)((out_w + 7) / 8), total = nv * (uint32_t)out_h;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t (*const my)[64] = win[wave];
    const uint32_t cval = src[0];                             // cval = image[0, 0]
    const int lo_i = (int)mm[0], hi_i = (int)mm[1];
    const double k00 = h00 * 0x1p+52;
    const uint32_t wu = (uint32_t)w;
    const int w_last = (int)w - 1, last_piece = (int)pitch - 8;

    // (row, first column, window start) of the it-th vector of this thread, and its three pieces
    auto place = [&](int it, int& r, int& c, int& ws, double& hr, double& x) {
        const uint32_t flat = ((uint32_t)blockIdx.x * ITER + (uint32_t)it) * 256u + threadIdx.x;
        const uint32_t f = flat < total ? flat : total - 1;
        r = (int)(__umulhi(f, kargs.nv_magic) >> kargs.nv_shift);      // f / nv (exact below 2^31: the host's choice of the pair)
        c = (int)(f - (uint32_t)r * nv) * 8;
        hr = h01 * (double)r;
        x = h00 * (double)c + hr + h02;                         // (the first pixel's position: the blend below takes it from here)
        const int i0 = __double2int_rz(floor(x));               // saturates far outside
        ws = min(max(i0, 0), w_last) & ~7;
        return flat < total;
    };
    // (the window's last two pixels are nobody's neighbours -- see the pairs below --: the third piece is 12 bytes.  A 16-byte load
    // whose last word is never read lets the compiler reuse that register straight away, and wait for the whole prefetch to land first.)
    // Byte offsets in 32 bits off the images' (scalar) bases -- the host checks that both images end below 4 GiB and that the pitches
    // fit 24 bits --: one multiply and an add-and-shift per piece where 64-bit pointers took a dozen instructions a vector.
    const char* const src_bytes = reinterpret_cast<const char*>(src);
    const uint32_t pitch_u = (uint32_t)pitch, dst_pitch_u = (uint32_t)dst_pitch, hu = (uint32_t)h;
    auto fetch = [&](int r, int ws, u32x4_t (&q)[2], u32x3_t& q2) {
        const uint32_t row = __umul24((uint32_t)r < hu ? (uint32_t)r : 0u, pitch_u);
#pragma unroll
        for (int t = 0; t < 2; ++t) q[t] = *reinterpret_cast<const u32x4_t*>(src_bytes + ((row + (uint32_t)min(ws + 8 * t, last_piece)) << 1));   // (a piece past the row: all of it outside the image)
        q2 = *reinterpret_cast<const u32x3_t*>(src_bytes + ((row + (uint32_t)min(ws + 16, last_piece)) << 1));
    };
    int r, c, ws;
    double hr_next, x_next;
    bool live = place(0, r, c, ws, hr_next, x_next);
    u32x4_t q[2];
    u32x3_t q2;
    fetch(r, ws, q, q2);
#pragma unroll 1
    for (int it = 0; it < ITER; ++it) {
        if (!__any(live)) break;
        {
            // every pair of neighbours a pixel of this vector can ask for, one word each: pair p = (pixel p, pixel p + 1) of the window
            // at row p + 1.  Even p: a word of the pieces as it is; odd p: the funnel shift of two (11 instructions a vector, where
            // every PIXEL spent four on picking its pair out of two words).  p = -1 serves a position just left of the image; the
            // window's last three pixels are never a left neighbour (a00 <= 1.75: p <= 7 + 13).
            const uint32_t wd[11] = {q[0].x, q[0].y, q[0].z, q[0].w, q[1].x, q[1].y, q[1].z, q[1].w, q2.x, q2.y, q2.z};
            my[0][lane] = wd[0] << 16;
#pragma unroll
            for (int k = 0; k < 11; ++k) my[1 + 2 * k][lane] = wd[k];
#pragma unroll
            for (int k = 0; k < 10; ++k) my[2 + 2 * k][lane] = __builtin_amdgcn_alignbit(wd[k + 1], wd[k], 16);
        }
        const int r_now = r, c_now = c, ws_now = ws;
        const double hr = hr_next, x_first = x_next;
        const bool live_now = live;
        if (it + 1 < ITER) {
            live = place(it + 1, r, c, ws, hr_next, x_next);
            fetch(r, ws, q, q2);
        }
        const uint32_t w_row = r_now < h ? wu : 0u;              // (a row the source does not have: every sample outside)
        const int ws1 = ws_now - 1;
        const double col0 = __hiloint2double(0x43300000, c_now);    // 2^52 + c: + j is exact
        uint32_t out[8];
        // INSIDE: every sample of the wave's 64 vectors lies in its row (all but the waves over the image's left and right edges):
        // no tests against the row's width, no cval, no clamp of the window offset -- 6 of a pixel's ~36 vector instructions, and
        // the kernel's time follows those (see above).
        auto pixels = [&](auto inside_tag) {
            constexpr bool INSIDE = decltype(inside_tag)::value;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                // h00 * (double)(c + j) with one instruction: 2^52 + (c + j) is the double whose low word is c + j, and (2^52 + n) h00 - 2^52 h00
                // rounds once, to fl(n h00) (the column step is at most 1.75: 2^52 h00 is exact)
                const double x = j == 0 ? x_first : __builtin_fma(col0 + (double)j, h00, -k00) + hr + h02;
                const double x0 = floor(x);
                const double dc = x - x0;
                const int i0 = __double2int_rz(x0);                  // saturates far outside: stays outside
                // the pair (pixel i0, pixel i0 + 1) out of the window (a position outside the window is outside the image: whatever it
                // reads is replaced by cval)
                const int row = INSIDE ? i0 - ws1 : min(max(i0 - ws1, 0), kWarpPairs - 1);
                const uint32_t pair = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(&my[0][lane]) + ((uint32_t)row << 8));
                double left, right;
                if constexpr (INSIDE) {
                    left = (double)(pair & 0xffffu);
                    right = (double)(pair >> 16);
                } else {
                    // the right neighbour is pixel i0 + 1 even where x is whole (the reference takes ceil(x) = i0 there): its weight dc is 0
                    // then, and 0 x any sample is +0 -- no test for it
                    const bool in0 = (uint32_t)i0 < w_row, in1 = (uint32_t)i0 + 1u < w_row;
                    left = (double)(in0 ? (pair & 0xffffu) : cval);
                    right = (double)(in1 ? pair >> 16 : cval);
                }
                const double v = (1.0 - dc) * left + dc * right;
                // np.clip(warped, image.min(), image.max()) and the truncation commute (both monotonic, the bounds whole numbers): on integers
                out[j] = (uint32_t)min(max((int)v, lo_i), hi_i);
            }
        };
        // (a00 >= 0: the first and the last pixel's positions bound the others'; the window starts at most 7 pixels before the first)
        const double x_last = __builtin_fma(col0 + 7.0, h00, -k00) + hr + h02;
        const bool inside = x_first >= 0.0 && x_last < (double)w_last && r_now < h;
        if (__all(inside)) pixels(std::true_type{});
        else pixels(std::false_type{});
        if (live_now) {
            uint16_t* o = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(dst) + ((__umul24((uint32_t)r_now, dst_pitch_u) + (uint32_t)c_now) << 1));
            if (c_now + 8 <= out_w) {
                const u32x4_t pk = {out[0] | (out[1] << 16), out[2] | (out[3] << 16), out[4] | (out[5] << 16), out[6] | (out[7] << 16)};
                *reinterpret_cast<u32x4_t*>(o) = pk;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (c_now + j < out_w) o[j] = (uint16_t)out[j];
            }
        }
    }
}

constexpr int64_t kWarpWideAbove = 16384;          // workgroups an 8-rows-a-lane launch must still have
int warp_rows_forced() {                           // SHG_WARP_ROWS=4|8|16: the A / B switch of the sweep
    static const int v = [] { const char* e = getenv("SHG_WARP_ROWS"); const int n = e ? atoi(e) : 0; return n == 4 || n == 8 || n == 16 ? n : 0; }();
    return v;
}

}  // namespace

extern "C" int shg_warp_rows_u16(const uint16_t* src, int64_t h, int64_t w, int64_t src_pitch, double h00, double h01,
                                 double h02, uint16_t* dst, int64_t out_h, int64_t out_w, int64_t dst_pitch,
                                 uint32_t* minmax, shg_stream_t stream) {
    SHG_REQUIRE(src && dst && minmax, SHG_E_ARG, "shg_warp_rows_u16: null pointer");
    SHG_REQUIRE(h > 0 && w > 0 && out_h > 0 && out_w > 0, SHG_E_ARG, "shg_warp_rows_u16: empty image");
    SHG_REQUIRE(src_pitch >= w && dst_pitch >= out_w, SHG_E_ARG, "shg_warp_rows_u16: pitch smaller than width");
    SHG_REQUIRE(out_h < 65536, SHG_E_UNSUPPORTED, "shg_warp_rows_u16: more than 65535 rows");
    hipStream_t st = shg::as_stream(stream);
    { SHG_PROF("minmax", st); k_minmax_init<<<1, 1, 0, st>>>(minmax); }
    int64_t blocks = h < 256 ? h : 256;
    { SHG_PROF("minmax", st); k_minmax<<<(unsigned)blocks, 256, 0, st>>>(src, h, w, src_pitch, minmax); }
    if (int e = shg::check_launch("k_minmax")) return e;
    const double h3[3] = {h00, h01, h02};
    const uint32_t* mm = minmax;
    return shg::warp_rows_batch(&src, 1, h, w, src_pitch, h3, &dst, out_h, out_w, dst_pitch, &mm, stream);
}

extern "C" int shg_warp_rows_minmax_u16(const uint16_t* src, int64_t h, int64_t w, int64_t src_pitch, double h00, double h01,
                                        double h02, uint16_t* dst, int64_t out_h, int64_t out_w, int64_t dst_pitch,
                                        const uint32_t* minmax2, shg_stream_t stream) {
    SHG_REQUIRE(src && dst && minmax2, SHG_E_ARG, "shg_warp_rows_minmax_u16: null pointer");
    const double h3[3] = {h00, h01, h02};
    return shg::warp_rows_batch(&src, 1, h, w, src_pitch, h3, &dst, out_h, out_w, dst_pitch, &minmax2, stream);
}

// k disks of one shape (a Doppler stack) in one launch: host_h3 is [k][3], host_minmax2[i] -> {min, max} of disk i on the device
int shg::warp_rows_batch(const uint16_t* const* host_srcs, int64_t k, int64_t h, int64_t w, int64_t src_pitch, const double* host_h3,
                         uint16_t* const* host_dsts, int64_t out_h, int64_t out_w, int64_t dst_pitch, const uint32_t* const* host_minmax2,
                         shg_stream_t stream) {
    SHG_REQUIRE(host_srcs && host_h3 && host_dsts && host_minmax2 && k > 0, SHG_E_ARG, "shg_warp_rows_minmax_u16: null pointer");
    SHG_REQUIRE(h > 0 && w > 0 && out_h > 0 && out_w > 0, SHG_E_ARG, "shg_warp_rows_minmax_u16: empty image");
    SHG_REQUIRE(src_pitch >= w && dst_pitch >= out_w, SHG_E_ARG, "shg_warp_rows_minmax_u16: pitch smaller than width");
    SHG_REQUIRE(out_h < 65536 && out_w < (1ll << 30) && w < (1ll << 30) && h < (1ll << 30), SHG_E_UNSUPPORTED, "shg_warp_rows_minmax_u16: image too large");
    for (int64_t i = 0; i < k; ++i) SHG_REQUIRE(host_srcs[i] && host_dsts[i] && host_minmax2[i], SHG_E_ARG, "shg_warp_rows_minmax_u16: null image");
    hipStream_t st = shg::as_stream(stream);
    SHG_PROF("warp", st);
    for (int64_t i0 = 0; i0 < k; i0 += kWarpBatch) {
        const int m = (int)std::min<int64_t>(kWarpBatch, k - i0);
        WarpRows rows = {};
        for (int d = 0; d < m; ++d)
            for (int j = 0; j < 3; ++j) rows.h[d][j] = host_h3[3 * (i0 + d) + j];
        // vectors a row, as a multiply-high and a shift: with s = ceil(log2 nv) - 1 and M = ceil(2^(32 + s) / nv) <= 2^32 - 1 (nv >= 2),
        // M nv - 2^(32 + s) < nv, so f M / 2^(32 + s) stays below the next whole number for every f < 2^(32 + s) / nv, which is > 2^31
        const uint64_t nv = (uint64_t)((out_w + 7) / 8);
        uint32_t nv_shift = 0;
        while ((2ull << nv_shift) < nv) ++nv_shift;
        const uint32_t nv_magic = nv >= 2 ? (uint32_t)(((1ull << (32 + nv_shift)) + nv - 1) / nv) : 0u;
        const WarpArgs args{shg::make_batch_n<kWarpBatch>(host_srcs, (int)i0, m), h, w, src_pitch, rows, shg::make_batch_n<kWarpBatch>(host_dsts, (int)i0, m),
                            out_h, out_w, dst_pitch, shg::make_batch_n<kWarpBatch>(host_minmax2, (int)i0, m), nv_magic, nv_shift};
        // eight pixels a lane where the pieces can be 16-byte loads and stores and the window holds a lane's samples (SHG_WARP_WIDE=0: never)
        const bool wide_ok = [] { const char* e = getenv("SHG_WARP_WIDE"); return !e || atoi(e) != 0; }();     // (read per call: the parity test switches it)
        bool wide = wide_ok && src_pitch % 8 == 0 && dst_pitch % 8 == 0 && w >= 8 && src_pitch >= 8 && nv >= 2 && out_h * ((out_w + 7) / 8) < (1ll << 31) &&
                    src_pitch < (1ll << 24) && dst_pitch < (1ll << 24) && h * src_pitch < (1ll << 31) && out_h * dst_pitch < (1ll << 31);
        for (int d = 0; d < m && wide; ++d) {
            const double a00 = rows.h[d][0], a01 = rows.h[d][1], a02 = rows.h[d][2];
            wide = a00 >= 0.0 && a00 <= 1.75 && std::fabs(a01) < 1e300 && std::fabs(a02) < 1e300 &&
                   (reinterpret_cast<uintptr_t>(host_srcs[i0 + d]) & 15) == 0 && (reinterpret_cast<uintptr_t>(host_dsts[i0 + d]) & 15) == 0;
        }
        if (wide) {
            // ITER vectors a thread (the next one's pieces asked for while the current one is blended) -- as long as that still
            // leaves every CU a few workgroups: a single 2000 x 2096 disk is 512 workgroups of 4 vectors a thread
            const int64_t vecs = out_h * ((out_w + 7) / 8);
            // (measured on one box, C2 / C4: 1 vector a thread 11.3 - 12.9 / 82.3 us, 4 vectors 11.3 - 12.1 / 82.2 - 83.3: the resident waves hide
            // the loads either way; profiles/r06_sweeps.txt)
            const bool few = vecs * m < 4 * 256 * 2048;
            auto go8 = [&](auto I) {
                dim3 grid((unsigned)((vecs + 256 * I.value - 1) / (256 * I.value)), 1u, (unsigned)m);
                return shg::launch(k_warp_rows8<I.value>, grid, dim3(256), 0, st, args, "k_warp_rows8");
            };
            if (int e = few ? go8(std::integral_constant<int, 1>{}) : go8(std::integral_constant<int, 4>{})) return e;
            continue;
        }
        const int64_t gx = (out_w + 255) / 256;
        int rows_per_lane = gx * ((out_h + 7) / 8) * m >= kWarpWideAbove ? 8 : 4;
        if (warp_rows_forced()) rows_per_lane = warp_rows_forced();
        auto go = [&](auto R) {
            dim3 grid((unsigned)gx, (unsigned)((out_h + R.value - 1) / R.value), (unsigned)m);
            return shg::launch(k_warp_rows<R.value>, grid, dim3(256), 0, st, args, "k_warp_rows");
        };
        const int e = rows_per_lane == 16 ? go(std::integral_constant<int, 16>{})
                    : rows_per_lane == 8 ? go(std::integral_constant<int, 8>{}) : go(std::integral_constant<int, 4>{});
        if (e) return e;
    }
    return 0;
}
