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
#include <stdlib.h>
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

// ---- exact order statistics of a float64 array: MSB-first radix select, 8 bits per launch -------------
// np.median / np.percentile (ellipse_to_circle.py:165, 241) need the k-th smallest values exactly.
__device__ __forceinline__ uint64_t f64_key(double v) {          // monotone map double -> uint64
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_f64(uint64_t k) {
    const uint64_t b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

struct SelectState { uint64_t prefix; int64_t rank; };

// Replay the digit choices of passes 0 .. pass-1 from their global histograms, with the whole
// 256-thread workgroup: one bin per thread, a workgroup-wide inclusive scan, the bin whose
// cumulative range holds the rank wins.  Every workgroup replays the same (deterministic) choices.
__device__ __forceinline__ SelectState select_replay(const uint32_t* __restrict__ hist, int pass, int64_t rank) {
    __shared__ int64_t wave_tot[4];
    __shared__ int64_t chosen[2];          // digit, count below it
    SelectState st{0, rank};
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int p = 0; p < pass; ++p) {
        const int64_t c = hist[p * 256 + tid];
        int64_t incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int64_t o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        for (int i = 0; i < wave; ++i) incl += wave_tot[i];
        const int64_t excl = incl - c;
        if (excl <= st.rank && st.rank < incl) { chosen[0] = tid; chosen[1] = excl; }
        __syncthreads();
        st.rank -= chosen[1];
        st.prefix = (st.prefix << 8) | (uint64_t)chosen[0];
        __syncthreads();
    }
    return st;
}

// grid (blocks, n_ranks).  hist: [n_ranks][8][256] u32, zeroed.
__global__ __launch_bounds__(256) void k_select_pass(const double* const* __restrict__ arrays, int64_t n, int pass,
                                                     const int64_t* __restrict__ ranks, uint32_t* __restrict__ hist) {
    __shared__ uint32_t lh[256];
    const double* __restrict__ v = arrays[blockIdx.y];
    uint32_t* myhist = hist + (int64_t)blockIdx.y * 8 * 256;
    const SelectState st = select_replay(myhist, pass, ranks[blockIdx.y]);
    lh[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t prefix = st.prefix;
    const int shift = 56 - 8 * pass;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const uint64_t k = f64_key(v[i]);
        if (pass == 0 || (k >> (shift + 8)) == prefix) atomicAdd(&lh[(k >> shift) & 0xff], 1u);
    }
    __syncthreads();
    if (lh[threadIdx.x]) atomicAdd(&myhist[pass * 256 + threadIdx.x], lh[threadIdx.x]);
}

// grid (n_ranks), 256 threads
__global__ __launch_bounds__(256) void k_select_final(const int64_t* __restrict__ ranks, const uint32_t* __restrict__ hist,
                                                      double* __restrict__ out) {
    const SelectState st = select_replay(hist + (int64_t)blockIdx.x * 8 * 256, 8, ranks[blockIdx.x]);
    if (threadIdx.x == 0) out[blockIdx.x] = key_f64(st.prefix);
}

// ---- order statistics of a box-blurred block-mean image through its integer window sums -------------------------
// The 4x4 block mean of uint16 / 65536 is a multiple of 2^-20 below 1, so a k x k window sum is an integer number of
// 2^-20 units below k*k * 2^20 (< 2^32 for k <= 63) and cv2.blur's value is that sum times 1/(k*k): monotone in the
// sum.  np.median / np.percentile of the blurred image are therefore order statistics of 32-bit integers -- three
// 11-bit radix passes instead of the eight 8-bit passes the float64 keys need -- and the selected value is rebuilt
// with the very multiply the blur kernel performs.
__global__ __launch_bounds__(256) void k_boxf_cols_key(const double* __restrict__ tmp, int h, int w, int k, double scale,
                                                       double* __restrict__ dst, uint32_t* __restrict__ keys) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= h * w) return;
    const int y = i / w, x = i - y * w;
    const int ya = y - k / 2;
    double s = 0.0;
    for (int j = 0; j < k; ++j) s += tmp[(int64_t)shg::reflect101(ya + j, h) * w + x];
    dst[i] = s * scale;
    keys[i] = (uint32_t)(s * 1048576.0);                  // exact: s is a whole number of 2^-20 units
}

constexpr int SEL32_BITS = 11, SEL32_BINS = 1 << SEL32_BITS, SEL32_PASSES = 3;      // 33 bits >= 32

struct Pairs8 { const uint32_t* keys[8]; int64_t rank[8]; double scale[8]; };

// the bin of hist[SEL32_BINS] whose cumulative range holds `rank` (workgroup of 256: eight bins per thread)
__device__ __forceinline__ void pick_bin(const uint32_t* __restrict__ hist, int64_t rank, int& digit, int64_t& below) {
    __shared__ int64_t wave_tot[4];
    __shared__ int64_t chosen[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int PER = SEL32_BINS / 256;
    int64_t c[PER], local = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) { c[j] = hist[tid * PER + j]; local += c[j]; }
    int64_t incl = local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int64_t o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    for (int i = 0; i < wave; ++i) incl += wave_tot[i];
    int64_t excl = incl - local;
    if (excl <= rank && rank < incl) {
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            if (rank < excl + c[j]) { chosen[0] = tid * PER + j; chosen[1] = excl; break; }
            excl += c[j];
        }
    }
    __syncthreads();
    digit = (int)chosen[0];
    below = chosen[1];
    __syncthreads();
}

// grid (blocks, n_pairs), 256 threads.  hist: [n_pairs][SEL32_PASSES][SEL32_BINS] u32, zeroed.
__global__ __launch_bounds__(256) void k_select32_pass(Pairs8 p, int64_t n, int pass, uint32_t* __restrict__ hist) {
    __shared__ uint32_t lh[SEL32_BINS];
    const uint32_t* __restrict__ v = p.keys[blockIdx.y];
    uint32_t* myhist = hist + (int64_t)blockIdx.y * SEL32_PASSES * SEL32_BINS;
    int64_t rank = p.rank[blockIdx.y];
    uint64_t prefix = 0;
    for (int q = 0; q < pass; ++q) {                       // replay the digits of the passes before
        int digit;
        int64_t below;
        pick_bin(myhist + q * SEL32_BINS, rank, digit, below);
        rank -= below;
        prefix = (prefix << SEL32_BITS) | (uint64_t)digit;
    }
    for (int i = threadIdx.x; i < SEL32_BINS; i += 256) lh[i] = 0;
    __syncthreads();
    const int shift = SEL32_BITS * (SEL32_PASSES - 1 - pass);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const uint64_t kk = v[i];
        if (pass == 0 || (kk >> (shift + SEL32_BITS)) == prefix) atomicAdd(&lh[(kk >> shift) & (SEL32_BINS - 1)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SEL32_BINS; i += 256)
        if (lh[i]) atomicAdd(&myhist[pass * SEL32_BINS + i], lh[i]);
}

// grid (n_pairs), 256 threads: out = (key * 2^-20) * scale, the blur kernel's own arithmetic
__global__ __launch_bounds__(256) void k_select32_final(Pairs8 p, const uint32_t* __restrict__ hist, double* __restrict__ out) {
    const uint32_t* myhist = hist + (int64_t)blockIdx.x * SEL32_PASSES * SEL32_BINS;
    int64_t rank = p.rank[blockIdx.x];
    uint64_t key = 0;
    for (int q = 0; q < SEL32_PASSES; ++q) {
        int digit;
        int64_t below;
        pick_bin(myhist + q * SEL32_BINS, rank, digit, below);
        rank -= below;
        key = (key << SEL32_BITS) | (uint64_t)digit;
    }
    if (threadIdx.x == 0) out[blockIdx.x] = ((double)key * 9.5367431640625e-07) * p.scale[blockIdx.x];
}

// ---- get_flood_image's statistics (ellipse_to_circle.py:159-169) ----------------------------------------
// stats[0] = sum(image) (every value is k / 2^20 and the total < 2^18: exact in any order),
// then over data = blurred[blurred < very_bright]: stats[1] = min, stats[2] = max, counts[20] = np.histogram(data, 20)
// accumulators, the 20 counters, and very_bright: either the caller's value or NumPy's _lerp of two order statistics
// that are still on the device (np.percentile(img_blurred, 99), ellipse_to_circle.py:165) -- no host round trip
// Workgroups add their partial results to one of FLOOD_SLOTS slots (blockIdx % FLOOD_SLOTS), the reader folds the slots:
// 128 workgroups on the same three addresses queue up in the memory-side atomic units for most of the kernel's 9 us.
constexpr int FLOOD_SLOTS = 9;                            // 4 + 3 * 9 = 31 u64 of the 256-byte workspace

__global__ void k_flood_init(unsigned long long* __restrict__ acc, uint32_t* __restrict__ counts, const double* __restrict__ order_stats,
                             double gamma, double very_bright) {
    if (threadIdx.x < 20) counts[threadIdx.x] = 0;
    if (threadIdx.x < FLOOD_SLOTS) {                      // per-slot sum, min key, max key (k_flood_minmax)
        acc[4 + 3 * threadIdx.x] = 0ull;
        acc[5 + 3 * threadIdx.x] = ~0ull;
        acc[6 + 3 * threadIdx.x] = 0ull;
    }
    if (threadIdx.x == 0) {
        acc[0] = 0ull;
        acc[1] = ~0ull;
        acc[2] = 0ull;
        if (order_stats) {
            const double a = order_stats[0], b = order_stats[1], diff = b - a;
            very_bright = gamma >= 0.5 ? b - diff * (1.0 - gamma) : a + diff * gamma;
        }
        acc[3] = (unsigned long long)__double_as_longlong(very_bright);
    }
}

__global__ __launch_bounds__(256) void k_flood_minmax(const double* __restrict__ image, const double* __restrict__ blurred, int64_t n,
                                                      unsigned long long* __restrict__ acc) {
    // acc[0] = sum as fixed point (units of 2^-20), acc[1] = min key, acc[2] = max key, acc[3] = very_bright (double bits, k_flood_init)
    const double very_bright = __longlong_as_double((long long)acc[3]);
    unsigned long long s = 0, lo = ~0ull, hi = 0ull;
    const int64_t stride = (int64_t)gridDim.x * 256;
    // four elements per trip, all eight loads issued before the first use (clamped index instead of a predicate)
    for (int64_t base = (int64_t)blockIdx.x * 256 + threadIdx.x; base < n; base += 4 * stride) {
        double im[4], bl[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = base + u * stride;
            ok[u] = i < n;
            im[u] = image[ok[u] ? i : 0];
            bl[u] = blurred[ok[u] ? i : 0];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!ok[u]) continue;
            s += (unsigned long long)(im[u] * 1048576.0);
            if (bl[u] < very_bright) {
                const uint64_t k = f64_key(bl[u]);
                lo = k < lo ? k : lo;
                hi = k > hi ? k : hi;
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        s += __shfl_xor(s, d);
        const unsigned long long ol = __shfl_xor(lo, d), oh = __shfl_xor(hi, d);
        lo = ol < lo ? ol : lo;
        hi = oh > hi ? oh : hi;
    }
    // one atomic triple per workgroup: same-address atomics serialise chip-wide
    __shared__ unsigned long long ws[4], wlo[4], whi[4];
    if ((threadIdx.x & 63) == 0) { ws[threadIdx.x >> 6] = s; wlo[threadIdx.x >> 6] = lo; whi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; ++i) { s += ws[i]; lo = wlo[i] < lo ? wlo[i] : lo; hi = whi[i] > hi ? whi[i] : hi; }
        unsigned long long* slot = acc + 4 + 3 * (blockIdx.x % FLOOD_SLOTS);
        atomicAdd(&slot[0], s);
        atomicMin(&slot[1], lo);
        atomicMax(&slot[2], hi);
    }
}

__global__ __launch_bounds__(256) void k_flood_hist(const double* __restrict__ blurred, int64_t n,
                                                    const unsigned long long* __restrict__ acc, double* __restrict__ stats,
                                                    uint32_t* __restrict__ counts) {
    const double very_bright = __longlong_as_double((long long)acc[3]);
    __shared__ double edges[21];
    __shared__ uint32_t lc[20];
    unsigned long long total = 0, klo = ~0ull, khi = 0ull;
    for (int k = 0; k < FLOOD_SLOTS; ++k) {
        total += acc[4 + 3 * k];
        klo = acc[5 + 3 * k] < klo ? acc[5 + 3 * k] : klo;
        khi = acc[6 + 3 * k] > khi ? acc[6 + 3 * k] : khi;
    }
    const double mn = key_f64(klo), mx = key_f64(khi);
    if (threadIdx.x < 21) {
        // np.histogram: first == last -> (first - 0.5, last + 0.5); bin_edges = np.linspace(first, last, 21)
        double first = mn, last = mx;
        if (first == last) { first = first - 0.5; last = last + 0.5; }
        const double step = (last - first) / 20.0;
        edges[threadIdx.x] = threadIdx.x == 20 ? last : (double)threadIdx.x * step + first;
    }
    if (threadIdx.x < 20) lc[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double b = blurred[i];
        if (!(b < very_bright)) continue;
        int bin = 0;                                  // largest bin with edges[bin] <= b; the last bin is closed
        for (int j = 1; j < 20; ++j) bin = (b >= edges[j]) ? j : bin;
        atomicAdd(&lc[bin], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 20 && lc[threadIdx.x]) atomicAdd(&counts[threadIdx.x], lc[threadIdx.x]);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        stats[0] = (double)total / 1048576.0;
        stats[1] = mn;
        stats[2] = mx;
    }
}

// ---- canny's hysteresis + the labelling of its result (ellipse_to_circle.py:245-252) ------------------------
// 8-connected components of low_mask by union-find on the pixel grid (each pixel links to its W, NW, N, NE
// neighbours; roots are the smallest linear index of a component, so sorting roots = scipy.ndimage.label's
// raster numbering).  Components holding a high_mask pixel survive (skimage's hysteresis); their pixels
// are emitted in raster order with their root.
__device__ __forceinline__ int ccl_find(const int* L, int x) {
    // agent-scope relaxed loads: parents are rewritten by other workgroups (other XCDs) during the merge,
    // and a CU's L1 / an XCD's L2 is not refreshed by them.  A stale parent would still be a valid older
    // ancestor (parents only ever decrease, and every link is validated by the atomicMin), but fresh reads
    // keep the chains short.
    int p = __hip_atomic_load(&L[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != x) { x = p; p = __hip_atomic_load(&L[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    return x;
}

__device__ __forceinline__ void ccl_union(int* L, int a, int b) {
    while (true) {
        a = ccl_find(L, a);
        b = ccl_find(L, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }      // link the larger root under the smaller
        const int old = atomicMin(&L[a], b);
        if (old == a) return;
        a = old;
    }
}

__global__ __launch_bounds__(256) void k_ccl_init(const uint8_t* __restrict__ low, int n, int* __restrict__ L, int* __restrict__ flag) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    L[i] = low[i] ? i : -1;
    flag[i] = 0;
}

__global__ __launch_bounds__(256) void k_ccl_merge(const uint8_t* __restrict__ low, int h, int w, int* __restrict__ L) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= h * w || !low[i]) return;
    const int y = i / w, x = i - y * w;
    if (x > 0 && low[i - 1]) ccl_union(L, i, i - 1);
    if (y > 0) {
        const int up = i - w;
        if (x > 0 && low[up - 1]) ccl_union(L, i, up - 1);
        if (low[up]) ccl_union(L, i, up);
        if (x < w - 1 && low[up + 1]) ccl_union(L, i, up + 1);
    }
}

__global__ __launch_bounds__(256) void k_ccl_flatten(const uint8_t* __restrict__ low, const uint8_t* __restrict__ high, int n,
                                                     int* __restrict__ L, int* __restrict__ flag) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !low[i]) return;
    const int r = ccl_find(L, i);
    L[i] = r;            // benign race: every writer stores a value on the path to the same root
    if (high[i]) flag[r] = 1;
}

// grid = h rows, 256 threads.  counts[y] = kept pixels of the row (pass 0) or emit them at offsets[y] (pass 1)
__global__ __launch_bounds__(256) void k_ccl_emit(const int* __restrict__ L, const int* __restrict__ flag, int w, int pass,
                                                  int* __restrict__ counts, const int* __restrict__ offsets,
                                                  int* __restrict__ out_idx, int* __restrict__ out_root) {
    __shared__ int wave_cnt[4];
    __shared__ int base;
    const int y = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) base = pass ? offsets[y] : 0;
    __syncthreads();
    for (int x0 = 0; x0 < w; x0 += 256) {
        const int x = x0 + threadIdx.x;
        int root = -1;
        if (x < w) {
            const int l = L[y * w + x];
            if (l >= 0) {
                const int r = L[l] == l ? l : ccl_find(L, l);
                if (flag[r]) root = r;
            }
        }
        const unsigned long long m = __ballot(root >= 0);
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int i = 0; i < wave; ++i) off += wave_cnt[i];
        if (pass && root >= 0) {
            const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
            out_idx[pos] = y * w + x;
            out_root[pos] = root;
        }
        __syncthreads();
        if (threadIdx.x == 0) base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
    if (!pass && threadIdx.x == 0) counts[y] = base;
}

// exclusive scan of the h row counts by one workgroup; total -> out_count[0]
__global__ __launch_bounds__(256) void k_ccl_scan(const int* __restrict__ counts, int h, int* __restrict__ offsets,
                                                  int* __restrict__ out_count) {
    __shared__ int wave_tot[4];
    __shared__ int carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int y0 = 0; y0 < h; y0 += 256) {
        const int y = y0 + threadIdx.x;
        const int c = y < h ? counts[y] : 0;
        int incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int off = carry;
        for (int i = 0; i < wave; ++i) off += wave_tot[i];
        if (y < h) offsets[y] = off + incl - c;
        __syncthreads();
        if (threadIdx.x == 0) carry += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) out_count[0] = carry;
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

extern "C" int shg_box_blur_key_f64(const double* src, int64_t h, int64_t w, int k, double* dst, uint32_t* keys, double* tmp,
                                    shg_stream_t stream) {
    SHG_REQUIRE(src && dst && keys && tmp, SHG_E_ARG, "shg_box_blur_key_f64: null pointer");
    SHG_REQUIRE(h > 0 && w > 0 && h * w < (1ll << 30), SHG_E_ARG, "shg_box_blur_key_f64: bad image size");
    SHG_REQUIRE(k > 0, SHG_E_ARG, "shg_box_blur_f64: kernel %d must be positive", k);
    SHG_REQUIRE(k <= 63, SHG_E_UNSUPPORTED, "shg_box_blur_key_f64: a %d x %d window sum does not fit 32 bits", k, k);
    hipStream_t st = shg::as_stream(stream);
    const unsigned blocks = (unsigned)((h * w + 255) / 256);
    SHG_PROF("box_blur_f64", st);
    k_boxf_rows<<<blocks, 256, 0, st>>>(src, (int)h, (int)w, k, tmp);
    if (int e = shg::check_launch("k_boxf_rows")) return e;
    k_boxf_cols_key<<<blocks, 256, 0, st>>>(tmp, (int)h, (int)w, k, 1.0 / ((double)k * (double)k), dst, keys);
    return shg::check_launch("k_boxf_cols_key");
}

extern "C" size_t shg_select_keys_workspace_bytes(int n_pairs) {
    if (n_pairs < 1 || n_pairs > 8) return 0;
    return (size_t)n_pairs * SEL32_PASSES * SEL32_BINS * sizeof(uint32_t);
}

extern "C" int shg_select_keys_u32(const uint32_t* const* host_keys, int64_t n, const int64_t* host_ranks, const int* host_k,
                                   int n_pairs, double* out, void* workspace, size_t workspace_bytes, shg_stream_t stream) {
    SHG_REQUIRE(host_keys && host_ranks && host_k && out && workspace, SHG_E_ARG, "shg_select_keys_u32: null pointer");
    SHG_REQUIRE(n > 0 && n_pairs >= 1 && n_pairs <= 8, SHG_E_ARG, "shg_select_keys_u32: bad sizes");
    SHG_REQUIRE(workspace_bytes >= shg_select_keys_workspace_bytes(n_pairs), SHG_E_WORKSPACE, "shg_select_keys_u32: workspace too small");
    Pairs8 p = {};
    for (int i = 0; i < n_pairs; ++i) {
        SHG_REQUIRE(host_keys[i], SHG_E_ARG, "shg_select_keys_u32: null array");
        SHG_REQUIRE(host_ranks[i] >= 0 && host_ranks[i] < n, SHG_E_ARG, "shg_select_keys_u32: rank %lld outside [0, %lld)", (long long)host_ranks[i], (long long)n);
        SHG_REQUIRE(host_k[i] > 0 && host_k[i] <= 63, SHG_E_ARG, "shg_select_keys_u32: window %d", host_k[i]);
        p.keys[i] = host_keys[i];
        p.rank[i] = host_ranks[i];
        p.scale[i] = 1.0 / ((double)host_k[i] * (double)host_k[i]);
    }
    hipStream_t st = shg::as_stream(stream);
    uint32_t* hist = static_cast<uint32_t*>(workspace);
    if (hipError_t e = hipMemsetAsync(hist, 0, shg_select_keys_workspace_bytes(n_pairs), st)) {
        shg::set_error("shg_select_keys_u32: %s", hipGetErrorString(e));
        return (int)e;
    }
    int64_t blocks = (n + 2047) / 2048;                  // (fewer, longer workgroups are slower: 64 / 32 / 16 -> 7.2 / 9.8 / 15.5 us per pass)
    if (blocks > 256) blocks = 256;
    SHG_PROF("select", st);
    for (int pass = 0; pass < SEL32_PASSES; ++pass) {
        k_select32_pass<<<dim3((unsigned)blocks, (unsigned)n_pairs), 256, 0, st>>>(p, n, pass, hist);
        if (int err = shg::check_launch("k_select32_pass")) return err;
    }
    k_select32_final<<<(unsigned)n_pairs, 256, 0, st>>>(p, hist, out);
    return shg::check_launch("k_select32_final");
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

extern "C" size_t shg_select_workspace_bytes(int n_ranks) {
    if (n_ranks < 1 || n_ranks > 8) return 0;
    return (size_t)n_ranks * (8 * 256 * sizeof(uint32_t) + sizeof(int64_t) + sizeof(void*));
}

extern "C" int shg_select_f64(const double* values, int64_t n, const int64_t* host_ranks, int n_ranks, double* out,
                              void* workspace, size_t workspace_bytes, shg_stream_t stream) {
    SHG_REQUIRE(values && n_ranks >= 1 && n_ranks <= 8, SHG_E_ARG, "shg_select_f64: bad arguments");
    const double* same[8];
    for (int i = 0; i < n_ranks; ++i) same[i] = values;
    return shg_select_multi_f64(same, n, host_ranks, n_ranks, out, workspace, workspace_bytes, stream);
}

extern "C" int shg_select_multi_f64(const double* const* host_arrays, int64_t n, const int64_t* host_ranks, int n_ranks, double* out,
                                    void* workspace, size_t workspace_bytes, shg_stream_t stream) {
    SHG_REQUIRE(host_arrays && host_ranks && out && workspace, SHG_E_ARG, "shg_select_f64: null pointer");
    SHG_REQUIRE(n > 0 && n_ranks >= 1 && n_ranks <= 8, SHG_E_ARG, "shg_select_f64: bad sizes");
    SHG_REQUIRE(workspace_bytes >= shg_select_workspace_bytes(n_ranks), SHG_E_WORKSPACE, "shg_select_f64: workspace too small");
    for (int i = 0; i < n_ranks; ++i)
        SHG_REQUIRE(host_ranks[i] >= 0 && host_ranks[i] < n, SHG_E_ARG, "shg_select_f64: rank %lld outside [0, %lld)",
                    (long long)host_ranks[i], (long long)n);
    hipStream_t st = shg::as_stream(stream);
    uint32_t* hist = static_cast<uint32_t*>(workspace);
    int64_t* ranks = reinterpret_cast<int64_t*>(hist + (size_t)n_ranks * 8 * 256);
    const double** arrays = reinterpret_cast<const double**>(ranks + n_ranks);
    for (int i = 0; i < n_ranks; ++i) SHG_REQUIRE(host_arrays[i], SHG_E_ARG, "shg_select_f64: null array");
    hipError_t e = hipMemsetAsync(hist, 0, (size_t)n_ranks * 8 * 256 * sizeof(uint32_t), st);
    if (e == hipSuccess) e = hipMemcpyAsync(ranks, host_ranks, n_ranks * sizeof(int64_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(arrays, host_arrays, n_ranks * sizeof(void*), hipMemcpyHostToDevice, st);
    if (e != hipSuccess) { shg::set_error("shg_select_f64: %s", hipGetErrorString(e)); return (int)e; }
    int64_t blocks = (n + 2047) / 2048;
    if (blocks > 256) blocks = 256;
    SHG_PROF("select", st);
    for (int pass = 0; pass < 8; ++pass) {
        k_select_pass<<<dim3((unsigned)blocks, (unsigned)n_ranks), 256, 0, st>>>(arrays, n, pass, ranks, hist);
        if (int err = shg::check_launch("k_select_pass")) return err;
    }
    k_select_final<<<(unsigned)n_ranks, 256, 0, st>>>(ranks, hist, out);
    return shg::check_launch("k_select_final");
}

static int flood_stats(const double* image, const double* blurred, int64_t n, const double* order_stats, double gamma,
                       double very_bright, double* stats, uint32_t* counts, void* workspace, shg_stream_t stream, const char* who) {
    SHG_REQUIRE(image && blurred && stats && counts && workspace, SHG_E_ARG, "%s: null pointer", who);
    SHG_REQUIRE(n > 0, SHG_E_ARG, "%s: empty image", who);
    hipStream_t st = shg::as_stream(stream);
    unsigned long long* acc = static_cast<unsigned long long*>(workspace);
    int64_t blocks = (n + 2047) / 2048;
    if (blocks > 256) blocks = 256;
    SHG_PROF("flood_stats", st);
    k_flood_init<<<1, 64, 0, st>>>(acc, counts, order_stats, gamma, very_bright);
    k_flood_minmax<<<(unsigned)blocks, 256, 0, st>>>(image, blurred, n, acc);
    if (int err = shg::check_launch("k_flood_minmax")) return err;
    k_flood_hist<<<(unsigned)blocks, 256, 0, st>>>(blurred, n, acc, stats, counts);
    return shg::check_launch("k_flood_hist");
}

extern "C" int shg_flood_stats_f64(const double* image, const double* blurred, int64_t n, double very_bright, double* stats,
                                   uint32_t* counts, void* workspace, shg_stream_t stream) {
    return flood_stats(image, blurred, n, nullptr, 0.0, very_bright, stats, counts, workspace, stream, "shg_flood_stats_f64");
}

extern "C" int shg_flood_stats_lerp_f64(const double* image, const double* blurred, int64_t n, const double* order_stats, double gamma,
                                        double* stats, uint32_t* counts, void* workspace, shg_stream_t stream) {
    SHG_REQUIRE(order_stats, SHG_E_ARG, "shg_flood_stats_lerp_f64: null pointer");
    SHG_REQUIRE(gamma >= 0.0 && gamma <= 1.0, SHG_E_ARG, "shg_flood_stats_lerp_f64: gamma %g outside [0, 1]", gamma);
    return flood_stats(image, blurred, n, order_stats, gamma, 0.0, stats, counts, workspace, stream, "shg_flood_stats_lerp_f64");
}

extern "C" size_t shg_edge_components_workspace_bytes(int64_t h, int64_t w) {
    if (h <= 0 || w <= 0) return 0;
    return ((size_t)2 * h * w + 2 * (size_t)h) * sizeof(int32_t);
}

extern "C" int shg_edge_components(const uint8_t* low_mask, const uint8_t* high_mask, int64_t h, int64_t w, int32_t* out_idx,
                                   int32_t* out_root, int32_t* out_count, void* workspace, size_t workspace_bytes,
                                   shg_stream_t stream) {
    SHG_REQUIRE(low_mask && high_mask && out_idx && out_root && out_count && workspace, SHG_E_ARG, "shg_edge_components: null pointer");
    SHG_REQUIRE(h > 0 && w > 0 && h * w < (1ll << 30), SHG_E_ARG, "shg_edge_components: bad image size");
    SHG_REQUIRE(workspace_bytes >= shg_edge_components_workspace_bytes(h, w), SHG_E_WORKSPACE, "shg_edge_components: workspace too small");
    hipStream_t st = shg::as_stream(stream);
    const int n = (int)(h * w);
    int* L = static_cast<int*>(workspace);
    int* flag = L + n;
    int* counts = flag + n;
    int* offsets = counts + h;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    SHG_PROF("edge_components", st);
    k_ccl_init<<<blocks, 256, 0, st>>>(low_mask, n, L, flag);
    k_ccl_merge<<<blocks, 256, 0, st>>>(low_mask, (int)h, (int)w, L);
    k_ccl_flatten<<<blocks, 256, 0, st>>>(low_mask, high_mask, n, L, flag);
    k_ccl_emit<<<(unsigned)h, 256, 0, st>>>(L, flag, (int)w, 0, counts, offsets, out_idx, out_root);
    k_ccl_scan<<<1, 256, 0, st>>>(counts, (int)h, offsets, out_count);
    k_ccl_emit<<<(unsigned)h, 256, 0, st>>>(L, flag, (int)w, 1, counts, offsets, out_idx, out_root);
    return shg::check_launch("k_ccl");
}
