// The limb stage in seven launches instead of twenty-three (ellipse_to_circle.py:148-291, 299-302).
//
// limb.hip holds one kernel per NumPy / OpenCV / scikit-image call of get_flood_image and get_edge_list: on the
// quarter-size image (250 k pixels at C2) every one of them is a 3-15 us launch, 119 us of a scan's 310 us chain.  Here the
// same arithmetic -- bit for bit: the tests hold each fused kernel against its limb.hip counterparts -- goes through LDS
// tiles:
//   shg_limb_prepare   (accumulators cleared by the extraction's last launch, or by k_zero_words) | block mean + cv2.blur (k and 5) +
//                      the image's smallest blurred value | first radix pass | second radix pass (its last workgroup forms the
//                      order statistics, very_bright and, from them, the flood's min / max) | flood histogram (its last
//                      workgroup stores the stage's 30 numbers where the host reads them)
//   shg_limb_edges     Gaussian (both axes) + Sobel + magnitude + non-maximum suppression + tile-local union-find |
//                      union across tile borders | raster-ordered emission straight into the host's staging area
// What makes the fusion exact rather than approximate:
//   * the 4x4 block mean of uint16 / 65536 is n * 2^-20 with n < 2^20 an integer, so every window sum cv2.blur forms is an
//     integer number of 2^-20 units below 2^53: the blur can be summed in any order, in integers, and the blurred value is
//     (W * 2^-20) * (1 / (k*k)) with one rounding -- the very product k_boxf_cols forms.  The blurred image is never
//     stored as float64: its consumers read the 32-bit window sums (`keys`) and form the value where they need it;
//   * the Gaussian of canny's all-ones mask is, after the first axis, a function of the row alone: one value per row instead
//     of a plane;
//   * SciPy's correlate1d order (centre tap first, then the symmetric pairs from the outside in) is kept inside the tile.
#include <math.h>
#include <stdlib.h>
#include <algorithm>
#include "shg_common.h"

namespace {

// "Everything this workgroup has added to the shared histograms has been performed": what a workgroup says before it counts
// itself done.  Its results are device-scope atomics (performed at the memory side, read back by the last workgroup with
// device-scope loads), so all that is needed is that every lane's atomics have been acknowledged (vmcnt) before lane 0 bumps the
// counter.  An acq_rel fetch_add at agent scope says the same and more: on this multi-XCD part it also writes back and
// invalidates the XCD's whole L2 (buffer_wbl2 / buffer_inv sc1), once per workgroup, under every kernel running beside it.
// THE HARDWARE ASSUMPTION, stated: on gfx950 an agent-scope atomic read-modify-write is carried out at the memory side (not in an
// XCD's L2) and its return / acknowledgement -- what vmcnt counts -- comes after it has been carried out; relaxed agent-scope
// atomic loads by another workgroup then see it.  That is this part's behaviour, not the HIP memory model's promise, so the file
// refuses to build for anything else (no second code path: this library is gfx950 only), and tests hold it:
// test_limb_stage_under_load_equals_the_separate_kernels (four streams x 2000 limb stages beside a looping pass A, fused ==
// separate bit for bit) and test_fused_limb_kernels_equal_the_separate_ones on images that span all eight XCDs.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "limb_fused.hip: published() relies on how gfx950 performs agent-scope atomics; use an acq_rel agent-scope fetch_add elsewhere"
#endif
__device__ __forceinline__ void published() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}
// "I am done": -> whether this workgroup is the last of `total`.  fence: the one-variable fallback for a box where the assumption above
// does not hold (SHG_LIMB_FENCE=1): the counter is bumped with an acq_rel agent-scope read-modify-write, which is what the memory
// model asks for (and costs the L2 write-back / invalidate the relaxed form avoids: +10 % on a C2 step, measured in round 4).
__device__ __forceinline__ bool count_done(uint32_t* done, uint32_t total, int fence) {
    const uint32_t before = fence ? __hip_atomic_fetch_add(done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT)
                                  : __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return before == total - 1;
}
inline int limb_fence() {                                    // (read at every call: a test holds one setting against the other in one process)
    const char* v = getenv("SHG_LIMB_FENCE");
    return v && v[0] == '1';
}


constexpr int TH = 16, TW = 64;              // output tile of both tiled kernels (1024 threads, a pixel each)
constexpr int MAXR = 16;
struct GaussW { double w[2 * MAXR + 1]; int radius; };
constexpr int FLOOD_SLOTS = 9;
constexpr int KMAX = 16;                     // largest cv2.blur window the fused path takes (int(0.01 * rows / 4): scans up to 6799 slit rows)
constexpr double kUnit = 9.5367431640625e-07;        // 2^-20

__device__ __forceinline__ uint64_t f64_key(double v) {          // monotone map double -> uint64
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_f64(uint64_t k) {
    const uint64_t b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

// ---- workspace layout of shg_limb_prepare (all 32-bit words unless noted) -------------------------------------------
struct PrepLayout {
    int bits0, bits1;                        // high / low digit of a window sum: two radix passes
    size_t hist0, hist1, coarse0, coarse1, acc, counts, done, zero_words;      // word offsets; hist0: [2 arrays][2^bits0], hist1: [4 pairs][2^bits1];
                                                                                // coarse0 / coarse1: [2][256] / [4][256], the same counts by their top 8 bits
    size_t keys5, keysk, total_words;
};

__host__ __device__ inline int bit_length(uint64_t v) {
    int b = 0;
    while (v) { ++b; v >>= 1; }
    return b;
}

PrepLayout prep_layout(int64_t sh, int64_t sw, int k) {
    PrepLayout L;
    const int kk = k > 5 ? k : 5;
    const int total_bits = bit_length((uint64_t)kk * kk * 16ull * 65535ull);       // a window sum is below this
    L.bits1 = total_bits / 2;
    L.bits0 = total_bits - L.bits1;
    size_t off = 0;
    L.hist0 = off; off += (size_t)2 << L.bits0;
    L.hist1 = off; off += (size_t)4 << L.bits1;
    L.coarse0 = off; off += 2 * 256;
    L.coarse1 = off; off += 4 * 256;
    L.acc = off; off += 2 * (4 + 3 * FLOOD_SLOTS + 4 + FLOOD_SLOTS);     // u64 each: very_bright, the slots, the four order statistics, then the
                                                                         // complemented keys of blur k's smallest values (k_limb_blur, one per slot)
    L.counts = off; off += 20;
    L.done = off; off += 4;                                     // [0]: workgroups of the second radix pass that are through, [1]: of the histogram
    L.zero_words = off;                                         // everything up to here is zeroed by the call
    off = (off + 63) / 64 * 64;
    const size_t n = (size_t)sh * (size_t)sw;
    L.keysk = off; off += (n + 63) / 64 * 64;
    L.keys5 = off; off += (n + 63) / 64 * 64;
    L.total_words = off;
    return L;
}

// A workgroup's LDS histogram lh[2^bits] out to the global fine histogram and to its 256-bin coarse companion (the same
// counts by their top 8 bits: what lets a reader find the bin of a rank with two coalesced loads instead of a walk over
// thousands of bins).  coarse_lds: 256 words of scratch.
__device__ __forceinline__ void flush_hist(const uint32_t* lh, int bits, uint32_t* coarse_lds, uint32_t* __restrict__ fine,
                                           uint32_t* __restrict__ coarse, int tid, int nthreads) {
    for (int e = tid; e < 256; e += nthreads) coarse_lds[e] = 0;
    __syncthreads();
    const int shift = bits - 8;
    for (int e = tid; e < (1 << bits); e += nthreads) {
        const uint32_t c = lh[e];
        if (c) {
            atomicAdd(&fine[e], c);
            atomicAdd(&coarse_lds[e >> shift], c);
        }
    }
    __syncthreads();
    for (int e = tid; e < 256; e += nthreads)
        if (coarse_lds[e]) atomicAdd(&coarse[e], coarse_lds[e]);
}

__device__ __forceinline__ int refl101(int i, int n) {           // BORDER_REFLECT_101; one fold covers a halo shorter than the image
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return (i >= 0 && i < n) ? i : (int)shg::reflect101(i, n);
}

// ---- K1: block mean + cv2.blur(k) (+ cv2.blur(5) when k != 5) + sum(image) -----------------------------------------------------
// grid: (ceil(sw / BT), ceil(sh / BT)), 256 threads, one output pixel each.  Small tiles on purpose: every phase is a short
// chain of dependent accesses, and several workgroups per CU hide each other's.  LDS: S[RH][RW] u32 | H[RH][BT] u32.
constexpr int BT = 16;
constexpr int NT1 = 1024;                    // the canny tile kernel: one pixel of the 16 x 64 tile per thread
// (a memset's worth of zeroes as a kernel, so that it can share a dispatch with the other scans' like every launch of the stage)
struct ZeroWordsArgs {
    uint32_t* dst;
    size_t n_words;
};
__global__ __launch_bounds__(256) void k_zero_words(const ZeroWordsArgs kargs) {
    uint32_t* __restrict__ dst = kargs.dst;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < kargs.n_words; i += (size_t)gridDim.x * 256) dst[i] = 0;
}

struct LimbBlurArgs {
    const uint16_t* img;
    int h, w;
    int64_t pitch;
    int sh, sw, k, vec4;
    uint32_t *keysk, *keys5;
    unsigned long long* acc;
};
__global__ __launch_bounds__(256) void k_limb_blur(const LimbBlurArgs kargs) {
    const uint16_t* __restrict__ img = kargs.img;
    const int h = kargs.h, w = kargs.w, sh = kargs.sh, sw = kargs.sw, k = kargs.k, vec4 = kargs.vec4;
    const int64_t pitch = kargs.pitch;
    uint32_t* __restrict__ keysk = kargs.keysk;
    uint32_t* __restrict__ keys5 = kargs.keys5;
    unsigned long long* __restrict__ acc = kargs.acc;
    __shared__ uint32_t S[(BT + KMAX) * (BT + KMAX)];
    __shared__ uint32_t H[(BT + KMAX) * BT];
    __shared__ unsigned long long wsum[4];
    __shared__ uint32_t wmin[4];
    const int hl = max(k / 2, 2), hr = max(k - 1 - k / 2, 2);           // halo of the union of the two windows
    const int RH = BT + hl + hr, RW = RH;
    const int x0 = blockIdx.x * BT, y0 = blockIdx.y * BT;
    const int tid = threadIdx.x;
    // (the call zeroes the accumulators; k_limb_select1's last workgroup writes the flood slots' minima and maxima)
    uint32_t my_min = 0xffffffffu;                                       // this pixel's window sum of blur k
    // block means of the region, BORDER_REFLECT_101 on the quarter-size image (cv2.blur's border)
    unsigned long long own = 0;
    for (int e = tid; e < RH * RW; e += 256) {
        const int r = e / RW, c = e - r * RW;
        const int sy = refl101(y0 - hl + r, sh), sx = refl101(x0 - hl + c, sw);
        uint32_t s = 0;                                                  // zero padded blocks (block_reduce cval=0)
        if (vec4 && sx * 4 + 4 <= w && sy * 4 + 4 <= h) {
            uint2 q[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) q[j] = *reinterpret_cast<const uint2*>(img + (int64_t)(sy * 4 + j) * pitch + sx * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) s += (q[j].x & 0xffffu) + (q[j].x >> 16) + (q[j].y & 0xffffu) + (q[j].y >> 16);
        } else {
            for (int j = 0; j < 4; ++j) {
                const int y = sy * 4 + j;
                if (y >= h) break;
                for (int i = 0; i < 4; ++i) {
                    const int x = sx * 4 + i;
                    if (x < w) s += img[(int64_t)y * pitch + x];
                }
            }
        }
        S[e] = s;
        const int oy = y0 - hl + r, ox = x0 - hl + c;                    // the tile's own pixels: np.sum(image)
        if (r >= hl && r < hl + BT && c >= hl && c < hl + BT && oy < sh && ox < sw) own += s;
    }
    own = shg::wave_sum((uint64_t)own);
    if ((tid & 63) == 0) wsum[tid >> 6] = own;
    const int n_arrays = k == 5 ? 1 : 2;
    for (int a = 0; a < n_arrays; ++a) {
        const int kw = a == 0 ? k : 5;
        uint32_t* keys = a == 0 ? keysk : keys5;
        __syncthreads();
        // horizontal window sums of the rows this window needs, then vertical
        const int c_off = hl - kw / 2, r_lo = hl - kw / 2, r_n = BT + kw - 1;
        for (int e = tid; e < r_n * BT; e += 256) {
            const int r = r_lo + e / BT, x = e % BT;
            uint32_t s = 0;
            for (int j = 0; j < kw; ++j) s += S[r * RW + x + c_off + j];
            H[r * BT + x] = s;
        }
        __syncthreads();
        const int y = tid / BT, x = tid % BT;
        if (y0 + y < sh && x0 + x < sw) {
            uint32_t s = 0;
            for (int j = 0; j < kw; ++j) s += H[(r_lo + y + j) * BT + x];
            keys[(int64_t)(y0 + y) * sw + x0 + x] = s;
            if (a == 0) my_min = s;
        }
    }
    my_min = shg::wave_fold_u32(my_min, [](uint32_t x, uint32_t y) { return x < y ? x : y; });
    if ((tid & 63) == 0) wmin[tid >> 6] = my_min;
    __syncthreads();
    if (tid == 0) {
        const int slot = (blockIdx.y * gridDim.x + blockIdx.x) % FLOOD_SLOTS;
        const unsigned long long t = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (t) atomicAdd(&acc[4 + 3 * slot], t);
        // the smallest blurred value of the image: what min(blurred[blurred < very_bright]) is, once very_bright is known
        // (k_limb_select1).  Kept as the complement of its order-preserving key, so that zeroed memory is the neutral start.
        const uint32_t m = min(min(wmin[0], wmin[1]), min(wmin[2], wmin[3]));
        if (m != 0xffffffffu) atomicMax(&acc[4 + 3 * FLOOD_SLOTS + 4 + slot], ~f64_key(((double)m * kUnit) * (1.0 / ((double)k * (double)k))));
    }
}

// The bin of a fine histogram (2^bits bins) whose cumulative range holds `rank`, through its coarse companion: the 256 coarse
// counts give the chunk, the chunk's 2^(bits-8) <= 64 fine bins the digit.  ONE WAVE does it (four coarse counts per lane, a
// wave scan, one fine count per lane, a wave scan): no workgroup barrier, so the waves of a workgroup can each take a rank.
// Agent-scope loads (other workgroups wrote the counts).  Every lane returns the result.
__device__ __forceinline__ void wave_pick(const uint32_t* __restrict__ fine, const uint32_t* __restrict__ coarse, int bits, int64_t rank,
                                          int& digit, int64_t& below) {
    const int lane = threadIdx.x & 63;
    const int per = 1 << (bits - 8);
    int64_t c[4], local = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        c[j] = __hip_atomic_load(&coarse[lane * 4 + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        local += c[j];
    }
    int64_t incl = shg::wave_scan((int64_t)local);
    int64_t excl = incl - local;
    int chunk = -1;
    int64_t base = 0;
    if (excl <= rank && rank < incl) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (chunk < 0 && rank < excl + c[j]) { chunk = lane * 4 + j; base = excl; }
            excl += c[j];
        }
    }
    const unsigned long long owner = __ballot(chunk >= 0);
    const int src = owner ? __ffsll((long long)owner) - 1 : 0;
    chunk = __shfl(chunk, src);
    base = __shfl(base, src);
    const int64_t f = lane < per ? (int64_t)__hip_atomic_load(&fine[chunk * per + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    int64_t fi = shg::wave_scan((int64_t)f);
    const int64_t fe = base + fi - f;
    const bool mine = lane < per && fe <= rank && rank < fe + f;
    const unsigned long long who = __ballot(mine);
    const int src2 = who ? __ffsll((long long)who) - 1 : 0;
    digit = chunk * per + src2;
    below = __shfl(fe, src2);
}

struct Ranks4 { int64_t rank[4]; int array[4]; double scale[4]; };       // array: 0 = blur(k), 1 = blur(5)

// ---- K2: first radix pass: the histogram of the window sums' high digit, one per distinct array.  grid (blocks, arrays) -----------
struct LimbSelect0Args {
    const uint32_t *keysk, *keys5;
    int64_t n;
    int bits0, bits1;
    uint32_t *hist0, *coarse0;
};
__global__ __launch_bounds__(256) void k_limb_select0(const LimbSelect0Args kargs) {
    const uint32_t* __restrict__ keysk = kargs.keysk;
    const uint32_t* __restrict__ keys5 = kargs.keys5;
    const int64_t n = kargs.n;
    const int bits0 = kargs.bits0, bits1 = kargs.bits1;
    uint32_t* __restrict__ hist0 = kargs.hist0;
    uint32_t* __restrict__ coarse0 = kargs.coarse0;
    extern __shared__ uint32_t lds[];
    __shared__ uint32_t lc[256];
    const int a = blockIdx.y;
    const uint32_t* __restrict__ v = a ? keys5 : keysk;
    for (int e = threadIdx.x; e < (1 << bits0); e += 256) lds[e] = 0;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t base = (int64_t)blockIdx.x * 256 + threadIdx.x; base < n; base += 4 * stride) {
        uint32_t kk[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int64_t i = base + u * stride; ok[u] = i < n; kk[u] = v[ok[u] ? i : 0]; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (ok[u]) atomicAdd(&lds[kk[u] >> bits1], 1u);
    }
    __syncthreads();
    flush_hist(lds, bits0, lc, hist0 + ((size_t)a << bits0), coarse0 + a * 256, threadIdx.x, 256);
}

// ---- K3: second radix pass, grid (blocks, 4 pairs); the last workgroup through forms the four order statistics
// ((key * 2^-20) * scale: the blur's own arithmetic) and very_bright = np.percentile(blurred, 99) by NumPy's _lerp ---------------
struct LimbSelect1Args {
    const uint32_t *keysk, *keys5;
    int64_t n;
    Ranks4 p;
    int bits0, bits1;
    const uint32_t *hist0, *coarse0;
    uint32_t *hist1, *coarse1;
    double gamma;
    uint32_t* done;
    unsigned long long* acc;
    double* out4;
    int fence;
};
__global__ __launch_bounds__(256) void k_limb_select1(const LimbSelect1Args kargs) {
    const uint32_t* __restrict__ keysk = kargs.keysk;
    const uint32_t* __restrict__ keys5 = kargs.keys5;
    const int64_t n = kargs.n;
    const Ranks4& p = kargs.p;
    const int bits0 = kargs.bits0, bits1 = kargs.bits1;
    const uint32_t* __restrict__ hist0 = kargs.hist0;
    const uint32_t* __restrict__ coarse0 = kargs.coarse0;
    uint32_t* __restrict__ hist1 = kargs.hist1;
    uint32_t* __restrict__ coarse1 = kargs.coarse1;
    const double gamma = kargs.gamma;
    uint32_t* __restrict__ done = kargs.done;
    unsigned long long* __restrict__ acc = kargs.acc;
    double* __restrict__ out4 = kargs.out4;
    extern __shared__ uint32_t lds[];
    __shared__ uint32_t lc[256];
    __shared__ int last;
    __shared__ double val[4];
    const int pair = blockIdx.y;
    const uint32_t* __restrict__ v = p.array[pair] ? keys5 : keysk;
    int digit;
    int64_t below;
    wave_pick(hist0 + ((size_t)p.array[pair] << bits0), coarse0 + p.array[pair] * 256, bits0, p.rank[pair], digit, below);   // (every wave: same result)
    for (int e = threadIdx.x; e < (1 << bits1); e += 256) lds[e] = 0;
    __syncthreads();
    const uint32_t mask = (1u << bits1) - 1u;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t base = (int64_t)blockIdx.x * 256 + threadIdx.x; base < n; base += 4 * stride) {
        uint32_t kk[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int64_t i = base + u * stride; ok[u] = i < n; kk[u] = v[ok[u] ? i : 0]; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (ok[u] && (int)(kk[u] >> bits1) == digit) atomicAdd(&lds[kk[u] & mask], 1u);
    }
    __syncthreads();
    flush_hist(lds, bits1, lc, hist1 + ((size_t)pair << bits1), coarse1 + pair * 256, threadIdx.x, 256);
    published();
    if (threadIdx.x == 0) last = count_done(done, gridDim.x * gridDim.y, kargs.fence);
    __syncthreads();
    if (!last) return;
    {   // wave q takes pair q: both digits, then the value
        const int q = threadIdx.x >> 6;
        int d0, d1;
        int64_t b0, b1;
        wave_pick(hist0 + ((size_t)p.array[q] << bits0), coarse0 + p.array[q] * 256, bits0, p.rank[q], d0, b0);
        wave_pick(hist1 + ((size_t)q << bits1), coarse1 + q * 256, bits1, p.rank[q] - b0, d1, b1);
        const uint64_t key = ((uint64_t)d0 << bits1) | (uint64_t)d1;
        if ((threadIdx.x & 63) == 0) val[q] = ((double)key * kUnit) * p.scale[q];
    }
    __syncthreads();
    __shared__ double vb_s;
    __shared__ int scan_s;
    __shared__ unsigned long long wmax[4];
    if (threadIdx.x == 0) {
        const double a = val[2], b = val[3], diff = b - a;
        const double very_bright = gamma >= 0.5 ? b - diff * (1.0 - gamma) : a + diff * gamma;
        out4[0] = val[0]; out4[1] = val[1]; out4[2] = a; out4[3] = b;
        acc[3] = (unsigned long long)__double_as_longlong(very_bright);
        vb_s = very_bright;
        scan_s = !(a < very_bright);
    }
    __syncthreads();
    // min / max of blurred[blurred < very_bright] (ellipse_to_circle.py:166-169) without another pass over the image: the minimum is
    // the image's smallest value (k_limb_blur left it) if that lies below very_bright; the maximum is the lower of the two order
    // statistics very_bright was interpolated between -- a < very_bright <= b, and no value lies between two consecutive order
    // statistics -- unless the two coincide (very_bright == a: a tie at the 99th percentile, a few scans in a hundred), and
    // then this workgroup looks the image over for the largest value below it.
    const double very_bright = vb_s;
    unsigned long long hi = 0ull;
    if (scan_s) {
        const uint32_t* __restrict__ vk = p.array[2] ? keys5 : keysk;
        for (int64_t i = threadIdx.x; i < n; i += 256) {
            const double bl = ((double)vk[i] * kUnit) * p.scale[2];
            if (bl < very_bright) { const unsigned long long kk = f64_key(bl); hi = kk > hi ? kk : hi; }
        }
        hi = shg::wave_max((uint64_t)hi);
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = hi;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (scan_s) { hi = wmax[0]; for (int i = 1; i < 4; ++i) hi = wmax[i] > hi ? wmax[i] : hi; }
        else hi = f64_key(val[2]);
        unsigned long long inv = 0ull;
        for (int s = 0; s < FLOOD_SLOTS; ++s) { const unsigned long long v = acc[4 + 3 * FLOOD_SLOTS + 4 + s]; inv = v > inv ? v : inv; }
        unsigned long long lo = ~inv;                            // the key of the smallest value
        if (inv == 0ull || !(key_f64(lo) < very_bright)) { lo = ~0ull; hi = 0ull; }      // nothing below very_bright
        for (int s = 0; s < FLOOD_SLOTS; ++s) { acc[5 + 3 * s] = s == 0 ? lo : ~0ull; acc[6 + 3 * s] = s == 0 ? hi : 0ull; }
    }
}

// ---- K4: np.histogram(data, 20) over data = blurred[blurred < very_bright]; the last workgroup to finish stores the stage's
// numbers where the host reads them: packed[0..3] order statistics, [4] sum(image), [5] min, [6] max, then 20 uint32 counts ---
struct LimbFloodHistArgs {
    const uint32_t* keysk;
    int64_t n;
    double scale_k;
    const unsigned long long* acc;
    const double* out4;
    uint32_t *counts, *done;
    double* packed;
    int fence;
};
__global__ __launch_bounds__(256) void k_limb_flood_hist(const LimbFloodHistArgs kargs) {
    const uint32_t* __restrict__ keysk = kargs.keysk;
    const int64_t n = kargs.n;
    const double scale_k = kargs.scale_k;
    const unsigned long long* __restrict__ acc = kargs.acc;
    const double* __restrict__ out4 = kargs.out4;
    uint32_t* __restrict__ counts = kargs.counts;
    uint32_t* __restrict__ done = kargs.done;
    double* __restrict__ packed = kargs.packed;
    const double very_bright = __longlong_as_double((long long)acc[3]);
    __shared__ double edges[21];
    __shared__ uint32_t lc[20];
    __shared__ int last;
    unsigned long long total = 0, klo = ~0ull, khi = 0ull;
    for (int s = 0; s < FLOOD_SLOTS; ++s) {
        total += acc[4 + 3 * s];
        klo = acc[5 + 3 * s] < klo ? acc[5 + 3 * s] : klo;
        khi = acc[6 + 3 * s] > khi ? acc[6 + 3 * s] : khi;
    }
    const double mn = key_f64(klo), mx = key_f64(khi);
    if (threadIdx.x < 21) {
        // np.histogram: first == last -> (first - 0.5, last + 0.5); bin_edges = np.linspace(first, last, 21)
        double first = mn, lastv = mx;
        if (first == lastv) { first = first - 0.5; lastv = lastv + 0.5; }
        const double step = (lastv - first) / 20.0;
        edges[threadIdx.x] = threadIdx.x == 20 ? lastv : (double)threadIdx.x * step + first;
    }
    if (threadIdx.x < 20) lc[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double b = ((double)keysk[i] * kUnit) * scale_k;
        if (!(b < very_bright)) continue;
        int bin = 0;                                  // largest bin with edges[bin] <= b; the last bin is closed
        for (int j = 1; j < 20; ++j) bin = (b >= edges[j]) ? j : bin;
        atomicAdd(&lc[bin], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 20 && lc[threadIdx.x]) atomicAdd(&counts[threadIdx.x], lc[threadIdx.x]);
    published();
    if (threadIdx.x == 0) last = count_done(done, gridDim.x, kargs.fence);
    __syncthreads();
    if (!last) return;
    if (threadIdx.x < 20) {
        uint32_t* pc = reinterpret_cast<uint32_t*>(packed + 8);
        pc[threadIdx.x] = __hip_atomic_load(&counts[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0) {
        packed[0] = out4[0]; packed[1] = out4[1]; packed[2] = out4[2]; packed[3] = out4[3];
        packed[4] = (double)total / 1048576.0;
        packed[5] = mn;
        packed[6] = mx;
    }
}

// ---- K5: canny up to its masks, and the union-find of the low mask inside the tile ----------------------------------------------
// glibc 2.35 hypot (sysdeps/ieee754/dbl-64/e_hypot.c, the non-FMA kernel), as limb.hip
__device__ __forceinline__ double hypot_glibc(double x, double y) {
    x = fabs(x);
    y = fabs(y);
    const double ax = x < y ? y : x;
    const double ay = x < y ? x : y;
    if (ax >= ay / 0x1p-54) return ax + ay;
    double hh = sqrt(ax * ax + ay * ay);
    double t1, t2;
    if (hh <= 2.0 * ay) {
        const double delta = hh - ay;
        t1 = ax * (2.0 * delta - ax);
        t2 = (delta - 2.0 * (ax - ay)) * delta;
    } else {
        const double delta = hh - ax;
        t1 = 2.0 * delta * (ax - 2.0 * ay);
        t2 = (4.0 * delta - ay) * ay + delta * delta;
    }
    hh -= (t1 + t2) / (2.0 * hh);
    return hh;
}

__device__ __forceinline__ int refl(int i, int n) {       // scipy mode 'reflect': d c b a | a b c d | d c b a
    return i < 0 ? -i - 1 : (i >= n ? 2 * n - 1 - i : i);
}

__device__ __forceinline__ int lds_find(volatile int* lab, int x) {
    int p = lab[x];
    while (p != x) { x = p; p = lab[x]; }
    return x;
}

__device__ __forceinline__ void lds_union(int* lab, int a, int b) {
    while (true) {
        a = lds_find(lab, a);
        b = lds_find(lab, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }      // the larger root goes under the smaller
        const int old = atomicMin(&lab[a], b);
        if (old == a) return;
        a = old;
    }
}

// Dynamic LDS (bytes): F u8 [(TH+4+2R)][(TW+4+2R)] | gv f64 [TH+4] | V f64 [(TH+4)][(TW+4+2R)] | Sm f64 [(TH+4)][(TW+4)] |
// I, J, M f64 [(TH+2)][(TW+2)] each | lab int [TH*TW]
struct LimbCannyArgs {
    const uint32_t* keysk;
    int h, w;
    double scale_k, flood_thresh;
    GaussW g;
    double low, high;
    uint8_t* mask;
    int *L, *row_counts;
    int tiles_x;
};
__global__ __launch_bounds__(NT1) void k_limb_canny_tile(const LimbCannyArgs kargs) {
    const uint32_t* __restrict__ keysk = kargs.keysk;
    const int h = kargs.h, w = kargs.w, tiles_x = kargs.tiles_x;
    const double scale_k = kargs.scale_k, flood_thresh = kargs.flood_thresh, low = kargs.low, high = kargs.high;
    const GaussW& g = kargs.g;
    uint8_t* __restrict__ mask = kargs.mask;
    int* __restrict__ L = kargs.L;
    int* __restrict__ row_counts = kargs.row_counts;
    extern __shared__ double lds_d[];
    const int R = g.radius;
    const int FW = TW + 4 + 2 * R, FH = TH + 4 + 2 * R, VW = FW, VH = TH + 4, SW = TW + 4, MW = TW + 2, MH = TH + 2;
    double* gv = lds_d;
    double* V = gv + ((VH + 7) & ~7);
    double* Sm = V + VH * VW;
    double* I = Sm + VH * SW;
    double* J = I + MH * MW;
    double* M = J + MH * MW;
    int* lab = reinterpret_cast<int*>(M + MH * MW);
    uint8_t* F = reinterpret_cast<uint8_t*>(lab + TH * TW);
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    // img_blurred[< thresh3] = 0, [>= thresh3] = 65000 (:226-227); outside the image: 0 (gaussian_filter mode='constant')
    for (int e = tid; e < FH * FW; e += NT1) {
        const int r = e / FW, c = e - r * FW;
        const int y = y0 - 2 - R + r, x = x0 - 2 - R + c;
        uint8_t f = 0;
        if (y >= 0 && y < h && x >= 0 && x < w) f = (((double)keysk[(int64_t)y * w + x] * kUnit) * scale_k < flood_thresh) ? 0 : 1;
        F[e] = f;
    }
    // the all-ones mask after the first axis: a function of the row alone
    if (tid < VH) {
        const int y = y0 - 2 + tid;
        auto one = [&](int yy) -> double { return (yy < 0 || yy >= h) ? 0.0 : 1.0; };
        double u = one(y) * g.w[R];
        for (int j = -R; j < 0; ++j) u += (one(y + j) + one(y - j)) * g.w[R + j];
        gv[tid] = u;
    }
    __syncthreads();
    // Gaussian along axis 0
    for (int e = tid; e < VH * VW; e += NT1) {
        const int r = e / VW, c = e - r * VW;
        const int fr = r + R;                                   // row of F
        auto px = [&](int rr) -> double { return F[rr * FW + c] ? 65000.0 : 0.0; };
        double t = px(fr) * g.w[R];
        for (int j = -R; j < 0; ++j) t += (px(fr + j) + px(fr - j)) * g.w[R + j];
        V[e] = t;
    }
    __syncthreads();
    // Gaussian along axis 1 of both planes, then smoothed = image / (bleed_over + eps)
    for (int e = tid; e < VH * SW; e += NT1) {
        const int r = e / SW, c = e - r * SW;
        const int y = y0 - 2 + r, x = x0 - 2 + c;
        double sm = 0.0;
        if (y >= 0 && y < h && x >= 0 && x < w) {
            const double* a = V + r * VW + c + R;
            const double gvr = gv[r];
            auto b = [&](int xx) -> double { return (xx < 0 || xx >= w) ? 0.0 : gvr; };
            double t = a[0] * g.w[R];
            double u = gvr * g.w[R];
            for (int j = -R; j < 0; ++j) {
                t += (a[j] + a[-j]) * g.w[R + j];
                u += (b(x + j) + b(x - j)) * g.w[R + j];
            }
            sm = t / (u + 2.220446049250313e-16);
        }
        Sm[e] = sm;
    }
    __syncthreads();
    // ndi.sobel along both axes (mode 'reflect') and the magnitude, one pixel beyond the tile
    for (int e = tid; e < MH * MW; e += NT1) {
        const int r = e / MW, c = e - r * MW;
        const int y = y0 - 1 + r, x = x0 - 1 + c;
        double iv = 0.0, jv = 0.0, mg = 0.0;
        if (y >= 0 && y < h && x >= 0 && x < w) {
            const int ym = refl(y - 1, h), yp = refl(y + 1, h), xm = refl(x - 1, w), xp = refl(x + 1, w);
            auto S = [&](int yy, int xx) -> double { return Sm[(yy - (y0 - 2)) * SW + (xx - (x0 - 2))]; };
            auto dy = [&](int xx) -> double { double t = S(y, xx) * 0.0; t += (S(ym, xx) - S(yp, xx)) * -1.0; return t; };
            auto dx = [&](int yy) -> double { double t = S(yy, x) * 0.0; t += (S(yy, xm) - S(yy, xp)) * -1.0; return t; };
            iv = dy(x) * 2.0;
            iv += (dy(xm) + dy(xp)) * 1.0;
            jv = dx(y) * 2.0;
            jv += (dx(ym) + dx(yp)) * 1.0;
            mg = hypot_glibc(iv, jv);
        }
        I[e] = iv; J[e] = jv; M[e] = mg;
    }
    __syncthreads();
    // 4-sector non-maximum suppression with interpolation, the two thresholds, and the tile's labels
    for (int e = tid; e < TH * TW; e += NT1) {
        const int ty = e / TW, tx = e - ty * TW;
        const int y = y0 + ty, x = x0 + tx;
        uint8_t bits = 0;
        if (y < h && x < w) {
            const int me = (ty + 1) * MW + tx + 1;
            const double m = M[me];
            bool local = false;
            if (y > 0 && y < h - 1 && x > 0 && x < w - 1 && m > 0.0) {
                const double is = I[me], js = J[me];
                const double ai = fabs(is), aj = fabs(js);
                auto Mn = [&](int dy, int dx) -> double { return M[me + dy * MW + dx]; };
                const bool same = (is >= 0 && js >= 0) || (is <= 0 && js <= 0);
                const bool opp = (is <= 0 && js >= 0) || (is >= 0 && js <= 0);
                auto test = [&](double wgt, double p1, double p2, double m1, double m2) -> bool {
                    const bool c_plus = p2 * wgt + p1 * (1 - wgt) <= m;
                    const bool c_minus = m2 * wgt + m1 * (1 - wgt) <= m;
                    return c_plus && c_minus;
                };
                if (same && ai >= aj) local = test(aj / ai, Mn(1, 0), Mn(1, 1), Mn(-1, 0), Mn(-1, -1));
                if (same && ai <= aj) local = test(ai / aj, Mn(0, 1), Mn(1, 1), Mn(0, -1), Mn(-1, -1));
                if (opp && ai <= aj) local = test(ai / aj, Mn(0, 1), Mn(-1, 1), Mn(0, -1), Mn(1, -1));
                if (opp && ai >= aj) local = test(aj / ai, Mn(-1, 0), Mn(-1, 1), Mn(1, 0), Mn(1, -1));
            }
            bits = (uint8_t)(((local && m >= low) ? 1 : 0) | ((local && m >= high) ? 2 : 0));
            mask[(int64_t)y * w + x] = bits;
        }
        lab[e] = (bits & 1) ? e : -1;
    }
    __syncthreads();
    // 8-connected union-find inside the tile: every low pixel links to its W, NW, N, NE neighbours
    for (int e = tid; e < TH * TW; e += NT1) {
        if (lab[e] < 0) continue;
        const int ty = e / TW, tx = e - ty * TW;
        if (tx > 0 && lab[e - 1] >= 0) lds_union(lab, e, e - 1);
        if (ty > 0) {
            if (tx > 0 && lab[e - TW - 1] >= 0) lds_union(lab, e, e - TW - 1);
            if (lab[e - TW] >= 0) lds_union(lab, e, e - TW);
            if (tx < TW - 1 && lab[e - TW + 1] >= 0) lds_union(lab, e, e - TW + 1);
        }
    }
    __syncthreads();
    // labels out: the global index of the tile-local root (local raster order = global raster order inside a tile)
    for (int e = tid; e < TH * TW; e += NT1) {
        const int ty = e / TW, tx = e - ty * TW;
        const int y = y0 + ty, x = x0 + tx;
        if (y >= h || x >= w) continue;
        int gl = -1;
        if (lab[e] >= 0) {
            const int r = lds_find(lab, e);
            gl = (y0 + r / TW) * w + x0 + (r % TW);
        }
        L[(int64_t)y * w + x] = gl;
    }
    // low pixels per (row, tile): the emission's raster offsets
    if (tid < TH && y0 + tid < h) {
        int c = 0;
        for (int tx = 0; tx < TW; ++tx) c += lab[tid * TW + tx] >= 0 ? 1 : 0;
        row_counts[(y0 + tid) * tiles_x + blockIdx.x] = c;
    }
}

// ---- K6: unions across tile borders (global atomics, as limb.hip's k_ccl_merge) ---------------------------------------------------
__device__ __forceinline__ int g_find(const int* L, int x) {
    int p = __hip_atomic_load(&L[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != x) { x = p; p = __hip_atomic_load(&L[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    return x;
}

__device__ __forceinline__ void g_union(int* L, int a, int b) {
    while (true) {
        a = g_find(L, a);
        b = g_find(L, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(&L[a], b);
        if (old == a) return;
        a = old;
    }
}

// one thread per pixel of a tile's first row (blockIdx.y = 0: tile row, x), first or last column (blockIdx.y = 1: column line, y)
struct LimbBorderArgs {
    const uint8_t* mask;
    int h, w;
    int* L;
};
__global__ __launch_bounds__(256) void k_limb_border_merge(const LimbBorderArgs kargs) {
    const uint8_t* __restrict__ mask = kargs.mask;
    const int h = kargs.h, w = kargs.w;
    int* __restrict__ L = kargs.L;
    const int j = blockIdx.x * 256 + threadIdx.x;
    int y, x;
    if (blockIdx.y == 0) {
        const int line = j / w;
        y = line * TH;
        x = j - line * w;
        if (y >= h) return;
    } else {
        const int line = j / h;                                // columns 0, TW-1, TW, 2TW-1, 2TW, ...
        y = j - line * h;
        x = (line + 1) / 2 * TW - ((line & 1) ? 1 : 0);
        if (x >= w || (y % TH) == 0) return;                   // (a tile's first row is the other half's)
    }
    const int i = y * w + x;
    if (!(mask[i] & 1)) return;
    const bool top = (y % TH) == 0, left = (x % TW) == 0, right = (x % TW) == TW - 1;
    if (left && x > 0 && (mask[i - 1] & 1)) g_union(L, i, i - 1);
    if (y > 0) {
        const int up = i - w;
        if ((top || left) && x > 0 && (mask[up - 1] & 1)) g_union(L, i, up - 1);
        if (top && (mask[up] & 1)) g_union(L, i, up);
        if ((top || right) && x < w - 1 && (mask[up + 1] & 1)) g_union(L, i, up + 1);
    }
}

// ---- K7: every low pixel in raster order with its component's root, the high bit in bit 30 of the root: comp = [m | idx[n] | root[n]]
// grid: h rows.
struct LimbEmitArgs {
    const uint8_t* mask;
    const int* L;
    int h, w;
    const int* row_counts;
    int tiles_x, n;
    int* comp;
};
__global__ __launch_bounds__(256) void k_limb_emit(const LimbEmitArgs kargs) {
    const uint8_t* __restrict__ mask = kargs.mask;
    const int* __restrict__ L = kargs.L;
    const int h = kargs.h, w = kargs.w, tiles_x = kargs.tiles_x, n = kargs.n;
    const int* __restrict__ row_counts = kargs.row_counts;
    int* __restrict__ comp = kargs.comp;
    __shared__ int wave_cnt[4];
    __shared__ int base;
    __shared__ int part[4];
    const int y = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int before = 0;
    for (int e = threadIdx.x; e < y * tiles_x; e += 256) before += row_counts[e];
    before = shg::wave_sum(before);
    if (lane == 0) part[wave] = before;
    __syncthreads();
    if (threadIdx.x == 0) base = part[0] + part[1] + part[2] + part[3];
    __syncthreads();
    for (int xb = 0; xb < w; xb += 256) {
        const int x = xb + threadIdx.x;
        int root = -1, hi = 0;
        if (x < w) {
            const int i = y * w + x;
            const int l = L[i];
            if (l >= 0) {
                root = g_find(L, l);
                hi = (mask[i] >> 1) & 1;
            }
        }
        const unsigned long long m = __ballot(root >= 0);
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int i = 0; i < wave; ++i) off += wave_cnt[i];
        if (root >= 0) {
            const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
            comp[1 + pos] = y * w + x;
            comp[1 + n + pos] = root | (hi << 30);
        }
        __syncthreads();
        if (threadIdx.x == 0) base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
    if (y == h - 1 && threadIdx.x == 0) comp[0] = base;
}

size_t canny_tile_lds(int R) {
    const int FW = TW + 4 + 2 * R, FH = TH + 4 + 2 * R, VW = FW, VH = TH + 4, SW = TW + 4, MW = TW + 2, MH = TH + 2;
    return (size_t)(((VH + 7) & ~7) + VH * VW + VH * SW + 3 * MH * MW) * 8 + (size_t)TH * TW * 4 + (size_t)FH * FW;
}

}  // namespace

// Whether the fused kernels take this image: the blur window (and its radix digits) must fit the LDS tile.
extern "C" int shg_limb_fused_fits(int64_t sh, int64_t sw, int k) {
    return sh > 2 && sw > 2 && sh * sw < (1ll << 30) && k >= 1 && k <= KMAX;
}

// words at the head of shg_limb_prepare's workspace that it zeroes for the disk [h][w] (the block-mean image is h/4 x w/4, the blur
// window int(0.01 * h/4)); 0 when the fused kernels do not take that image
size_t shg::limb_prepare_zero_words(int64_t h, int64_t w) {
    const int64_t sh = (h + 3) / 4, sw = (w + 3) / 4;
    const int k = (int)((double)sh * 0.01);
    if (!shg_limb_fused_fits(sh, sw, k)) return 0;
    return prep_layout(sh, sw, k).zero_words;
}

extern "C" size_t shg_limb_prepare_workspace_bytes(int64_t sh, int64_t sw, int k) {
    if (!shg_limb_fused_fits(sh, sw, k)) return 0;
    return prep_layout(sh, sw, k).total_words * 4;
}

// get_flood_image's reductions for the disk `img` (uint16 [h][w], rows `pitch` apart) on its 4x4 block mean [sh][sw]
// (ellipse_to_circle.py:159-169, 241, 299-301): cv2.blur with windows k and 5, the order statistics host_ranks4 = {two for
// np.median(blur 5), two for np.percentile(blur k, 99)}, very_bright by NumPy's _lerp with weight gamma99, sum(image), min /
// max / np.histogram(., 20) of blur k below very_bright.  packed (device, or GPU-mapped host memory): [0..3] the order
// statistics, [4] sum, [5] min, [6] max, then 20 uint32 counts at packed + 8.  *keys_out: the window sums of blur k in the
// workspace (what shg_limb_edges takes).
extern "C" int shg_limb_prepare(const uint16_t* img, int64_t h, int64_t w, int64_t pitch, int k, const int64_t* host_ranks4, double gamma99,
                                double* packed, const uint32_t** keys_out, void* workspace, size_t workspace_bytes, shg_stream_t stream) {
    SHG_REQUIRE(img && host_ranks4 && packed && keys_out && workspace, SHG_E_ARG, "shg_limb_prepare: null pointer");
    SHG_REQUIRE(h > 0 && w > 0 && pitch >= w, SHG_E_ARG, "shg_limb_prepare: bad image size");
    const int64_t sh = (h + 3) / 4, sw = (w + 3) / 4, n = sh * sw;
    SHG_REQUIRE(shg_limb_fused_fits(sh, sw, k), SHG_E_UNSUPPORTED, "shg_limb_prepare: a %d x %d window on %lld x %lld is outside the fused path", k, k,
                (long long)sh, (long long)sw);
    SHG_REQUIRE(gamma99 >= 0.0 && gamma99 <= 1.0, SHG_E_ARG, "shg_limb_prepare: gamma %g outside [0, 1]", gamma99);
    const PrepLayout lay = prep_layout(sh, sw, k);
    SHG_REQUIRE(workspace_bytes >= lay.total_words * 4, SHG_E_WORKSPACE, "shg_limb_prepare: workspace too small");
    SHG_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 7) == 0, SHG_E_ARG, "shg_limb_prepare: workspace not 8-byte aligned");
    for (int i = 0; i < 4; ++i) SHG_REQUIRE(host_ranks4[i] >= 0 && host_ranks4[i] < n, SHG_E_ARG, "shg_limb_prepare: rank outside the image");
    hipStream_t st = shg::as_stream(stream);
    uint32_t* ws = static_cast<uint32_t*>(workspace);
    uint32_t *hist0 = ws + lay.hist0, *hist1 = ws + lay.hist1, *coarse0 = ws + lay.coarse0, *coarse1 = ws + lay.coarse1;
    uint32_t *counts = ws + lay.counts, *done = ws + lay.done;
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(ws + lay.acc);
    uint32_t *keysk = ws + lay.keysk, *keys5 = k == 5 ? keysk : ws + lay.keys5;
    double* out4 = reinterpret_cast<double*>(acc + 4 + 3 * FLOOD_SLOTS);
    SHG_PROF("limb_prepare", st);
    if (shg::t_prezeroed == workspace) {                       // the extraction's last launch has cleared them on its way (shg_scan_file)
        shg::t_prezeroed = nullptr;
    } else if (int e = shg::launch(k_zero_words, dim3((unsigned)std::min<size_t>((lay.zero_words + 255) / 256, 64)), dim3(256), 0, st, ZeroWordsArgs{ws, lay.zero_words}, "k_zero_words")) {
        return e;
    }
    Ranks4 p;
    const double scale_k = 1.0 / ((double)k * (double)k), scale_5 = 1.0 / 25.0;
    for (int i = 0; i < 4; ++i) { p.rank[i] = host_ranks4[i]; p.array[i] = (i < 2 && k != 5) ? 1 : 0; p.scale[i] = (i < 2) ? scale_5 : scale_k; }
    dim3 grid1((unsigned)((sw + BT - 1) / BT), (unsigned)((sh + BT - 1) / BT));
    const int vec4 = (reinterpret_cast<uintptr_t>(img) & 7) == 0 && pitch % 4 == 0;
    if (int e = shg::launch(k_limb_blur, grid1, dim3(256), 0, st, LimbBlurArgs{img, (int)h, (int)w, pitch, (int)sh, (int)sw, k, vec4, keysk, keys5, acc}, "k_limb_blur")) return e;
    static const bool lds_ok = [] {                             // a 2^14-bin histogram is the default 64 KB of dynamic LDS, to the byte
        return hipFuncSetAttribute(reinterpret_cast<const void*>(k_limb_select0), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess &&
               hipFuncSetAttribute(reinterpret_cast<const void*>(k_limb_select1), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess;
    }();
    if (!lds_ok) (void)hipGetLastError();
    int64_t blocks = (n + 2047) / 2048;
    if (blocks > 256) blocks = 256;
    if (int e = shg::launch(k_limb_select0, dim3((unsigned)blocks, k == 5 ? 1u : 2u), dim3(256), ((size_t)1 << lay.bits0) * 4, st,
                           LimbSelect0Args{keysk, keys5, n, lay.bits0, lay.bits1, hist0, coarse0}, "k_limb_select0"))
        return e;
    if (int e = shg::launch(k_limb_select1, dim3((unsigned)blocks, 4u), dim3(256), ((size_t)1 << lay.bits1) * 4, st,
                           LimbSelect1Args{keysk, keys5, n, p, lay.bits0, lay.bits1, hist0, coarse0, hist1, coarse1, gamma99, done, acc, out4, limb_fence()}, "k_limb_select1"))
        return e;
    *keys_out = keysk;
    return shg::launch(k_limb_flood_hist, dim3((unsigned)blocks), dim3(256), 0, st, LimbFloodHistArgs{keysk, n, scale_k, acc, out4, counts, done + 1, packed, limb_fence()}, "k_limb_flood_hist");
}

extern "C" size_t shg_limb_edges_workspace_bytes(int64_t sh, int64_t sw) {
    if (sh <= 0 || sw <= 0) return 0;
    const size_t n = (size_t)sh * (size_t)sw, tiles_x = (size_t)((sw + TW - 1) / TW);
    return ((n + 255) / 256 * 256) + n * 4 + (size_t)sh * tiles_x * 4 + 256;
}

// skimage.feature.canny(flooded, sigma, low, high) and the labelling of its low mask (ellipse_to_circle.py:245-252) on the
// blurred image given by its window sums (blurred = keys * 2^-20 * 1/(k*k); flooded = blurred < flood_thresh ? 0 : 65000):
// comp (device, or GPU-mapped host memory) = [m | idx[n] | root[n]], the m pixels of the LOW mask in raster order, root =
// smallest linear index of the pixel's 8-connected component, bit 30 of root set where the pixel is in the HIGH mask -- the
// hysteresis (keep the components that hold a high pixel) is one pass over the list for the caller.
extern "C" int shg_limb_edges(const uint32_t* keys, int64_t sh, int64_t sw, int k, double flood_thresh, const double* host_gauss_weights,
                              int radius, double low, double high, int32_t* comp, void* workspace, size_t workspace_bytes, shg_stream_t stream) {
    SHG_REQUIRE(keys && host_gauss_weights && comp && workspace, SHG_E_ARG, "shg_limb_edges: null pointer");
    SHG_REQUIRE(sh > 2 && sw > 2 && sh * sw < (1ll << 30) && k >= 1, SHG_E_ARG, "shg_limb_edges: bad image size");
    SHG_REQUIRE(radius >= 0 && radius <= MAXR, SHG_E_UNSUPPORTED, "shg_limb_edges: Gaussian radius %d > %d", radius, MAXR);
    SHG_REQUIRE(workspace_bytes >= shg_limb_edges_workspace_bytes(sh, sw), SHG_E_WORKSPACE, "shg_limb_edges: workspace too small");
    GaussW g;
    g.radius = radius;
    for (int i = 0; i < 2 * radius + 1; ++i) g.w[i] = host_gauss_weights[i];
    const size_t n = (size_t)sh * (size_t)sw;
    const int tiles_x = (int)((sw + TW - 1) / TW);
    uint8_t* mask = static_cast<uint8_t*>(workspace);
    int* L = reinterpret_cast<int*>(mask + (n + 255) / 256 * 256);
    int* row_counts = L + n;
    hipStream_t st = shg::as_stream(stream);
    SHG_PROF("limb_edges", st);
    dim3 grid((unsigned)tiles_x, (unsigned)((sh + TH - 1) / TH));
    static const bool lds_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(k_limb_canny_tile), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
    if (!lds_ok) (void)hipGetLastError();
    if (int e = shg::launch(k_limb_canny_tile, grid, dim3(NT1), canny_tile_lds(radius), st,
                           LimbCannyArgs{keys, (int)sh, (int)sw, 1.0 / ((double)k * (double)k), flood_thresh, g, low, high, mask, L, row_counts, tiles_x}, "k_limb_canny_tile"))
        return e;
    {
        const int64_t tops = ((sh + TH - 1) / TH) * sw, sides = 2 * (int64_t)tiles_x * sh;
        if (int e = shg::launch(k_limb_border_merge, dim3((unsigned)((std::max(tops, sides) + 255) / 256), 2u), dim3(256), 0, st, LimbBorderArgs{mask, (int)sh, (int)sw, L}, "k_limb_border_merge"))
            return e;
    }
    return shg::launch(k_limb_emit, dim3((unsigned)sh), dim3(256), 0, st, LimbEmitArgs{mask, L, (int)sh, (int)sw, row_counts, tiles_x, (int)n, comp}, "k_limb_emit");
}
