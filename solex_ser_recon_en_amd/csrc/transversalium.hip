// Transversalium (row-defect) correction: device parts.
// Reference: correct_transversalium2 (solex_util.py:383-395, 489, 515-516) and
// reject_outliers (solex_util.py:76-86).
//
// k_rowpair_stats: one workgroup per row pair.  The log-ratios of the chord are held in LDS as
// order-preserving 64-bit keys; the median and the MAD are exact order statistics read by a bucket select:
// one histogram of the keys over [min, max] in 512 equal steps (a monotone map, so buckets are ordered like the
// keys), the handful of keys in the bucket that holds the wanted rank gathered and ranked directly -- two sweeps over
// the row per statistic.  Rows with infinities (zero pixels) or pathological spreads take the general MSB-first radix
// select (8 bits per pass, both middle order statistics in the same passes; up to ten sweeps).  Then the 2-MAD
// inliers are averaged.  All float64.  (A select moves ~16x less LDS data than sorting the row.)
// The reference sums the inliers in image order with NumPy's pairwise scheme; here they
// are summed by a fixed-shape tree, so the mean can differ in the last bits (as NumPy's
// own log already does between CPUs); the correction factors agree to ~1e-15 relative.
//
// k_scale_rows: img * c[y], saturate, truncate -- a pure streaming pass.
#include <math.h>
#include <algorithm>
#include "shg_common.h"
#include "fast_log.h"

namespace {

constexpr int MAXN = SHG_TRANSV_MAX_COLS;   // 19456 keys = 152 KiB of the CU's 160 KiB LDS (plus the static scratch)
constexpr int NT = 256;

__device__ __forceinline__ uint64_t f64_key(double v) {          // monotone map double -> uint64 (no NaN)
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_f64(uint64_t k) {
    const uint64_t b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

constexpr int NB = 512;          // buckets of the bucket select (= the radix select's two 256-bin histograms)
constexpr int CAP = 128;         // candidates ranked directly

struct alignas(16) Scratch {
    uint32_t hist[2][256];       // radix select: one histogram per rank; bucket select: one of NB bins
    int64_t wave_tot[2][4];
    int64_t pick[2][3];          // [which rank][digit / bucket, count below, count in the bin]
    unsigned long long found[2]; // the key a singleton bin holds (early exit of select2) / the ranked candidates
    unsigned long long cand[CAP];
    unsigned long long kmin, kmax;
    uint32_t n_cand;
    double red[8];
    double vfound[2];            // bucket select: the two values found
    int bad, nonfinite;
    uint32_t pick32[8];          // bucket select: [bucket of rank_lo, values before it, values in it, -, bucket of rank_hi, values in it]
    float redf[8];               // the sums that only steer the bucket select
    uint32_t kept[4];            // the inliers each wave counted
};

// one histogram increment with wave-level aggregation of the most common digit
__device__ __forceinline__ void hist_add(uint32_t* hist, bool valid, uint32_t digit) {
    const unsigned long long act = __ballot(valid);
    if (act == 0) return;
    const int leader = __ffsll((long long)act) - 1;
    const uint32_t d0 = __shfl(digit, leader);
    const unsigned long long same = __ballot(valid && digit == d0);
    if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[d0], (uint32_t)__popcll(same));
    if (valid && digit != d0) atomicAdd(&hist[digit], 1u);
}

// Inclusive prefix sum across the wave, without LDS (round 6: DPP row shifts inside each row of 16 lanes -- a lane without a source
// adds 0 --, then the totals of the rows before a lane's own from scalar registers).  The ds_bpermute ladder it replaces was six
// dependent LDS round trips.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_or_zero(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
    v += dpp_or_zero<0x111>(v);          // row_shr:1
    v += dpp_or_zero<0x112>(v);          // row_shr:2
    v += dpp_or_zero<0x114>(v);          // row_shr:4
    v += dpp_or_zero<0x118>(v);          // row_shr:8
    const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 15), t1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 31),
                   t2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 47);
    const int lane = threadIdx.x & 63;
    return v + (lane >= 16 ? t0 : 0u) + (lane >= 32 ? t1 : 0u) + (lane >= 48 ? t2 : 0u);
}
__device__ __forceinline__ int64_t wave_inclusive_scan(int64_t v) {          // (the radix select's counts: 64 bits, the rare path)
    int self = threadIdx.x & 63;
    asm volatile("" : "+v"(self));
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int64_t o = __shfl(v, self - d);
        if (self >= d) v += o;
    }
    return v;
}

// The two order statistics rank_lo <= rank_hi (0-based) of key(i), i < n, for the whole workgroup.
template <typename KeyFn>
__device__ __forceinline__ void select2(KeyFn key, int n, int64_t rank_lo, int64_t rank_hi, Scratch& sc, uint64_t& out_lo,
                                        uint64_t& out_hi) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint64_t pref[2] = {0, 0};
    int64_t rank[2] = {rank_lo, rank_hi};
    bool same = true;                                  // both ranks still follow the same prefix
    for (int p = 0; p < 8; ++p) {
        const int shift = 56 - 8 * p;
        sc.hist[0][tid] = 0;
        sc.hist[1][tid] = 0;
        __syncthreads();
        for (int i0 = 0; i0 < n; i0 += NT) {
            const int i = i0 + tid;
            const bool in = i < n;
            const uint64_t k = in ? key(i) : 0;
            const uint64_t top = p == 0 ? 0 : (k >> (shift + 8));
            const uint32_t digit = (uint32_t)(k >> shift) & 0xffu;
            hist_add(sc.hist[0], in && (p == 0 || top == pref[0]), digit);
            if (!same) hist_add(sc.hist[1], in && top == pref[1], digit);
        }
        __syncthreads();
        // workgroup-wide inclusive scan of both histograms, one bin per thread
        int64_t c[2], incl[2];
        c[0] = sc.hist[0][tid];
        c[1] = same ? c[0] : sc.hist[1][tid];
        incl[0] = wave_inclusive_scan(c[0]);
        incl[1] = wave_inclusive_scan(c[1]);
        if (lane == 63) { sc.wave_tot[0][wave] = incl[0]; sc.wave_tot[1][wave] = incl[1]; }
        __syncthreads();
        for (int i = 0; i < wave; ++i) { incl[0] += sc.wave_tot[0][i]; incl[1] += sc.wave_tot[1][i]; }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int64_t excl = incl[r] - c[r];
            if (excl <= rank[r] && rank[r] < incl[r]) { sc.pick[r][0] = tid; sc.pick[r][1] = excl; sc.pick[r][2] = c[r]; }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            rank[r] -= sc.pick[r][1];
            pref[r] = (pref[r] << 8) | (uint64_t)sc.pick[r][0];
        }
        same = same && (pref[0] == pref[1]);
        const bool singletons = sc.pick[0][2] == 1 && sc.pick[1][2] == 1;
        __syncthreads();
        // Both ranks alone in their bins: the remaining digits are those of the one key with that prefix.  For real
        // data (distinct log-ratios) this happens after 2-3 of the 8 passes; rows with many equal keys run them all.
        if (singletons && p < 7) {
            for (int i = tid; i < n; i += NT) {
                const uint64_t k = key(i);
                if ((k >> shift) == pref[0]) sc.found[0] = k;
                if ((k >> shift) == pref[1]) sc.found[1] = k;
            }
            __syncthreads();
            out_lo = sc.found[0];
            out_hi = sc.found[1];
            __syncthreads();
            return;
        }
    }
    out_lo = pref[0];
    out_hi = pref[1];
}

// body(i) for i = tid, tid + NT, ... < n with the trip count in a scalar register: n / NT whole rounds that every lane takes -- no
// per-lane index compare, no juggling of the exec mask: three of the eight to ten VALU instructions an element cost in the plain
// `for (i = tid; i < n; i += NT)` form, and this kernel is bound by its VALU instructions -- and one last round for the lanes below n % NT.
// (n must be uniform over the workgroup.)
template <bool UNROLL = true, typename F>
__device__ __forceinline__ void sweep(int n, F body) {
    const int rounds = __builtin_amdgcn_readfirstlane(n / NT);
    if (UNROLL) {
#pragma unroll 2
        for (int r = 0; r < rounds; ++r) body((int)threadIdx.x + r * NT);
    } else {                                 // (a body with a ballot: the compiler will not unroll it, and says so)
        for (int r = 0; r < rounds; ++r) body((int)threadIdx.x + r * NT);
    }
    const int last = (int)threadIdx.x + rounds * NT;
    if (last < n) body(last);
}

// The same two order statistics by bucketing, for keys that are all finite, and ADJACENT ranks (rank_hi - rank_lo is 0 or 1: a
// median).  x -> min(NB-1, max(0, int(x * inv + off))) never decreases with x (a fused multiply-add rounds a rising function once),
// so every key of a bucket is <= every key of the next one and the bucket holding a rank is found by a prefix sum; its keys (a dozen
// of ~2000 for real rows) are ranked against each other.  A crowded bucket is bucketed again between its own extrema.  Returns false
// (block-uniform) when it gives up: the caller then runs select2.
//
// Round 6: this kernel's time follows the vector instructions it issues (150 M wave instructions a C4 launch, 135 a pixel pair,
// profiles/r06_sq_k_rowpair_stats.json; 21 % fewer of them: 14 % less time) and a thread has only ~7 values, so what a select costs per WORKGROUP counts as much as what it costs
// per value: (a) the buckets' prefix sum and the search for the two ranks are one wave's work (eight buckets a lane) instead of
// every wave repeating them, in 32-bit counts; (b) 1 / (hi - lo) is the hardware's approximate reciprocal -- the map only has to
// be monotone --, the bucket one multiply-add; (c) the candidates are the values whose position lies in [b0, b1 + 1): two compares
// instead of convert / clamp / two compares (the buckets between two adjacent ranks' are empty).
template <typename ValFn>
__device__ __forceinline__ bool select2_buckets(ValFn val, int n, int rank_lo, int rank_hi, double lo, double hi,
                                                Scratch& sc, double& out_lo, double& out_hi) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t* hist = &sc.hist[0][0];
    double* cand = reinterpret_cast<double*>(sc.cand);
    uint32_t below = 0;                                 // values smaller than wlo
    double wlo = 0.0, whi = 0.0;                        // the values still in play, from the second level on (all of them at first)
    for (int level = 0; level < 4; ++level) {
        if (level > 0) {
            if (wlo == whi) { out_lo = out_hi = wlo; return true; }
            lo = wlo;
            hi = whi;
        }
        // [lo, hi] only has to spread the keys: values beyond it fall into the end buckets, which keeps the map monotone.
        // The caller passes the bulk of the distribution (mean +- 3 sigma) rather than the extrema: with the range set by
        // a few outliers the bulk shares a dozen buckets and the LDS atomics below serialise (measured: 69 us instead of
        // 51 for the radix select; profiles/).
        const double inv = (double)NB * __builtin_amdgcn_rcp(hi - lo);
        const double off = -lo * inv;
        if (!(inv > 0.0) || inv > 1.7e308 || !(fabs(off) <= 1.7e308)) return false;
        auto place = [&](double v) { return fma(v, inv, off); };      // (the sweeps read the values as they are: no sort keys made and unmade)
        auto bucket = [&](double v) {
            int b = (int)place(v);                       // toward zero, saturating (v_cvt_i32_f64): still monotone
            b = b < 0 ? 0 : b;
            return b > NB - 1 ? NB - 1 : b;
        };
        auto in_play = [&](double v) { return level == 0 || (v >= wlo && v <= whi); };
        hist[tid] = 0;
        hist[tid + 256] = 0;
        if (tid == 0) sc.n_cand = 0;
        __syncthreads();
        sweep(n, [&](int i) {
            const double v = val(i);
            if (in_play(v)) atomicAdd(&hist[bucket(v)], 1u);
        });
        __syncthreads();
        if (wave == 0) {
            // prefix sum over the buckets, eight per lane; p[k] = the values up to and including the lane's bucket k
            const uint4 h0 = *reinterpret_cast<const uint4*>(&hist[8 * lane]), h1 = *reinterpret_cast<const uint4*>(&hist[8 * lane + 4]);
            const uint32_t c[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
            const uint32_t tot = ((c[0] + c[1]) + (c[2] + c[3])) + ((c[4] + c[5]) + (c[6] + c[7]));
            const uint32_t run = below + wave_inclusive_scan(tot) - tot;
            uint32_t p[8];
            p[0] = run + c[0];
#pragma unroll
            for (int k = 1; k < 8; ++k) p[k] = p[k - 1] + c[k];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const uint32_t rk = (uint32_t)(r ? rank_hi : rank_lo);
                if (run <= rk && rk < p[7]) {            // this lane's buckets hold the rank: walk them
                    uint32_t b = 0, before = run, in = c[0];
#pragma unroll
                    for (int k = 0; k < 7; ++k) {
                        const bool past = rk >= p[k];    // (true for every k before the bucket, false from it on)
                        b += past ? 1u : 0u;
                        before = past ? p[k] : before;
                        in = past ? c[k + 1] : in;
                    }
                    sc.pick32[4 * r] = 8u * (uint32_t)lane + b;
                    sc.pick32[4 * r + 1] = before;
                    sc.pick32[4 * r + 2] = in;
                }
            }
        }
        __syncthreads();
        const int b0 = (int)sc.pick32[0], b1 = (int)sc.pick32[4];
        const uint32_t base = sc.pick32[1];
        const uint32_t cnt = sc.pick32[2] + (b1 != b0 ? sc.pick32[6] : 0u);
        if (cnt <= (uint32_t)CAP) {
            // adjacent ranks: the buckets between b0 and b1 are empty, so the candidates -- every value of the buckets b0 ... b1 --
            // are consecutive in rank.  bucket >= b0 <=> place >= b0 (or b0 = 0), bucket <= b1 <=> place < b1 + 1 (or b1 = NB - 1).
            const double from = b0 == 0 ? -__builtin_huge_val() : (double)b0, to = b1 == NB - 1 ? __builtin_huge_val() : (double)(b1 + 1);
            sweep(n, [&](int i) {
                const double v = val(i);
                const double t = place(v);
                if (in_play(v) && t >= from && t < to) cand[atomicAdd(&sc.n_cand, 1u)] = v;
            });
            __syncthreads();
            const int m = (int)sc.n_cand;
            if (tid < m) {
                const double mine = cand[tid];      // (-0.0 and +0.0 tie here and have different keys in select2: same correction)
                uint32_t r = base;
                for (int j = 0; j < m; ++j) {
                    const double o = cand[j];
                    r += (o < mine || (o == mine && j < tid)) ? 1u : 0u;
                }
                if (r == (uint32_t)rank_lo) sc.vfound[0] = mine;
                if (r == (uint32_t)rank_hi) sc.vfound[1] = mine;
            }
            __syncthreads();
            out_lo = sc.vfound[0];
            out_hi = sc.vfound[1];
            __syncthreads();
            return true;
        }
        if (b0 != b1) return false;                     // a crowded bucket next to the one with the other rank: rare enough
        if (tid == 0) { sc.kmin = ~0ull; sc.kmax = 0ull; }
        __syncthreads();
        uint64_t mn = ~0ull, mx = 0ull;                 // (extrema through the keys: 64-bit atomics order them)
        for (int i = tid; i < n; i += NT) {
            const double v = val(i);
            if (in_play(v) && bucket(v) == b0) { const uint64_t k = f64_key(v); mn = k < mn ? k : mn; mx = k > mx ? k : mx; }
        }
        if (mn <= mx) { atomicMin(&sc.kmin, (unsigned long long)mn); atomicMax(&sc.kmax, (unsigned long long)mx); }
        __syncthreads();
        wlo = key_f64(sc.kmin);
        whi = key_f64(sc.kmax);
        below = base;
        __syncthreads();
    }
    return false;
}

// The sum of a double over the wave, in every lane, without LDS (round 6): four DPP steps inside each row of 16 lanes (quad
// butterflies, then the mirrored half and the mirrored row: every lane ends with its row's sum), then the four rows' sums read into
// scalar registers and added.  The ds_bpermute butterfly it replaces was six dependent LDS round trips per value (~1000 cycles a
// call): k_rowpair_stats' two block sums were 38 of its 300 us at C4 (profiles/r06_sweeps.txt).  The tree's shape -- and with it the
// last bits of a sum -- differs from the butterfly's; it is as fixed as that one was.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_f64<0xB1>(v);               // quad_perm [1, 0, 3, 2]
    v += dpp_f64<0x4E>(v);               // quad_perm [2, 3, 0, 1]
    v += dpp_f64<0x141>(v);              // row_half_mirror
    v += dpp_f64<0x140>(v);              // row_mirror
    auto row = [&](int l) { return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l)); };
    return (row(0) + row(16)) + (row(32) + row(48));
}

__device__ __forceinline__ double block_sum(double v, double* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// two sums in one round of barriers (red: 8 doubles)
__device__ __forceinline__ void block_sum2(double& a, double& b, double* red) {
    a = wave_sum(a);
    b = wave_sum(b);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = a; red[4 + (threadIdx.x >> 6)] = b; }
    __syncthreads();
    a = (red[0] + red[1]) + (red[2] + red[3]);
    b = (red[4] + red[5]) + (red[6] + red[7]);
}

// The same for two sums that only have to be roughly right (the bulk of a row's values, for the bucket select's range): rounded to
// float first, where a DPP step is ONE instruction (v_add_f32_dpp) instead of two moves and an add -- 11 instructions a sum
// instead of 26.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f32<0xB1>(v);
    v += dpp_f32<0x4E>(v);
    v += dpp_f32<0x141>(v);
    v += dpp_f32<0x140>(v);
    auto row = [&](int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
    return (row(0) + row(16)) + (row(32) + row(48));
}
__device__ __forceinline__ void block_sum2_rough(float& a, float& b, float* red) {
    a = wave_sum(a);
    b = wave_sum(b);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = a; red[4 + (threadIdx.x >> 6)] = b; }
    __syncthreads();
    a = (red[0] + red[1]) + (red[2] + red[3]);
    b = (red[4] + red[5]) + (red[6] + red[7]);
}

// grid (rows, 1, disks): blockIdx.z picks the image and its slice (out_stride doubles apart) of out / mirror
struct RowpairArgs {
    shg::PtrBatch imgs;
    int64_t pitch, y1;
    const int32_t *xa, *xb;
    const double* row_factor;
    double *out, *mirror;
    int64_t out_stride;
};

__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(8))) void k_rowpair_stats(const RowpairArgs kargs) {
    const shg::PtrBatch& imgs = kargs.imgs;
    const int64_t pitch = kargs.pitch, y1 = kargs.y1, out_stride = kargs.out_stride;
    const int32_t* __restrict__ xa = kargs.xa;
    const int32_t* __restrict__ xb = kargs.xb;
    const double* __restrict__ row_factor = kargs.row_factor;
    double* __restrict__ out = kargs.out;
    double* __restrict__ mirror = kargs.mirror;
    extern __shared__ double vals[];     // [n]: the log-ratios of the chord
    __shared__ Scratch sc;
    const uint16_t* __restrict__ img = imgs.at<const uint16_t>(blockIdx.z);
    out += (int64_t)blockIdx.z * out_stride;
    if (mirror) mirror += (int64_t)blockIdx.z * out_stride;
    const int t = blockIdx.x + 1;        // out[0] stays 0 (solex_util.py:386)
    auto emit = [&](double v) {          // mirror: the same values where the host reads them (pinned memory), if wanted
        if (threadIdx.x == 0) {
            out[t] = v;
            if (mirror) mirror[t] = v;
            if (t == 1) { out[0] = 0.0; if (mirror) mirror[0] = 0.0; }
        }
    };
    const int64_t y = y1 + t;
    const int a = xa[t], b = xb[t];
    const int n = b - a;
    if (n <= 0) {                        // np.mean of an empty slice
        emit(__builtin_nan(""));
        return;
    }
    if (threadIdx.x == 0) { sc.bad = 0; sc.nonfinite = 0; }
    __syncthreads();
    const uint16_t* r1 = img + y * pitch + a;
    const uint16_t* r0 = img + (y - 1) * pitch + a;
    // Zero pixels give -inf / +inf / NaN ratios.  NumPy keeps infinities as ordinary (sortable) values and
    // lets any NaN poison the row statistic (np.median -> nan -> empty inlier set -> nan); same here.
    // a de-vignetted frame is the float64 image img * row_factor[y] (removeVignette, solex_util.py:654)
    const double f1 = row_factor ? row_factor[y] : 1.0, f0 = row_factor ? row_factor[y - 1] : 1.0;
    double sum1 = 0.0, sum2 = 0.0;       // only steer the bucket select (where the bulk of the row lies): any rounding will do
    if (!row_factor) {
        // Plain 16-bit rows (every scan but a de-vignetted one): log_ratio_u16 -- one reciprocal serves the quotient and the
        // logarithm's own division -- for pixel pairs without a zero; four pairs' loads in flight before the first is used.
        bool odd = false;
        auto one = [&](uint32_t a, uint32_t b, int i) {
            // (a pair with a zero pixel fails log_ratio_u16's range test like any quotient far from 1: fast_log.h)
            const double x = shg::log_ratio_u16(a, b, [&](double q) {
                if (a != 0 && b != 0) return shg::log_normal(q);
                const double l = log((double)a / (double)b);      // 0, inf or NaN quotient -- the library's log knows what to return
                if (l != l) sc.bad = 1;
                odd = odd || !(fabs(l) <= 1.7976931348623157e308);
                return l;
            });
            vals[i] = x;
            sum1 += x;
            sum2 = fma(x, x, sum2);
        };
        // whole groups of 4 x NT pairs first (a scalar trip count, no index clamps or per-lane bounds: see sweep()), then the rest
        const int groups = __builtin_amdgcn_readfirstlane(n / (4 * NT));
        for (int g = 0; g < groups; ++g) {
            const int i0 = (int)threadIdx.x + g * 4 * NT;
            uint32_t a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a[u] = r1[i0 + u * NT];
                b[u] = r0[i0 + u * NT];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) one(a[u], b[u], i0 + u * NT);
        }
        {
            const int i0 = (int)threadIdx.x + groups * 4 * NT;
            if (i0 < n) {
                uint32_t a[4], b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + u * NT < n ? i0 + u * NT : i0;
                    a[u] = r1[i];
                    b[u] = r0[i];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (i0 + u * NT >= n) break;
                    one(a[u], b[u], i0 + u * NT);
                }
            }
        }
        if (odd) sc.nonfinite = 1;
    } else {
        bool odd = false;
        for (int i = threadIdx.x; i < n; i += NT) {
            const double q = ((double)r1[i] * f1) / ((double)r0[i] * f0);              // np.log(strip1 / strip0)
            // (zero pixels make 0, inf or NaN quotients: those go through the library's log, which knows what to return)
            const double x = (q >= 2.2250738585072014e-308 && q <= 1.7976931348623157e308) ? shg::log_normal(q) : log(q);
            if (x != x) sc.bad = 1;
            odd = odd || !(fabs(x) <= 1.7976931348623157e308);
            vals[i] = x;
            sum1 += x;
            sum2 = fma(x, x, sum2);
        }
        if (odd) sc.nonfinite = 1;
    }
    // Where the bulk of the row lies, roughly: single precision and the hardware's approximate reciprocal and square root will do
    // (a range that spreads the values badly costs the bucket select a second level, never a wrong answer).
    float rough1 = (float)sum1, rough2 = (float)sum2;
    block_sum2_rough(rough1, rough2, sc.redf);
    __syncthreads();
    if (sc.bad) {
        emit(__builtin_nan(""));
        return;
    }
    const bool finite = !sc.nonfinite;
    const float per_n = __builtin_amdgcn_rcpf((float)n);
    const float meanf = rough1 * per_n, varf = rough2 * per_n - meanf * meanf;
    const double mean = (double)meanf;
    const double sigma = varf > 0.0f ? (double)__builtin_amdgcn_sqrtf(varf) : 0.0;
    // np.median: the middle order statistic, or the mean of the two middle ones
    const int lo = (n & 1) ? (n >> 1) : (n >> 1) - 1, hi = n >> 1;
    double va, vb;
    auto general = [&](auto key) {                 // the radix select, over sort keys made on the fly
        uint64_t ka, kb;
        select2(key, n, lo, hi, sc, ka, kb);
        va = key_f64(ka);
        vb = key_f64(kb);
    };
    // The buckets' range: mean +- 1 sigma for the median, [0, 1 sigma] for the MAD (0.67 sigma for clean data).  A row's sigma is set by the
    // few pixel pairs at its limb ends (0.06 where the bulk's is 0.007: in-kernel clock + counts, profiles/r06_sweeps.txt), and with +- 3 /
    // [0, 2] sigma such a row's middle bucket held 45 - 120 candidates -- whose ranking against each other (a loop of LDS reads per
    // candidate) made its selects take twice a clean row's.  A median beyond 1 sigma of the mean lands in an end bucket: second level.
    const double kMedSpan = 1.0, kMadSpan = 1.0;
    auto row_val = [&](int i) { return vals[i]; };
    if (!(finite && select2_buckets(row_val, n, lo, hi, mean - kMedSpan * sigma, mean + kMedSpan * sigma, sc, va, vb)))
        general([&](int i) { return f64_key(vals[i]); });
    const double med = (n & 1) ? va : (va + vb) / 2.0;
    if (!finite) {                       // (finite values and a finite median cannot make a NaN)
        for (int i = threadIdx.x; i < n; i += NT) {
            const double dv = fabs(vals[i] - med);                      // inf - inf or a NaN median -> NaN
            if (dv != dv) sc.bad = 1;
        }
        __syncthreads();
        if (sc.bad) {
            emit(__builtin_nan(""));
            return;
        }
    }
    auto dev_val = [&](int i) { return fabs(vals[i] - med); };
    if (!(finite && select2_buckets(dev_val, n, lo, hi, 0.0, kMadSpan * sigma, sc, va, vb)))
        general([&](int i) { return f64_key(fabs(vals[i] - med)); });    // |x - med| >= 0: its bit pattern is already order preserving
    const double mdev = (n & 1) ? va : (va + vb) / 2.0;
    // `dev / mdev < 2` as `dev < 2 * mdev`: the rounded quotient is below 2 exactly when the true one is below 2 - 2^-53, and no
    // double lies in [2 mdev (1 - 2^-54), 2 mdev) -- the two tests agree for every pair of doubles (infinities and NaN included),
    // and the sweep has no division.
    const double twice = 2.0 * mdev;
    const bool all = !(mdev != 0.0);                                       // s = d/mdev if mdev else zeros; data[s < 2] (see above)
    double s = 0.0;
    uint32_t kept = 0;                                                     // (the wave's count, in a scalar register)
    sweep<false>(n, [&](int i) {
        const double x = vals[i];
        const bool keep = all || fabs(x - med) < twice;
        kept += (uint32_t)__popcll(__ballot(keep));
        if (keep) s += x;
    });
    s = wave_sum(s);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sc.red[threadIdx.x >> 6] = s; sc.kept[threadIdx.x >> 6] = kept; }
    __syncthreads();
    s = (sc.red[0] + sc.red[1]) + (sc.red[2] + sc.red[3]);
    const double cnt = (double)((sc.kept[0] + sc.kept[1]) + (sc.kept[2] + sc.kept[3]));
    emit(s / cnt);
}

__global__ __launch_bounds__(256) void k_scale_rows(shg::PtrBatch imgs, int64_t w, int64_t pitch,
                                                    const double* __restrict__ c, const double* __restrict__ row_factor,
                                                    shg::PtrBatch dsts, int64_t dst_pitch) {
    const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t y = blockIdx.y;
    if (x >= w) return;
    const uint16_t* __restrict__ img = imgs.at<const uint16_t>(blockIdx.z);
    uint16_t* __restrict__ dst = dsts.at<uint16_t>(blockIdx.z);
    c += (int64_t)blockIdx.z * gridDim.y;
    double v = (double)img[y * pitch + x];
    if (row_factor) v = v * row_factor[y];              // the float64 de-vignetted pixel, rounded as NumPy stores it
    v = v * c[y];
    v = v > 65535.0 ? 65535.0 : v;
    dst[y * dst_pitch + x] = (uint16_t)(int)v;
}

// The same with eight pixels per lane (rows 16-byte aligned, pitches multiples of 8; a row's last, partial vector goes
// pixel by pixel) and SCALE_ROWS rows per lane, their loads issued before the first use (one row per workgroup: one request
// in flight per wave, 2.8 TB/s).
// grid (ceil(vectors per row * row groups / 256), 1, disks): blockIdx.z picks source, destination and the disk's h factors
constexpr int SCALE_ROWS = 4;
__global__ __launch_bounds__(256) void k_scale_rows8(shg::PtrBatch imgs, int64_t h, int64_t w, int64_t pitch,
                                                     const double* __restrict__ c, const double* __restrict__ row_factor,
                                                     shg::PtrBatch dsts, int64_t dst_pitch) {
    // lanes are dealt (row group, vector) pairs in one flat sequence (an (x, y) grid wastes every second workgroup on a width
    // just past a multiple of 2048 pixels: 2096 at C2)
    const uint32_t nv = (uint32_t)((w + 7) / 8);
    const uint32_t flat = blockIdx.x * 256u + threadIdx.x;
    const uint32_t yg = flat / nv;
    const int64_t x = (int64_t)(flat - yg * nv) * 8;
    const int64_t ya = (int64_t)yg * SCALE_ROWS;
    if (ya >= h) return;
    const uint16_t* __restrict__ img = imgs.at<const uint16_t>(blockIdx.z);
    uint16_t* __restrict__ dst = dsts.at<uint16_t>(blockIdx.z);
    c += (int64_t)blockIdx.z * h;
    const bool factored = row_factor != nullptr;
    const bool full = x + 8 <= w;
    uint4 q[SCALE_ROWS];
    double cy[SCALE_ROWS], fy[SCALE_ROWS];
#pragma unroll
    for (int rr = 0; rr < SCALE_ROWS; ++rr) {
        const int64_t y = ya + rr < h ? ya + rr : h - 1;
        if (full) q[rr] = *reinterpret_cast<const uint4*>(img + y * pitch + x);
        cy[rr] = c[y];
        fy[rr] = factored ? row_factor[y] : 1.0;
    }
#pragma unroll
    for (int rr = 0; rr < SCALE_ROWS; ++rr) {
        const int64_t y = ya + rr;
        if (y >= h) break;
        const double cyr = cy[rr], fyr = fy[rr];
        auto one = [&](uint32_t px) {
            double v = (double)px;
            if (factored) v = v * fyr;
            v = v * cyr;
            v = v > 65535.0 ? 65535.0 : v;
            return (uint32_t)(int)v;
        };
        if (full) {
            uint4 o;
            o.x = one(q[rr].x & 0xffffu) | (one(q[rr].x >> 16) << 16);
            o.y = one(q[rr].y & 0xffffu) | (one(q[rr].y >> 16) << 16);
            o.z = one(q[rr].z & 0xffffu) | (one(q[rr].z >> 16) << 16);
            o.w = one(q[rr].w & 0xffffu) | (one(q[rr].w >> 16) << 16);
            *reinterpret_cast<uint4*>(dst + y * dst_pitch + x) = o;
        } else {
            for (int64_t i = x; i < w; ++i) dst[y * dst_pitch + i] = (uint16_t)one(img[y * pitch + i]);
        }
    }
}

// scipy.ndimage.correlate1d(rows, weights, axis=-1, mode='constant', cval=0) for k rows of n float64 samples, in
// NI_Correlate1D's own order of operations (the interior of scipy.signal.savgol_filter, solex_util.py:400):
//   symmetric weights : t = x[0]*w[0]; for j = -R..-1: t += (x[j] + x[-j]) * w[j]      (w indexed from the centre)
//   antisymmetric     : t = x[0]*w[0]; for j = -R..-1: t += (x[j] - x[-j]) * w[j]
//   otherwise         : t = x[R]*w[R]; for j = -R..R-1: t += x[j] * w[j]
// One workgroup per (row, 256 output samples): the samples it needs (256 + 2R, zero beyond the row's ends) and the
// weights are staged in LDS first, so the inner loop has no bounds test and no global load to wait for (with
// predicated global loads a single 1800-sample row took 30 us: one request in flight per lane).
constexpr int CORR_MAXR = 1024;

struct CorrelateArgs {
    const double* src;
    int64_t n;
    const double* weights;
    int radius, symmetric;
    double* dst;
};

__global__ __launch_bounds__(256) void k_correlate1d_rows(const CorrelateArgs kargs) {
    const double* __restrict__ src = kargs.src;
    const int64_t n = kargs.n;
    const double* __restrict__ weights = kargs.weights;
    const int radius = kargs.radius, symmetric = kargs.symmetric;
    double* __restrict__ dst = kargs.dst;
    extern __shared__ double corr_lds[];                 // [256 + 2R] samples, then [2R + 1] weights
    double* xs = corr_lds;
    double* ws = corr_lds + 256 + 2 * radius;
    const int64_t row = blockIdx.y, x0 = (int64_t)blockIdx.x * 256;
    const double* line = src + row * n;
    for (int i = threadIdx.x; i < 256 + 2 * radius; i += 256) {
        const int64_t xx = x0 - radius + i;
        xs[i] = (xx >= 0 && xx < n) ? line[xx] : 0.0;
    }
    for (int i = threadIdx.x; i < 2 * radius + 1; i += 256) ws[i] = weights[i];
    __syncthreads();
    const int64_t x = x0 + threadIdx.x;
    if (x >= n) return;
    const double* c = xs + radius + threadIdx.x;         // this lane's sample
    const double* w = ws + radius;
    double t;
    if (symmetric > 0) {
        t = c[0] * w[0];
        for (int j = -radius; j < 0; ++j) t += (c[j] + c[-j]) * w[j];
    } else if (symmetric < 0) {
        t = c[0] * w[0];
        for (int j = -radius; j < 0; ++j) t += (c[j] - c[-j]) * w[j];
    } else {
        t = c[radius] * w[radius];
        for (int j = -radius; j < radius; ++j) t += c[j] * w[j];
    }
    dst[row * n + x] = t;
}

// Two order statistics of every row (axis 1) or column (axis 0) of a uint16 image: np.percentile(img, q, axis)
// (removeVignette, solex_util.py:591-592).  One workgroup per line, 16-bit keys, two 8-bit radix passes.
__global__ __launch_bounds__(NT) void k_line_order_stats(const uint16_t* __restrict__ img, int64_t pitch, int n, int64_t line_stride,
                                                         int64_t elem_stride, int rank_lo, int rank_hi,
                                                         uint16_t* __restrict__ out_lo, uint16_t* __restrict__ out_hi) {
    __shared__ uint32_t hist[256];
    __shared__ int pick[2][2];
    const uint16_t* line = img + (int64_t)blockIdx.x * line_stride;
    int rank[2] = {rank_lo, rank_hi};
    int hi_digit[2] = {0, 0};
    int result[2] = {0, 0};
    // pass 0: high byte, one histogram serves both ranks; pass 1: low byte within the chosen high byte (per rank)
    for (int pass = 0; pass < 3; ++pass) {
        if (pass == 2 && hi_digit[0] == hi_digit[1]) break;        // both ranks share the high byte: pass 1 served both
        const int which = pass == 2 ? 1 : 0;
        hist[threadIdx.x] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += NT) {
            const uint32_t v = line[(int64_t)i * elem_stride];
            if (pass == 0) atomicAdd(&hist[v >> 8], 1u);
            else if ((int)(v >> 8) == hi_digit[which]) atomicAdd(&hist[v & 0xff], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {                                    // 256 bins: a serial scan is a few hundred cycles
            for (int r = 0; r < 2; ++r) {
                if (pass == 2 && r == 0) continue;
                if (pass == 1 && r == 1 && hi_digit[1] != hi_digit[0]) continue;
                int below = 0, d = 0;
                for (; d < 256; ++d) {
                    const int c = (int)hist[d];
                    if (below + c > rank[r]) break;
                    below += c;
                }
                pick[r][0] = d;
                pick[r][1] = below;
            }
        }
        __syncthreads();
        for (int r = 0; r < 2; ++r) {
            if (pass == 2 && r == 0) continue;
            if (pass == 1 && r == 1 && hi_digit[1] != hi_digit[0]) continue;
            if (pass == 0) { hi_digit[r] = pick[r][0]; rank[r] -= pick[r][1]; }
            else result[r] = (hi_digit[r] << 8) | pick[r][0];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out_lo[blockIdx.x] = (uint16_t)result[0];
        out_hi[blockIdx.x] = (uint16_t)result[1];
    }
}

}  // namespace

extern "C" int shg_line_order_stats_u16(const uint16_t* img, int64_t h, int64_t w, int64_t pitch, int axis, int64_t rank_lo,
                                        int64_t rank_hi, uint16_t* out_lo, uint16_t* out_hi, shg_stream_t stream) {
    SHG_REQUIRE(img && out_lo && out_hi, SHG_E_ARG, "shg_line_order_stats_u16: null pointer");
    SHG_REQUIRE(h > 0 && w > 0 && pitch >= w && (axis == 0 || axis == 1), SHG_E_ARG, "shg_line_order_stats_u16: bad arguments");
    const int64_t n = axis == 0 ? h : w, lines = axis == 0 ? w : h;
    SHG_REQUIRE(n < (1ll << 31) && rank_lo >= 0 && rank_lo <= rank_hi && rank_hi < n, SHG_E_ARG,
                "shg_line_order_stats_u16: ranks [%lld, %lld] outside a line of %lld", (long long)rank_lo, (long long)rank_hi, (long long)n);
    hipStream_t st = shg::as_stream(stream);
    SHG_PROF("line_order_stats", st);
    k_line_order_stats<<<(unsigned)lines, NT, 0, st>>>(img, pitch, (int)n, axis == 0 ? 1 : pitch, axis == 0 ? pitch : 1, (int)rank_lo,
                                                      (int)rank_hi, out_lo, out_hi);
    return shg::check_launch("k_line_order_stats");
}

extern "C" int shg_rowpair_logratio_stats(const uint16_t* img, int64_t h, int64_t w, int64_t pitch, int64_t y1, int64_t y2,
                                          const int32_t* xa, const int32_t* xb, const double* row_factor, double* out,
                                          shg_stream_t stream) {
    return shg_rowpair_logratio_stats_mirrored(img, h, w, pitch, y1, y2, xa, xb, row_factor, out, nullptr, stream);
}

extern "C" int shg_rowpair_logratio_stats_mirrored(const uint16_t* img, int64_t h, int64_t w, int64_t pitch, int64_t y1, int64_t y2,
                                                   const int32_t* xa, const int32_t* xb, const double* row_factor, double* out,
                                                   double* out_mirror, shg_stream_t stream) {
    SHG_REQUIRE(img, SHG_E_ARG, "shg_rowpair_logratio_stats: null pointer");
    return shg::rowpair_stats_batch(&img, 1, h, w, pitch, y1, y2, xa, xb, row_factor, out, out_mirror, stream);
}

// The k images of a Doppler stack in one launch (they share circle, borders and shape, Solex_recon.py:105-133): out /
// out_mirror are [k][max(y2 - y1, 1)].
int shg::rowpair_stats_batch(const uint16_t* const* host_imgs, int64_t k, int64_t h, int64_t w, int64_t pitch, int64_t y1, int64_t y2,
                             const int32_t* xa, const int32_t* xb, const double* row_factor, double* out, double* out_mirror,
                             shg_stream_t stream) {
    SHG_REQUIRE(host_imgs && xa && xb && out && k > 0, SHG_E_ARG, "shg_rowpair_logratio_stats: null pointer");
    SHG_REQUIRE(h > 0 && w > 0 && pitch >= w, SHG_E_ARG, "shg_rowpair_logratio_stats: bad image size");
    SHG_REQUIRE(y1 >= 0 && y2 <= h && y2 > y1, SHG_E_ARG, "shg_rowpair_logratio_stats: rows [%lld, %lld) outside the image",
                (long long)y1, (long long)y2);
    SHG_REQUIRE(w <= MAXN, SHG_E_UNSUPPORTED, "shg_rowpair_logratio_stats: width %lld > %d", (long long)w, MAXN);
    for (int64_t i = 0; i < k; ++i) SHG_REQUIRE(host_imgs[i], SHG_E_ARG, "shg_rowpair_logratio_stats: null image");
    hipStream_t st = shg::as_stream(stream);
    const int64_t rows = y2 - y1 - 1, n = y2 - y1;
    if (rows <= 0) {                                     // a single row: its statistic is the leading 0 (the kernel writes it otherwise)
        hipError_t e = hipMemsetAsync(out, 0, (size_t)k * sizeof(double), st);
        if (e == hipSuccess && out_mirror) e = hipMemsetAsync(out_mirror, 0, (size_t)k * sizeof(double), st);
        if (e != hipSuccess) {
            shg::set_error("shg_rowpair_logratio_stats: memset: %s", hipGetErrorString(e));
            return (int)e;
        }
        return 0;
    }
    const size_t lds_bytes = (size_t)w * sizeof(double);
    static const bool attr_set =                         // (a function-local static: initialised once, also with several pool threads here)
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_rowpair_stats), hipFuncAttributeMaxDynamicSharedMemorySize, MAXN * 8) == hipSuccess;
    (void)attr_set;
    SHG_PROF("rowpair_stats", st);
    for (int64_t i0 = 0; i0 < k; i0 += shg::kMaxBatch) {
        const int m = (int)std::min<int64_t>(shg::kMaxBatch, k - i0);
        if (int e = shg::launch(k_rowpair_stats, dim3((unsigned)rows, 1u, (unsigned)m), dim3(NT), lds_bytes, st,
                               RowpairArgs{shg::make_batch(host_imgs, (int)i0, m), pitch, y1, xa, xb, row_factor, out + i0 * n, out_mirror ? out_mirror + i0 * n : nullptr, n}, "k_rowpair_stats"))
            return e;
    }
    return 0;
}

extern "C" int shg_correlate1d_rows_f64(const double* src, int64_t k, int64_t n, const double* weights, int radius, int symmetric,
                                        double* dst, shg_stream_t stream) {
    SHG_REQUIRE(src && weights && dst, SHG_E_ARG, "shg_correlate1d_rows_f64: null pointer");
    SHG_REQUIRE(k > 0 && k < 65536 && n > 0 && radius >= 0, SHG_E_ARG, "shg_correlate1d_rows_f64: bad sizes");
    SHG_REQUIRE(radius <= CORR_MAXR, SHG_E_UNSUPPORTED, "shg_correlate1d_rows_f64: radius %d > %d", radius, CORR_MAXR);
    hipStream_t st = shg::as_stream(stream);
    const size_t lds = (size_t)(256 + 4 * radius + 1) * sizeof(double);
    SHG_PROF("correlate1d_rows", st);
    static const bool attr_set = hipFuncSetAttribute(reinterpret_cast<const void*>(k_correlate1d_rows), hipFuncAttributeMaxDynamicSharedMemorySize, (256 + 4 * CORR_MAXR + 1) * 8) == hipSuccess;
    (void)attr_set;
    return shg::launch(k_correlate1d_rows, dim3((unsigned)((n + 255) / 256), (unsigned)k), dim3(256), lds, st,
                      CorrelateArgs{src, n, weights, radius, symmetric > 0 ? 1 : (symmetric < 0 ? -1 : 0), dst}, "k_correlate1d_rows");
}

extern "C" int shg_scale_rows_u16(const uint16_t* img, int64_t h, int64_t w, int64_t pitch, const double* c,
                                  const double* row_factor, uint16_t* dst, int64_t dst_pitch, shg_stream_t stream) {
    SHG_REQUIRE(img && dst, SHG_E_ARG, "shg_scale_rows_u16: null pointer");
    return shg::scale_rows_batch(&img, 1, h, w, pitch, c, row_factor, &dst, dst_pitch, stream);
}

// k images in one launch: c is [k][h] (row_factor, if given, applies to all of them)
int shg::scale_rows_batch(const uint16_t* const* host_imgs, int64_t k, int64_t h, int64_t w, int64_t pitch, const double* c,
                          const double* row_factor, uint16_t* const* host_dsts, int64_t dst_pitch, shg_stream_t stream) {
    SHG_REQUIRE(host_imgs && c && host_dsts && k > 0, SHG_E_ARG, "shg_scale_rows_u16: null pointer");
    SHG_REQUIRE(h > 0 && w > 0 && pitch >= w && dst_pitch >= w, SHG_E_ARG, "shg_scale_rows_u16: bad image size");
    SHG_REQUIRE(h < 65536, SHG_E_UNSUPPORTED, "shg_scale_rows_u16: more than 65535 rows");
    hipStream_t st = shg::as_stream(stream);
    uintptr_t bits = 0;
    for (int64_t i = 0; i < k; ++i) {
        SHG_REQUIRE(host_imgs[i] && host_dsts[i], SHG_E_ARG, "shg_scale_rows_u16: null image");
        bits |= reinterpret_cast<uintptr_t>(host_imgs[i]) | reinterpret_cast<uintptr_t>(host_dsts[i]);
    }
    const bool vec = (bits & 15) == 0 && pitch % 8 == 0 && dst_pitch % 8 == 0;
    SHG_PROF("scale_rows", st);
    for (int64_t i0 = 0; i0 < k; i0 += shg::kMaxBatch) {
        const int m = (int)std::min<int64_t>(shg::kMaxBatch, k - i0);
        const shg::PtrBatch src = shg::make_batch(host_imgs, (int)i0, m), dst = shg::make_batch(host_dsts, (int)i0, m);
        if (vec) {                                       // eight pixels per lane: 16-byte loads and stores
            const int64_t lanes = ((w + 7) / 8) * ((h + SCALE_ROWS - 1) / SCALE_ROWS);
            dim3 grid((unsigned)((lanes + 255) / 256), 1u, (unsigned)m);
            k_scale_rows8<<<grid, 256, 0, st>>>(src, h, w, pitch, c + i0 * h, row_factor, dst, dst_pitch);
        } else {
            dim3 grid((unsigned)((w + 255) / 256), (unsigned)h, (unsigned)m);
            k_scale_rows<<<grid, 256, 0, st>>>(src, w, pitch, c + i0 * h, row_factor, dst, dst_pitch);
        }
        if (int e = shg::check_launch("k_scale_rows")) return e;
    }
    return 0;
}
