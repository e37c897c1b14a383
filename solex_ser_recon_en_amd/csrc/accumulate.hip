// Pass A: sum and max of every pixel over the frame stack.
// Replaces the frame loop of compute_mean_max (reference solex_util.py:174-188).
//
// HBM-bound streaming read: N*H*W*B bytes in, 6..10 bytes per pixel out.  Layout:
// the stack stays in file layout [N][H*W]; a lane owns 16 contiguous bytes of the
// frame (8 u16 / 16 u8 pixels) and walks the frame axis, so every wave-instruction
// is one fully coalesced 1 KiB global_load_dwordx4.  The frame axis is cut into
// `nsplit` ranges so that the grid has >= ~16 waves per CU; each range writes a
// partial (u32 sum, u16 max) slab that k_reduce_partials folds (integer, order
// independent, so any split / any rank sharding gives identical bits).
#include <math.h>
#include <stdlib.h>
#include <mutex>
#include <unordered_map>
#include <algorithm>
#include "shg_common.h"

namespace {

typedef unsigned short __attribute__((ext_vector_type(2))) ushort2_t;
typedef unsigned int __attribute__((ext_vector_type(4))) u32x4;   // native vector: nontemporal-loadable

struct Plan {
    int64_t npix, frame_bytes, vecs;   // vecs = 16-byte vectors per frame (vector path only)
    int64_t stride_px, stride_vecs;    // distance between consecutive frames (>= npix: a padded frame pitch)
    int nsplit, frames_per_split, unroll;
    bool vector_path;
};

// Launch-shape overrides for the tuning sweeps in tools/ (SHG_ACC_*).  Read once per process: the plan is made on every
// launch and must not go through getenv each time.
struct Tuning {
    int inflight_kib, nsplit, unroll, xcd, nt, lds_kib, prio, interleave;
};

const Tuning& tuning() {
    static const Tuning t = [] {
        auto env_int = [](const char* name, int dflt) {
            const char* s = getenv(name);
            return (s && *s) ? atoi(s) : dflt;
        };
        return Tuning{env_int("SHG_ACC_INFLIGHT_KIB", 7168), env_int("SHG_ACC_NSPLIT", 0), env_int("SHG_ACC_UNROLL", 0),
                      env_int("SHG_ACC_XCD", 0), env_int("SHG_ACC_NT", 1), env_int("SHG_ACC_LDS_KIB", 0), env_int("SHG_ACC_PRIO", 1),
                      env_int("SHG_ACC_INTERLEAVE", 0)};
    }();
    return t;
}

Plan make_plan(const void* stack, int64_t n, int64_t h, int64_t w, int bpp, int64_t frame_stride_px) {
    Plan p;
    p.npix = h * w;
    p.frame_bytes = p.npix * bpp;
    p.stride_px = frame_stride_px > 0 ? frame_stride_px : p.npix;
    p.stride_vecs = p.stride_px * bpp / 16;
    // the alignment of `stack` is only known at call time; hipMalloc/torch give >= 256 B,
    // so the workspace query (stack == nullptr) assumes an aligned base
    p.vector_path = (p.frame_bytes % 16 == 0) && ((p.stride_px * bpp) % 16 == 0) && ((reinterpret_cast<uintptr_t>(stack) & 15) == 0);
    p.vecs = p.frame_bytes / 16;
    // Launch shape.  Measured on MI355X (tools/sweep_acc*.sh): the read rate peaks when about 7 MiB of loads are
    // in flight chip-wide (waves x unroll x 1 KiB) -- 6.7 TB/s at C2 -- and drops on either side (fewer: latency
    // bound; more: 5.5-6.2 TB/s, the extra concurrent frame streams cost DRAM locality).  So choose
    // (nsplit, unroll) for that footprint instead of for maximum occupancy.
    const int64_t wave_cols = p.vector_path ? (p.vecs + 63) / 64 : (p.npix + 63) / 64;
    const double target_kib = (double)tuning().inflight_kib;
    int64_t min_split = (n + 65536) / 65537;                    // a u32 partial holds 65537 full-scale frames
    if (min_split < 1) min_split = 1;
    double best = 1e30;
    int best_split = (int)min_split, best_unroll = 4;
    const int unrolls[3] = {4, 2, 8};
    const double unroll_penalty[3] = {0.0, 0.03, 0.05};
    for (int ui = 0; ui < 3; ++ui) {
        for (int64_t sp = min_split; sp <= 64 && sp <= n; ++sp) {
            const double kib = (double)wave_cols * (double)sp * unrolls[ui];
            double score = kib > target_kib ? kib / target_kib : target_kib / kib;
            score = log(score) + 0.01 * (double)sp + unroll_penalty[ui];
            if (score < best) { best = score; best_split = (int)sp; best_unroll = unrolls[ui]; }
        }
    }
    int64_t nsplit = best_split;
    const int forced = tuning().nsplit;
    if (forced > 0) nsplit = forced > n ? n : forced;
    p.nsplit = (int)nsplit;
    p.frames_per_split = (int)((n + nsplit - 1) / nsplit);
    p.unroll = tuning().unroll > 0 ? tuning().unroll : best_unroll;
    return p;
}

template <int BPP>
struct Acc;

template <>
struct Acc<2> {
    static constexpr int PX = 8;
    uint32_t sum[8];
    uint32_t mx[4];   // packed u16 pairs
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int i = 0; i < 8; ++i) sum[i] = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) mx[i] = 0;
    }
    __device__ __forceinline__ void add_dword(int i, uint32_t d) {
        sum[2 * i] += d & 0xffffu;
        sum[2 * i + 1] += d >> 16;
        ushort2_t a = __builtin_bit_cast(ushort2_t, mx[i]);
        ushort2_t b = __builtin_bit_cast(ushort2_t, d);
        mx[i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(a, b));   // v_pk_max_u16
    }
    __device__ __forceinline__ void add(const u32x4& r) {
        add_dword(0, r.x); add_dword(1, r.y); add_dword(2, r.z); add_dword(3, r.w);
    }
    __device__ __forceinline__ void store(uint32_t* ps, uint16_t* pm) const {
        uint4* s4 = reinterpret_cast<uint4*>(ps);
        s4[0] = make_uint4(sum[0], sum[1], sum[2], sum[3]);
        s4[1] = make_uint4(sum[4], sum[5], sum[6], sum[7]);
        *reinterpret_cast<uint4*>(pm) = make_uint4(mx[0], mx[1], mx[2], mx[3]);
    }
};

template <>
struct Acc<1> {
    // 16 bytes per lane.  Bytes are widened two at a time into packed u16 pairs (even bytes / odd bytes of a dword),
    // so a frame costs 2 masks + 2 v_add_u32 + 2 v_pk_max_u16 per 4 pixels instead of 12 scalar-per-byte ops.  A u16
    // running sum holds 257 full-scale frames: it is folded into the u32 totals every 256 frames.
    static constexpr int PX = 16;
    uint32_t sum[16];
    uint32_t s16[8];      // s16[2i]: bytes 0, 2 of dword i; s16[2i+1]: bytes 1, 3
    uint32_t m16[8];      // maxima, same pairing
    int pending;
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int i = 0; i < 16; ++i) sum[i] = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { s16[i] = 0; m16[i] = 0; }
        pending = 0;
    }
    __device__ __forceinline__ void flush() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sum[4 * i] += s16[2 * i] & 0xffffu;
            sum[4 * i + 2] += s16[2 * i] >> 16;
            sum[4 * i + 1] += s16[2 * i + 1] & 0xffffu;
            sum[4 * i + 3] += s16[2 * i + 1] >> 16;
            s16[2 * i] = 0;
            s16[2 * i + 1] = 0;
        }
        pending = 0;
    }
    __device__ __forceinline__ void add_dword(int i, uint32_t d) {
        const uint32_t e = d & 0x00ff00ffu, o = (d >> 8) & 0x00ff00ffu;
        s16[2 * i] += e;                 // no carry between the halves: each stays below 65536 until the flush
        s16[2 * i + 1] += o;
        m16[2 * i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(ushort2_t, m16[2 * i]),
                                                                            __builtin_bit_cast(ushort2_t, e)));
        m16[2 * i + 1] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(ushort2_t, m16[2 * i + 1]),
                                                                                __builtin_bit_cast(ushort2_t, o)));
    }
    __device__ __forceinline__ void add(const u32x4& r) {
        add_dword(0, r.x); add_dword(1, r.y); add_dword(2, r.z); add_dword(3, r.w);
        if (++pending == 256) flush();
    }
    __device__ __forceinline__ void store(uint32_t* ps, uint16_t* pm) {
        flush();
        uint4* s4 = reinterpret_cast<uint4*>(ps);
#pragma unroll
        for (int i = 0; i < 4; ++i) s4[i] = make_uint4(sum[4 * i], sum[4 * i + 1], sum[4 * i + 2], sum[4 * i + 3]);
        // pixel 4i+b of dword i: b = 0, 2 in m16[2i] (low, high), b = 1, 3 in m16[2i+1]
        uint32_t w[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            w[2 * i] = (m16[2 * i] & 0xffffu) | (m16[2 * i + 1] << 16);
            w[2 * i + 1] = (m16[2 * i] >> 16) | (m16[2 * i + 1] & 0xffff0000u);
        }
        uint4* m4 = reinterpret_cast<uint4*>(pm);
        m4[0] = make_uint4(w[0], w[1], w[2], w[3]);
        m4[1] = make_uint4(w[4], w[5], w[6], w[7]);
    }
};

// grid: (ceil(vecs/256), nsplit).  One lane = 16 bytes of the frame, all frames of its split.
template <int BPP, int UNROLL, bool NT>
__global__ __launch_bounds__(256) void k_accumulate_vec(const u32x4* __restrict__ stack, int64_t vecs, int64_t fstride,
                                                        int n_frames, int frames_per_split,
                                                        uint32_t* __restrict__ psum, uint16_t* __restrict__ pmax,
                                                        int64_t npix, int nsplit, int xcd_per_split, int prio, int interleave) {
    // Column block and frame split of this workgroup.  Plain mapping: (blockIdx.x, blockIdx.y).  XCD-aware mapping
    // (1-D grid): workgroups b and b + 8 share an XCD under round-robin dispatch, so giving the XCDs with
    // (b % 8) / xcd_per_split == s split s keeps every XCD on one frame range (tried for TLB / DRAM-page locality).
    int64_t cb = blockIdx.x;
    int split = blockIdx.y;
    if (xcd_per_split > 0) {
        const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
        split = (int)(xcd / (unsigned)xcd_per_split);
        cb = (int64_t)slot * xcd_per_split + (xcd % (unsigned)xcd_per_split);
        if (split >= nsplit) return;
    }
    const int64_t v = cb * 256 + threadIdx.x;
    if (v >= vecs) return;
    // this kernel's waves ahead of the other scans' at the CU's issue arbiter (SHG_ACC_PRIO=0: not; measured 0.379-0.381 against 0.387-0.390 ms per step)
    if (prio > 0) __builtin_amdgcn_s_setprio(3);
    // A split is a range of consecutive frames -- or (interleave, a tuning experiment) every nsplit-th frame: then all workgroups read
    // the same few frames at any moment however many splits there are, instead of nsplit regions of the stack far apart.
    const int kstep = interleave ? nsplit : 1;
    const int k0 = interleave ? split : split * frames_per_split;
    const int k1 = interleave ? n_frames : min(n_frames, k0 + frames_per_split);
    const int64_t fs = fstride * kstep;
    Acc<BPP> acc;
    acc.init();
    const u32x4* p = stack + (int64_t)k0 * fstride + v;
    int k = k0;
    for (; k + (UNROLL - 1) * kstep < k1; k += UNROLL * kstep) {
        u32x4 r[UNROLL];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) r[j] = NT ? __builtin_nontemporal_load(p + (int64_t)j * fs) : p[(int64_t)j * fs];
        p += (int64_t)UNROLL * fs;
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) acc.add(r[j]);
    }
    for (; k < k1; k += kstep) {
        acc.add(NT ? __builtin_nontemporal_load(p) : *p);
        p += fs;
    }
    const int64_t pix = v * Acc<BPP>::PX;
    acc.store(psum + (int64_t)split * npix + pix, pmax + (int64_t)split * npix + pix);
}

// Generic path (frame size not a multiple of 16 bytes, or unaligned base): one lane per pixel.
template <typename T>
__global__ __launch_bounds__(256) void k_accumulate_scalar(const T* __restrict__ stack, int64_t npix, int64_t fstride, int n_frames,
                                                           int frames_per_split, uint32_t* __restrict__ psum,
                                                           uint16_t* __restrict__ pmax) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    const int split = blockIdx.y;
    const int k0 = split * frames_per_split;
    const int k1 = min(n_frames, k0 + frames_per_split);
    uint32_t s = 0, m = 0;
    for (int k = k0; k < k1; ++k) {
        const uint32_t v = stack[(int64_t)k * fstride + i];
        s += v;
        m = m > v ? m : v;
    }
    psum[(int64_t)split * npix + i] = s;
    pmax[(int64_t)split * npix + i] = (uint16_t)m;
}

__global__ __launch_bounds__(256) void k_reduce_partials(const uint32_t* __restrict__ psum, const uint16_t* __restrict__ pmax,
                                                         int nsplit, int64_t npix, uint64_t* __restrict__ sum_out,
                                                         uint16_t* __restrict__ max_out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    uint64_t s = 0;
    uint32_t m = 0;
    for (int j = 0; j < nsplit; ++j) {
        s += psum[(int64_t)j * npix + i];
        const uint32_t v = pmax[(int64_t)j * npix + i];
        m = m > v ? m : v;
    }
    sum_out[i] = s;
    max_out[i] = (uint16_t)m;
}

// Where the finalising kernels take a pixel's total and maximum from: the reduced arrays (a sharded scan: after the
// all-reduce) or pass A's per-slab partials directly (one launch and one array round trip less for a scan on one GPU).
struct FromSums {
    const uint64_t* sum;
    const uint16_t* mx;
    __device__ __forceinline__ void get(int64_t i, uint64_t& s, uint32_t& m) const { s = sum[i]; m = mx[i]; }
};
struct FromPartials {
    const uint32_t* psum;
    const uint16_t* pmax;
    int nsplit;
    int64_t npix;
    __device__ __forceinline__ void get(int64_t i, uint64_t& s, uint32_t& m) const {
        s = 0;
        m = 0;
        for (int j = 0; j < nsplit; ++j) {
            s += psum[(int64_t)j * npix + i];
            const uint32_t v = pmax[(int64_t)j * npix + i];
            m = m > v ? m : v;
        }
    }
};

// mean = sum // n (== trunc(float64(sum)/n) for sums < 2^53, solex_util.py:188), x256 for 8-bit
// (video_reader.py:121-122), rotated: out[y][x] = in[x][W-1-y] when W > H (video_reader.py:119-120).
template <typename Src>
struct FinalizeArgs {
    Src in;
    uint64_t n_total;
    int64_t height, width;
    int scale;
    uint16_t *mean_out, *max_out;
};

template <typename Src> __global__ __launch_bounds__(256) void k_finalize(const FinalizeArgs<Src> kargs) {
    const Src in = kargs.in;
    const uint64_t n_total = kargs.n_total;
    const int64_t height = kargs.height, width = kargs.width;
    const int scale = kargs.scale;
    uint16_t* __restrict__ mean_out = kargs.mean_out;
    uint16_t* __restrict__ max_out = kargs.max_out;
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= height * width) return;
    int64_t src;
    if (width > height) {
        const int64_t iw = height;
        const int64_t y = o / iw, x = o - y * iw;
        src = x * width + (width - 1 - y);
    } else {
        src = o;
    }
    uint64_t sv;
    uint32_t mv;
    in.get(src, sv, mv);
    mean_out[o] = (uint16_t)((sv * (uint64_t)scale) / n_total);
    max_out[o] = (uint16_t)(mv * scale);
}

// The rotated case through a 32 x 32 LDS tile: the plain kernel reads `sum` along file columns (one 8-byte value per 16 KB
// of addresses: 15.6 MB fetched for a 4 MB input at C2, PMC round 1).  Here a workgroup reads 32 file rows x 32 file
// columns row-wise (256-byte runs) and writes 32 output rows x 32 output columns row-wise.
template <typename Src> __global__ __launch_bounds__(256) void k_finalize_rot(const FinalizeArgs<Src> kargs) {
    const Src in = kargs.in;
    const uint64_t n_total = kargs.n_total;
    const int64_t height = kargs.height, width = kargs.width;
    const int scale = kargs.scale;
    uint16_t* __restrict__ mean_out = kargs.mean_out;
    uint16_t* __restrict__ max_out = kargs.max_out;
    __shared__ uint16_t tm[32][33], tx[32][33];
    const int64_t fx0 = (int64_t)blockIdx.x * 32;          // file column block  (slit rows y = W-1-fx, descending)
    const int64_t fy0 = (int64_t)blockIdx.y * 32;          // file row block     (spectral columns x = fy)
    const int tx_ = threadIdx.x & 31, ty_ = threadIdx.x >> 5;      // 32 x 8
    for (int r = ty_; r < 32; r += 8) {
        const int64_t fy = fy0 + r, fx = fx0 + tx_;
        if (fy < height && fx < width) {
            uint64_t sv;
            uint32_t mv;
            in.get(fy * width + fx, sv, mv);
            tm[r][tx_] = (uint16_t)((sv * (uint64_t)scale) / n_total);
            tx[r][tx_] = (uint16_t)(mv * scale);
        }
    }
    __syncthreads();
    // out[y][x] = in[x][W-1-y]: output row y <-> file column W-1-y, output column x <-> file row x
    for (int r = ty_; r < 32; r += 8) {
        const int64_t fx = fx0 + r, fy = fy0 + tx_;
        if (fx < width && fy < height) {
            const int64_t y = width - 1 - fx;
            mean_out[y * height + fy] = tm[tx_][r];
            max_out[y * height + fy] = tx[tx_][r];
        }
    }
}

template <int BPP, bool NT>
void launch_vec(const Plan& p, const void* stack, int n, uint32_t* psum, uint16_t* pmax, hipStream_t st) {
    const int64_t nblk = (p.vecs + 255) / 256;
    dim3 grid((unsigned)nblk, (unsigned)p.nsplit);
    int xcd_per = 0;
    if (tuning().xcd && p.nsplit <= 8 && 8 % p.nsplit == 0) {      // tuning experiment, off by default
        xcd_per = 8 / p.nsplit;
        grid = dim3((unsigned)(8 * ((nblk + xcd_per - 1) / xcd_per)), 1u);
    }
    const u32x4* s = static_cast<const u32x4*>(stack);
    // SHG_ACC_LDS_KIB (tuning experiment): unused dynamic LDS per workgroup, which caps the workgroups a CU holds at once -- with more,
    // shorter workgroups than the device can hold (a larger nsplit) the hardware then hands them out as earlier ones finish: CUs
    // that the other scans' kernels keep busy simply take fewer of them
    const size_t pad_lds = (size_t)std::min(std::max(tuning().lds_kib, 0), 160) * 1024;
#define SHG_ACC_LAUNCH(U) { SHG_PROF("accumulate", st);                                                                                    \
        if (pad_lds > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_accumulate_vec<BPP, U, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad_lds); \
        k_accumulate_vec<BPP, U, NT><<<grid, 256, pad_lds, st>>>(s, p.vecs, p.stride_vecs, n, p.frames_per_split, psum, pmax, p.npix, p.nsplit, xcd_per, tuning().prio, tuning().interleave); }
    switch (p.unroll) {
        case 2: SHG_ACC_LAUNCH(2) break;
        case 4: SHG_ACC_LAUNCH(4) break;
        case 16: SHG_ACC_LAUNCH(16) break;
        default: SHG_ACC_LAUNCH(8) break;
    }
#undef SHG_ACC_LAUNCH
}

size_t slab_bytes(const Plan& p) {
    // u32 sums then u16 maxima; npix*4*nsplit is 16-byte aligned on the vector path
    return (size_t)p.nsplit * (size_t)p.npix * 6 + 64;
}

}  // namespace

extern "C" size_t shg_accumulate_workspace_bytes(int64_t n_frames, int64_t height, int64_t width, int bytes_per_px) {
    if (n_frames <= 0 || height <= 0 || width <= 0 || (bytes_per_px != 1 && bytes_per_px != 2)) return 0;
    return slab_bytes(make_plan(nullptr, n_frames, height, width, bytes_per_px, 0));
}

namespace {
// pass A into the per-slab partials at the head of `workspace`: psum u32 [nsplit][npix], pmax u16 [nsplit][npix]
int accumulate_partials(const void* stack, int64_t n_frames, int64_t height, int64_t width, int bytes_per_px, int64_t frame_stride_px,
                        void* workspace, size_t workspace_bytes, shg_stream_t stream, Plan* plan_out, uint32_t** psum_out, uint16_t** pmax_out) {
    SHG_REQUIRE(stack && workspace, SHG_E_ARG, "shg_accumulate_sum_max: null pointer");
    SHG_REQUIRE(n_frames > 0 && height > 0 && width > 0, SHG_E_ARG, "shg_accumulate_sum_max: empty stack (%lld x %lld x %lld)",
                (long long)n_frames, (long long)height, (long long)width);
    SHG_REQUIRE(bytes_per_px == 1 || bytes_per_px == 2, SHG_E_ARG, "shg_accumulate_sum_max: bytes_per_px must be 1 or 2");
    SHG_REQUIRE(n_frames < (1ll << 31), SHG_E_UNSUPPORTED, "shg_accumulate_sum_max: too many frames");
    SHG_REQUIRE(frame_stride_px == 0 || frame_stride_px >= height * width, SHG_E_ARG, "shg_accumulate_sum_max: frame stride smaller than a frame");
    Plan p = make_plan(stack, n_frames, height, width, bytes_per_px, frame_stride_px);
    // a u32 partial holds 65537 frames of 16-bit samples
    SHG_REQUIRE(p.frames_per_split <= 65537, SHG_E_UNSUPPORTED, "shg_accumulate_sum_max: %d frames per split overflow u32",
                p.frames_per_split);
    SHG_REQUIRE(workspace_bytes >= slab_bytes(p), SHG_E_WORKSPACE, "shg_accumulate_sum_max: workspace %zu < %zu bytes",
                workspace_bytes, slab_bytes(p));
    SHG_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, SHG_E_ARG, "shg_accumulate_sum_max: workspace not 16-byte aligned");
    hipStream_t st = shg::as_stream(stream);
    uint32_t* psum = static_cast<uint32_t*>(workspace);
    uint16_t* pmax = reinterpret_cast<uint16_t*>(psum + (size_t)p.nsplit * p.npix);
    const int n = (int)n_frames;
    if (p.vector_path) {
        const bool nt = tuning().nt != 0;
        if (bytes_per_px == 2) { if (nt) launch_vec<2, true>(p, stack, n, psum, pmax, st); else launch_vec<2, false>(p, stack, n, psum, pmax, st); }
        else { if (nt) launch_vec<1, true>(p, stack, n, psum, pmax, st); else launch_vec<1, false>(p, stack, n, psum, pmax, st); }
    } else {
        dim3 grid((unsigned)((p.npix + 255) / 256), (unsigned)p.nsplit);
        if (bytes_per_px == 2)
            { SHG_PROF("accumulate", st); k_accumulate_scalar<uint16_t><<<grid, 256, 0, st>>>(static_cast<const uint16_t*>(stack), p.npix, p.stride_px, n, p.frames_per_split, psum, pmax); }
        else
            { SHG_PROF("accumulate", st); k_accumulate_scalar<uint8_t><<<grid, 256, 0, st>>>(static_cast<const uint8_t*>(stack), p.npix, p.stride_px, n, p.frames_per_split, psum, pmax); }
    }
    *plan_out = p;
    *psum_out = psum;
    *pmax_out = pmax;
    return shg::check_launch("k_accumulate");
}

template <typename Src>
int launch_finalize(Src in, int64_t n_total, int64_t height, int64_t width, int bytes_per_px, uint16_t* mean_out, uint16_t* max_out, hipStream_t st) {
    SHG_PROF("finalize", st);
    const int scale = bytes_per_px == 1 ? 256 : 1;
    const FinalizeArgs<Src> fa{in, (uint64_t)n_total, height, width, scale, mean_out, max_out};
    if (width > height) {
        dim3 grid((unsigned)((width + 31) / 32), (unsigned)((height + 31) / 32));
        return shg::launch(k_finalize_rot<Src>, grid, dim3(256), 0, st, fa, "k_finalize_rot");
    }
    return shg::launch(k_finalize<Src>, dim3((unsigned)((height * width + 255) / 256)), dim3(256), 0, st, fa, "k_finalize");
}
}  // namespace

extern "C" int shg_accumulate_sum_max(const void* stack, int64_t n_frames, int64_t height, int64_t width,
                                      int bytes_per_px, int64_t frame_stride_px, uint64_t* sum_out, uint16_t* max_out,
                                      void* workspace, size_t workspace_bytes, shg_stream_t stream) {
    SHG_REQUIRE(sum_out && max_out, SHG_E_ARG, "shg_accumulate_sum_max: null pointer");
    Plan p;
    uint32_t* psum;
    uint16_t* pmax;
    if (int e = accumulate_partials(stack, n_frames, height, width, bytes_per_px, frame_stride_px, workspace, workspace_bytes, stream, &p, &psum, &pmax)) return e;
    hipStream_t st = shg::as_stream(stream);
    { SHG_PROF("reduce_partials", st); k_reduce_partials<<<(unsigned)((p.npix + 255) / 256), 256, 0, st>>>(psum, pmax, p.nsplit, p.npix, sum_out, max_out); }
    return shg::check_launch("k_reduce_partials");
}

namespace {
struct PassA {
    const void* stack;
    int64_t n_frames, height, width;
    int bytes_per_px;
    int64_t frame_stride_px;
    void* workspace;
    size_t workspace_bytes;
    Plan p;
    uint32_t* psum;
    uint16_t* pmax;
};
int launch_pass_a(hipStream_t st, void* arg) {
    PassA* a = static_cast<PassA*>(arg);
    return accumulate_partials(a->stack, a->n_frames, a->height, a->width, a->bytes_per_px, a->frame_stride_px, a->workspace, a->workspace_bytes,
                               reinterpret_cast<shg_stream_t>(st), &a->p, &a->psum, &a->pmax);
}
}  // namespace

// A pass launched ahead of its scan (shg_pass_a_prelaunch), keyed by the workspace its partials went to.
namespace {
struct Ahead {
    PassA a;
    hipEvent_t done;
};
std::mutex g_ahead_mu;
std::unordered_map<const void*, Ahead> g_ahead;

bool same_pass(const PassA& x, const PassA& y) {
    return x.stack == y.stack && x.n_frames == y.n_frames && x.height == y.height && x.width == y.width && x.bytes_per_px == y.bytes_per_px &&
           x.frame_stride_px == y.frame_stride_px && x.workspace == y.workspace;
}

// -> true when a pass into `workspace` had been launched ahead; it has finished when this returns (whatever it was a pass over:
// a pass that is not ours still writes to our workspace, so it is waited for all the same) and *same says whether it is the
// pass `want` asks for, with its plan and partials copied into `want`.
bool take_ahead(PassA* want, bool* same, int* status) {
    Ahead got;
    {
        std::lock_guard<std::mutex> lk(g_ahead_mu);
        auto it = g_ahead.find(want->workspace);
        if (it == g_ahead.end()) return false;
        got = it->second;
        g_ahead.erase(it);
    }
    hipError_t e;
    {
        SHG_HOST_TIME("lane wait (queue + pass A)");
        e = hipEventSynchronize(got.done);
    }
    (void)hipEventDestroy(got.done);
    *status = 0;
    if (e != hipSuccess) { shg::set_error("frame-pass lane: %s", hipGetErrorString(e)); *status = (int)e; }
    *same = same_pass(got.a, *want) && got.a.workspace_bytes <= want->workspace_bytes;
    if (*same) { want->p = got.a.p; want->psum = got.a.psum; want->pmax = got.a.pmax; }
    return true;
}
}  // namespace

// Start pass A of a scan that has not begun yet (the scan pool does, for the scans in its queue: pool.hip).  The lane then
// never idles between the scans' passes while a worker is still busy with the chain of an earlier scan -- the pass is the one
// kernel of a scan that is bound by the device, and it is what the rate of a batch is bound by.  shg_accumulate_mean_max with the
// same arguments later finds the pass here, waits for it and only finalises.  Needs a lane (-> *launched = 0 without one).
extern "C" int shg_pass_a_prelaunch(const void* stack, int64_t n_frames, int64_t height, int64_t width, int bytes_per_px,
                                    int64_t frame_stride_px, void* workspace, size_t workspace_bytes, shg_stream_t after, int* launched) {
    SHG_REQUIRE(launched, SHG_E_ARG, "shg_pass_a_prelaunch: null pointer");
    *launched = 0;
    PassA a{stack, n_frames, height, width, bytes_per_px, frame_stride_px, workspace, workspace_bytes, Plan{}, nullptr, nullptr};
    {   // a pass somebody started into this workspace and never used: let it finish before this one writes there
        bool same = false;
        int status = 0;
        PassA old = a;
        if (take_ahead(&old, &same, &status) && status) return status;
    }
    hipEvent_t done = nullptr;
    const int r = shg::prelaunch_on_lane(shg::as_stream(after), launch_pass_a, &a, &done);
    if (r == 1) return 0;                                      // no lane: the scan launches its pass itself
    if (r != 0) return r;
    std::lock_guard<std::mutex> lk(g_ahead_mu);
    g_ahead[workspace] = Ahead{a, done};
    *launched = 1;
    return 0;
}

// Wait for and drop a pass launched ahead into `workspace` that no scan came to use (a scan that failed before its first stage).
extern "C" int shg_pass_a_forget(const void* workspace) {
    PassA a{};
    a.workspace = const_cast<void*>(workspace);
    bool same = false;
    int status = 0;
    (void)take_ahead(&a, &same, &status);
    return status;
}

// Pass A goes through the frame-pass lane when the process has one (streams.hip): the passes of all scans in flight run
// one after the other there instead of halving each other's bandwidth.  Only the pass itself: its partials live in the
// caller's workspace, which no other stream touches, while mean_out / max_out may be memory the caller's stream has
// only just let go of -- the finalising kernel stays on `stream`, behind the lane's event.
extern "C" int shg_accumulate_mean_max(const void* stack, int64_t n_frames, int64_t height, int64_t width, int bytes_per_px,
                                       int64_t frame_stride_px, uint16_t* mean_out, uint16_t* max_out, void* workspace,
                                       size_t workspace_bytes, shg_stream_t stream) {
    SHG_REQUIRE(mean_out && max_out, SHG_E_ARG, "shg_accumulate_mean_max: null pointer");
    PassA a{stack, n_frames, height, width, bytes_per_px, frame_stride_px, workspace, workspace_bytes, Plan{}, nullptr, nullptr};
    bool same = false;
    int status = 0;
    const bool ahead = take_ahead(&a, &same, &status);
    if (ahead && status) return status;
    if (!(ahead && same))
        if (int e = shg::on_frame_pass_lane(shg::as_stream(stream), launch_pass_a, &a)) return e;
    return launch_finalize(FromPartials{a.psum, a.pmax, a.p.nsplit, a.p.npix}, n_frames, height, width, bytes_per_px, mean_out, max_out, shg::as_stream(stream));
}

// ---- the frame statistics of a sharded scan: the G ranks' pieces, as one all-gather brought them, folded in one launch -------------
// pieces: n_pieces records of piece_words 32-bit words each: [npix partial sums as 32-bit words | npix partial maxima as 16-bit
// words (padded to a whole word) | anything else] -- dist.exchange_frame_stats' message.  -> sum_out [npix] (64 bits), max_out [npix].
namespace {
struct ReducePiecesArgs {
    const uint32_t* pieces;
    int n_pieces;
    int64_t piece_words, npix;
    uint64_t* sum_out;
    uint16_t* max_out;
};
__global__ __launch_bounds__(256) void k_reduce_frame_stats(const ReducePiecesArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.npix) return;
    uint64_t s = 0;
    uint32_t m = 0;
    for (int j = 0; j < a.n_pieces; ++j) {
        const uint32_t* piece = a.pieces + (int64_t)j * a.piece_words;
        s += piece[i];
        const uint32_t v = reinterpret_cast<const uint16_t*>(piece + a.npix)[i];
        m = m > v ? m : v;
    }
    a.sum_out[i] = s;
    a.max_out[i] = (uint16_t)m;
}
}  // namespace

extern "C" int shg_reduce_frame_stats(const uint32_t* pieces, int n_pieces, int64_t piece_words, int64_t npix, uint64_t* sum_out,
                                      uint16_t* max_out, shg_stream_t stream) {
    SHG_REQUIRE(pieces && sum_out && max_out, SHG_E_ARG, "shg_reduce_frame_stats: null pointer");
    SHG_REQUIRE(n_pieces > 0 && npix > 0 && piece_words >= npix + (npix + 1) / 2, SHG_E_ARG, "shg_reduce_frame_stats: %d pieces of %lld words for %lld pixels",
                n_pieces, (long long)piece_words, (long long)npix);
    return shg::launch(k_reduce_frame_stats, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, shg::as_stream(stream),
                       ReducePiecesArgs{pieces, n_pieces, piece_words, npix, sum_out, max_out}, "k_reduce_frame_stats");
}

extern "C" int shg_finalize_mean_max(const uint64_t* sum, const uint16_t* max_raw, int64_t n_total,
                                     int64_t height, int64_t width, int bytes_per_px,
                                     uint16_t* mean_out, uint16_t* max_out, shg_stream_t stream) {
    SHG_REQUIRE(sum && max_raw && mean_out && max_out, SHG_E_ARG, "shg_finalize_mean_max: null pointer");
    SHG_REQUIRE(n_total > 0 && height > 0 && width > 0, SHG_E_ARG, "shg_finalize_mean_max: bad size");
    SHG_REQUIRE(bytes_per_px == 1 || bytes_per_px == 2, SHG_E_ARG, "shg_finalize_mean_max: bytes_per_px must be 1 or 2");
    return launch_finalize(FromSums{sum, max_raw}, n_total, height, width, bytes_per_px, mean_out, max_out, shg::as_stream(stream));
}
