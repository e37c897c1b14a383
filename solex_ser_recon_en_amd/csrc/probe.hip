// Trivial streaming-read kernels: the measured HBM read ceilings that bench.py quotes beside the
// 8 TB/s spec peak (SURVEY.md section 8d asks for both).
//   mode 0  a grid-strided contiguous sweep, 16-byte non-temporal loads XOR-folded (the device's read ceiling)
//   mode 1  the same sweep with pass A's per-vector arithmetic (shows the arithmetic is not what binds)
//   mode 2  pass A's addresses -- a lane owns a 16-byte column of the frame and walks the frame axis --
//           with the XOR fold only (the ceiling of that access pattern; `blocks` = frame-axis splits)
// One word per workgroup is stored so that the loads stay live.
#include "shg_common.h"

namespace {
typedef unsigned int __attribute__((ext_vector_type(4))) u32x4;

typedef unsigned short __attribute__((ext_vector_type(2))) ushort2_t;

// mode 1: the same sweep with pass A's per-vector arithmetic (8 u32 sums + 4 packed u16 maxima)
template <int UNROLL>
__global__ __launch_bounds__(256) void k_stream_probe_heavy(const u32x4* __restrict__ p, int64_t n_vecs, uint32_t* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mx[4] = {0, 0, 0, 0};
    auto fold = [&](const u32x4& r) {
        const uint32_t d[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            sum[2 * k] += d[k] & 0xffffu;
            sum[2 * k + 1] += d[k] >> 16;
            mx[k] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(ushort2_t, mx[k]), __builtin_bit_cast(ushort2_t, d[k])));
        }
    };
    for (; i + (UNROLL - 1) * stride < n_vecs; i += UNROLL * stride) {
        u32x4 r[UNROLL];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) r[j] = __builtin_nontemporal_load(p + i + j * stride);
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) fold(r[j]);
    }
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) v ^= sum[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) v ^= mx[k];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v ^= __shfl_xor(v, d);
    if ((threadIdx.x & 63) == 0) atomicXor(&out[blockIdx.x & 1023], v);
}

// mode 2: pass A's addresses (lane = fixed 16-byte column of the frame, walks the frame axis; grid.y frame splits)
template <int UNROLL>
__global__ __launch_bounds__(256) void k_stream_probe_frames(const u32x4* __restrict__ p, int64_t vecs, int n_frames, int fps,
                                                             uint32_t* __restrict__ out) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= vecs) return;
    const int k0 = blockIdx.y * fps, k1 = min(n_frames, k0 + fps);
    const u32x4* q = p + (int64_t)k0 * vecs + v;
    u32x4 acc = {0, 0, 0, 0};
    int k = k0;
    for (; k + UNROLL <= k1; k += UNROLL) {
        u32x4 r[UNROLL];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) r[j] = __builtin_nontemporal_load(q + (int64_t)j * vecs);
        q += (int64_t)UNROLL * vecs;
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) acc ^= r[j];
    }
    uint32_t x = acc.x ^ acc.y ^ acc.z ^ acc.w;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x ^= __shfl_xor(x, d);
    if ((threadIdx.x & 63) == 0) atomicXor(&out[blockIdx.x & 1023], x);
}

template <int UNROLL>
__global__ __launch_bounds__(256) void k_stream_probe(const u32x4* __restrict__ p, int64_t n_vecs, uint32_t* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    for (; i + (UNROLL - 1) * stride < n_vecs; i += UNROLL * stride) {
        u32x4 r[UNROLL];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) r[j] = __builtin_nontemporal_load(p + i + j * stride);
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) acc ^= r[j];
    }
    for (; i < n_vecs; i += stride) acc ^= __builtin_nontemporal_load(p + i);
    uint32_t v = acc.x ^ acc.y ^ acc.z ^ acc.w;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v ^= __shfl_xor(v, d);
    if ((threadIdx.x & 63) == 0) atomicXor(&out[blockIdx.x & 1023], v);
}
}  // namespace

extern "C" int shg_stream_read_probe(const void* buf, int64_t bytes, int mode, int blocks, int unroll, int64_t vecs_per_frame,
                                     uint32_t* out1024, shg_stream_t stream) {
    SHG_REQUIRE(buf && out1024, SHG_E_ARG, "shg_stream_read_probe: null pointer");
    SHG_REQUIRE(bytes >= 16 && blocks > 0 && (reinterpret_cast<uintptr_t>(buf) & 15) == 0, SHG_E_ARG, "shg_stream_read_probe: bad arguments");
    SHG_REQUIRE(mode >= 0 && mode <= 2 && (unroll == 1 || unroll == 2 || unroll == 4 || unroll == 8), SHG_E_ARG,
                "shg_stream_read_probe: mode must be 0..2 and unroll 1, 2, 4 or 8");
    hipStream_t st = shg::as_stream(stream);
    const u32x4* p = static_cast<const u32x4*>(buf);
    const int64_t n = bytes / 16;
    SHG_PROF("stream_probe", st);
#define SHG_PROBE(KERNEL, GRID, ...)                                                   \
    switch (unroll) {                                                                  \
        case 1: KERNEL<1><<<GRID, 256, 0, st>>>(__VA_ARGS__); break;                   \
        case 2: KERNEL<2><<<GRID, 256, 0, st>>>(__VA_ARGS__); break;                   \
        case 8: KERNEL<8><<<GRID, 256, 0, st>>>(__VA_ARGS__); break;                   \
        default: KERNEL<4><<<GRID, 256, 0, st>>>(__VA_ARGS__); break;                  \
    }
    if (mode == 2) {
        SHG_REQUIRE(vecs_per_frame > 0 && n >= vecs_per_frame, SHG_E_ARG, "shg_stream_read_probe: bad frame size");
        const int n_frames = (int)(n / vecs_per_frame), nsplit = blocks, fps = (n_frames + nsplit - 1) / nsplit;
        dim3 grid((unsigned)((vecs_per_frame + 255) / 256), (unsigned)nsplit);
        SHG_PROBE(k_stream_probe_frames, grid, p, vecs_per_frame, n_frames, fps, out1024)
    } else if (mode == 1) {
        SHG_PROBE(k_stream_probe_heavy, blocks, p, n, out1024)
    } else {
        SHG_PROBE(k_stream_probe, blocks, p, n, out1024)
    }
#undef SHG_PROBE
    return shg::check_launch("k_stream_probe");
}
