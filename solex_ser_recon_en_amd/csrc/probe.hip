// A trivial streaming-read kernel: the measured HBM read ceiling that bench.py quotes beside the
// 8 TB/s spec peak (SURVEY.md section 8d asks for both).  Every lane XOR-folds 16-byte non-temporal
// loads of a grid-strided sweep; one word per workgroup is stored so that the loads stay live.
#include "shg_common.h"

namespace {
typedef unsigned int __attribute__((ext_vector_type(4))) u32x4;

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

extern "C" int shg_stream_read_probe(const void* buf, int64_t bytes, int blocks, int unroll, uint32_t* out1024, shg_stream_t stream) {
    SHG_REQUIRE(buf && out1024, SHG_E_ARG, "shg_stream_read_probe: null pointer");
    SHG_REQUIRE(bytes >= 16 && blocks > 0 && (reinterpret_cast<uintptr_t>(buf) & 15) == 0, SHG_E_ARG, "shg_stream_read_probe: bad arguments");
    hipStream_t st = shg::as_stream(stream);
    const u32x4* p = static_cast<const u32x4*>(buf);
    const int64_t n = bytes / 16;
    SHG_PROF("stream_probe", st);
    switch (unroll) {
        case 1: k_stream_probe<1><<<blocks, 256, 0, st>>>(p, n, out1024); break;
        case 2: k_stream_probe<2><<<blocks, 256, 0, st>>>(p, n, out1024); break;
        case 8: k_stream_probe<8><<<blocks, 256, 0, st>>>(p, n, out1024); break;
        default: k_stream_probe<4><<<blocks, 256, 0, st>>>(p, n, out1024); break;
    }
    return shg::check_launch("k_stream_probe");
}
