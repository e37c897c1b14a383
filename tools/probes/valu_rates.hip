// How many cycles does a SIMD of gfx950 spend on one wave64 instruction of the kinds this library's float64 kernels are made of?
// (k_rowpair_stats, k_warp_rows, k_extract, k_products8 and the fused CLAHE histogram kernel are bound by their VALU instructions:
// which ones to trade for which is decided by these numbers, not by a guess.)  Every workgroup is ONE wave (one SIMD busy per CU
// slot, no sharing), each lane runs UNROLL independent chains of the instruction N times; cycles from s_memtime around the loop.
//     hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates tools/probes/valu_rates.hip && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int N = 2048, UNROLL = 8;

#define STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")

// one kernel per instruction: OP(i) is an asm statement working on the i-th of UNROLL independent registers
#define RATE_KERNEL(name, DECL, OP, SINK)                                                           \
    __global__ __launch_bounds__(64) void name(unsigned long long* out, double seed) {              \
        DECL;                                                                                       \
        unsigned long long t0, t1;                                                                  \
        STAMP(t0);                                                                                  \
        for (int n = 0; n < N; ++n) {                                                               \
            _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) { OP; }                              \
        }                                                                                           \
        STAMP(t1);                                                                                  \
        SINK;                                                                                       \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                            \
    }

#define DECL_D double a[UNROLL], b = seed, c = seed * 0.5; _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) a[i] = seed + i + threadIdx.x
#define SINK_D double s_ = 0; _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) s_ += a[i]; if (s_ == 12345.678) out[1000] = 1
#define DECL_U uint32_t u[UNROLL]; double a[UNROLL]; _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) { u[i] = threadIdx.x + i; a[i] = seed + i; }
#define SINK_U double s_ = 0; _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) s_ += a[i] + u[i]; if (s_ == 12345.678) out[1000] = 1

RATE_KERNEL(k_fma_f64, DECL_D, asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)), SINK_D)
RATE_KERNEL(k_mul_f64, DECL_D, asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b)), SINK_D)
RATE_KERNEL(k_add_f64, DECL_D, asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b)), SINK_D)
RATE_KERNEL(k_min_f64, DECL_D, asm volatile("v_min_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b)), SINK_D)
RATE_KERNEL(k_floor_f64, DECL_D, asm volatile("v_floor_f64 %0, %0" : "+v"(a[i])), SINK_D)
RATE_KERNEL(k_rcp_f64, DECL_D, asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i])), SINK_D)
RATE_KERNEL(k_cmp_f64, DECL_D, asm volatile("v_cmp_lt_f64 vcc, %0, %1" ::"v"(a[i]), "v"(b) : "vcc"), SINK_D)
RATE_KERNEL(k_cvt_f64_u32, DECL_U, asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(a[i]) : "v"(u[i])), SINK_U)
RATE_KERNEL(k_cvt_i32_f64, DECL_U, asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(u[i]) : "v"(a[i])), SINK_U)
RATE_KERNEL(k_cvt_f32_f64, DECL_U, asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(u[i]) : "v"(a[i])), SINK_U)
RATE_KERNEL(k_add_u32, DECL_U, asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % UNROLL])), SINK_U)
RATE_KERNEL(k_and_b32, DECL_U, asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % UNROLL])), SINK_U)
RATE_KERNEL(k_mul_lo_u32, DECL_U, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % UNROLL])), SINK_U)
RATE_KERNEL(k_fma_f32, DECL_U, asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(u[(i + 1) % UNROLL])), SINK_U)
RATE_KERNEL(k_cndmask, DECL_U, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(u[(i + 1) % UNROLL])), SINK_U)
RATE_KERNEL(k_lshl_add_u64, DECL_D, asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b)), SINK_D)

template <typename K>
void run(const char* what, K kernel, unsigned long long* d, int waves_per_simd) {
    // 256 CUs x 4 SIMDs x waves_per_simd single-wave workgroups (the dispatcher spreads them; a few land together: the median is reported)
    const int blocks = 256 * 4 * waves_per_simd;
    unsigned long long* h = (unsigned long long*)malloc(sizeof(unsigned long long) * blocks);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(64), 0, 0, d, 1.000001);
        CHECK(hipDeviceSynchronize());
    }
    CHECK(hipMemcpy(h, d, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost));
    unsigned long long best = ~0ull, sum = 0;
    for (int i = 0; i < blocks; ++i) { best = h[i] < best ? h[i] : best; sum += h[i]; }
    printf("%-28s %d wave(s) a SIMD: %6.2f cycles an instruction (fastest wave), %6.2f (mean) -> x %d waves = %6.2f SIMD cycles per wave-instruction\n", what,
           waves_per_simd, (double)best / (N * UNROLL), (double)sum / blocks / (N * UNROLL), waves_per_simd,
           (double)sum / blocks / (N * UNROLL) / waves_per_simd);
    free(h);
}

int main() {
    unsigned long long* d;
    CHECK(hipMalloc(&d, sizeof(unsigned long long) * 16384));
#define RUN(k) run(#k, k, d, 1); run(#k, k, d, 4)
    RUN(k_fma_f64); RUN(k_mul_f64); RUN(k_add_f64); RUN(k_min_f64); RUN(k_floor_f64); RUN(k_rcp_f64); RUN(k_cmp_f64);
    RUN(k_cvt_f64_u32); RUN(k_cvt_i32_f64); RUN(k_cvt_f32_f64); RUN(k_add_u32); RUN(k_and_b32); RUN(k_mul_lo_u32); RUN(k_fma_f32);
    RUN(k_cndmask); RUN(k_lshl_add_u64);
    return 0;
}
