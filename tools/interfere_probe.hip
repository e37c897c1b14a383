// What kind of traffic beside a streaming read costs it most?  (tools/interference.py says which PARTS of a scan's chain slow pass A
// down; this says which KIND of memory access does.)  A 1.6 GB non-temporal read sweep -- pass A's traffic -- loops on a
// high-priority stream; one synthetic co-runner loops on another stream with a small grid; printed per co-runner: the sweep's
// duration beside it, the co-runner's own rate, and the sweep's time lost per MB the co-runner moves.
//     hipcc --offload-arch=gfx950 -O3 -o /tmp/interfere_probe tools/interfere_probe.hip && /tmp/interfere_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

typedef unsigned int __attribute__((ext_vector_type(4))) u32x4;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_sweep(const u32x4* __restrict__ p, int64_t n_vecs, uint32_t* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    for (; i + 3 * stride < n_vecs; i += 4 * stride) {
        u32x4 r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = __builtin_nontemporal_load(p + i + j * stride);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc ^= r[j];
    }
    uint32_t x = acc.x ^ acc.y ^ acc.z ^ acc.w;
    for (int d = 32; d >= 1; d >>= 1) x ^= __shfl_xor(x, d);
    if ((threadIdx.x & 63) == 0) atomicXor(&out[blockIdx.x & 1023], x);
}

// co-runners: `n_vecs` 16-byte vectors per launch, spread over the grid
enum Mode { READ = 0, READ_NT, WRITE, WRITE_NT, COPY, COPY_NT, WRITE_2B, WRITE_2B_STRIDED, ATOMIC_HOT, ATOMIC_SPREAD, SPIN_LOAD, LDS_ONLY, ALU_ONLY, READ_GATHER, N_MODES };
const char* kNames[N_MODES] = {"read 16 B", "read 16 B non-temporal", "write 16 B", "write 16 B non-temporal", "copy 16 B", "copy 16 B non-temporal",
                               "write 2 B (consecutive lanes)", "write 2 B (lanes 4 KB apart)", "atomicMax on one word", "atomicAdd on 64 Ki words", "load one word (glc) in a loop",
                               "LDS atomics only", "ALU only", "read 2 B (lanes 4 KB apart)"};

template <int MODE>
__global__ __launch_bounds__(256) void k_co(u32x4* __restrict__ dst, const u32x4* __restrict__ src, int64_t n_vecs, uint32_t* __restrict__ words, int iters) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    __shared__ uint32_t lds[1024];
    if (MODE == LDS_ONLY) { for (int j = threadIdx.x; j < 1024; j += 256) lds[j] = 0; __syncthreads(); }
    for (int64_t i = i0; i < n_vecs; i += stride) {
        if (MODE == READ) acc ^= src[i];
        if (MODE == READ_NT) acc ^= __builtin_nontemporal_load(src + i);
        if (MODE == WRITE) dst[i] = u32x4{(uint32_t)i, 1, 2, 3};
        if (MODE == WRITE_NT) __builtin_nontemporal_store(u32x4{(uint32_t)i, 1, 2, 3}, dst + i);
        if (MODE == COPY) dst[i] = src[i];
        if (MODE == COPY_NT) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
        if (MODE == WRITE_2B) reinterpret_cast<uint16_t*>(dst)[i] = (uint16_t)i;
        if (MODE == WRITE_2B_STRIDED) reinterpret_cast<uint16_t*>(dst)[((i & 63) * 2048 + (i >> 6)) % (n_vecs * 8)] = (uint16_t)i;
        if (MODE == READ_GATHER) acc.x ^= reinterpret_cast<const uint16_t*>(src)[((i & 63) * 2048 + (i >> 6)) % (n_vecs * 8)];
        if (MODE == ATOMIC_HOT) atomicMax(&words[0], (uint32_t)i);
        if (MODE == ATOMIC_SPREAD) atomicAdd(&words[(i * 2654435761u) & 65535], 1u);
        if (MODE == SPIN_LOAD) acc.x ^= __hip_atomic_load(&words[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == LDS_ONLY) { for (int j = 0; j < iters; ++j) atomicAdd(&lds[(threadIdx.x * 7 + j) & 1023], 1u); }
        if (MODE == ALU_ONLY) { double v = (double)i; for (int j = 0; j < iters; ++j) v = fma(v, 1.0000001, 0.5); acc.x ^= (uint32_t)v; }
    }
    if (MODE == LDS_ONLY) { __syncthreads(); acc.x ^= lds[threadIdx.x]; }
    uint32_t x = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (x == 0x12345678u) words[65536 + (threadIdx.x & 63)] = x;   // keeps the loads alive
}

template <int MODE>
void launch_co(hipStream_t st, int grid, u32x4* dst, const u32x4* src, int64_t n_vecs, uint32_t* words, int iters) {
    hipLaunchKernelGGL(k_co<MODE>, dim3(grid), dim3(256), 0, st, dst, src, n_vecs, words, iters);
}
typedef void (*CoFn)(hipStream_t, int, u32x4*, const u32x4*, int64_t, uint32_t*, int);
CoFn kCo[N_MODES] = {launch_co<0>, launch_co<1>, launch_co<2>, launch_co<3>, launch_co<4>, launch_co<5>, launch_co<6>, launch_co<7>,
                     launch_co<8>, launch_co<9>, launch_co<10>, launch_co<11>, launch_co<12>, launch_co<13>};

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 0.4;
    const int64_t sweep_bytes = 1600000000ll;
    const int64_t co_bytes_small = 16ll << 20, co_bytes_big = 1ll << 30;
    u32x4 *big, *co_src, *co_dst;
    uint32_t *out, *words;
    CK(hipMalloc(&big, sweep_bytes));
    CK(hipMemset(big, 1, sweep_bytes));
    CK(hipMalloc(&co_src, co_bytes_big));
    CK(hipMalloc(&co_dst, co_bytes_big));
    CK(hipMemset(co_src, 2, co_bytes_big));
    CK(hipMemset(co_dst, 0, co_bytes_big));
    CK(hipMalloc(&out, 4096));
    CK(hipMemset(out, 0, 4096));
    CK(hipMalloc(&words, 4 * (65536 + 64)));
    CK(hipMemset(words, 0, 4 * (65536 + 64)));
    int lo, hi;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, hi));
    CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, lo));
    hipEvent_t e0, e1, b0, b1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
    const int sweep_grid = 256 * 8;

    auto sweep_ms = [&](int reps) {
        CK(hipEventRecord(e0, sa));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_sweep, dim3(sweep_grid), dim3(256), 0, sa, big, sweep_bytes / 16, out);
        CK(hipEventRecord(e1, sa));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        return ms / reps;
    };
    sweep_ms(5);
    const double alone = sweep_ms(20);
    printf("sweep alone: %.1f us = %.2f TB/s\n", alone * 1e3, sweep_bytes / alone / 1e9);
    printf("%-34s %6s %5s %9s %9s %8s %9s %10s\n", "co-runner", "set", "grid", "sweep us", "x alone", "co GB/s", "co us", "lost us/MB");

    struct Case { int mode; int64_t bytes; int grid; int iters; };
    std::vector<Case> cases;
    for (int m : {READ, READ_NT, WRITE, WRITE_NT, COPY, COPY_NT})
        for (int64_t bytes : {co_bytes_small, co_bytes_big})
            for (int grid : {64, 512}) cases.push_back({m, bytes, grid, 0});
    for (int grid : {64, 512}) {
        cases.push_back({WRITE_2B, co_bytes_small, grid, 0});
        cases.push_back({WRITE_2B_STRIDED, co_bytes_small, grid, 0});
        cases.push_back({READ_GATHER, co_bytes_small, grid, 0});
        cases.push_back({ATOMIC_HOT, 1 << 20, grid, 0});
        cases.push_back({ATOMIC_SPREAD, 4 << 20, grid, 0});
        cases.push_back({SPIN_LOAD, 1 << 20, grid, 0});
        cases.push_back({LDS_ONLY, 1 << 20, grid, 64});
        cases.push_back({ALU_ONLY, 1 << 20, grid, 256});
    }
    cases.push_back({LDS_ONLY, 8 << 20, 2048, 64});
    cases.push_back({ALU_ONLY, 8 << 20, 2048, 256});

    for (const Case& c : cases) {
        const int64_t n_vecs = c.bytes / 16;
        // the co-runner alone
        for (int r = 0; r < 3; ++r) kCo[c.mode](sb, c.grid, co_dst, co_src, n_vecs, words, c.iters);
        CK(hipStreamSynchronize(sb));
        std::atomic<bool> stop{false};
        std::atomic<long> launches{0};
        std::thread feeder([&] {
            while (!stop.load()) {
                for (int r = 0; r < 4; ++r) kCo[c.mode](sb, c.grid, co_dst, co_src, n_vecs, words, c.iters);
                launches += 4;
                CK(hipStreamSynchronize(sb));
            }
        });
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
        const long l0 = launches.load();
        const auto t0 = std::chrono::steady_clock::now();
        const int reps = (int)(secs * 1e3 / alone) + 1;
        const double beside = sweep_ms(reps);
        const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const long l1 = launches.load();
        stop = true;
        feeder.join();
        const double co_launches_per_s = (l1 - l0) / wall;
        double moved = (double)c.bytes;                       // bytes of HBM-visible traffic per launch (algorithmic)
        if (c.mode == COPY || c.mode == COPY_NT) moved *= 2;
        if (c.mode == WRITE_2B || c.mode == WRITE_2B_STRIDED || c.mode == READ_GATHER) moved = (double)n_vecs * 2;
        if (c.mode >= ATOMIC_HOT && c.mode <= ALU_ONLY) moved = (double)n_vecs * 4;
        const double co_gbs = moved * co_launches_per_s / 1e9;
        const double lost_per_sweep_us = (beside - alone) * 1e3;
        const double mb_per_sweep = co_gbs * 1e3 * beside * 1e-3;      // MB the co-runner moved during one sweep
        printf("%-34s %6s %5d %9.1f %9.2f %8.0f %9.1f %10.3f\n", kNames[c.mode], c.bytes >= co_bytes_big ? "1 GB" : (c.bytes >= co_bytes_small ? "16 MB" : "small"),
               c.grid, beside * 1e3, beside / alone, co_gbs, co_launches_per_s > 0 ? 1e6 / co_launches_per_s : 0.0, mb_per_sweep > 0 ? lost_per_sweep_us / mb_per_sweep : 0.0);
        fflush(stdout);
    }
    return 0;
}
