// Pass B's gather with wider lanes -- the experiment before any kernel work (round-4 review, item 4).
// k_extract (csrc/extract.hip) gives a lane ONE slit row: on a rotated file a wave reads a 128-byte piece of a file row per
// (frame, shift, side), and the pieces of consecutive frames lie a frame (800 KB) apart.  The proposal: a lane owns R consecutive
// slit rows and loads 2 R bytes at once (R = 8: a 16-byte load, a wave reads 1 KiB of one file row).  This probe times exactly that
// access pattern over a C4-shaped stack -- [2000 frames][200 file rows][2000 columns] u16, frame pitch 802 816 bytes; 22 distinct
// file rows per frame (S = 21 consecutive shifts), read as (shifts / SC) groups of SC shifts x 2 sides like the kernel -- with NO
// arithmetic and no LDS tile: the loads are XORed together and one word per lane is stored.  Same launch shape as the kernel: a
// workgroup of 4 waves takes 64 x R slit rows and 64 frames, a wave the frames wave, wave + 4, ..., BATCH of them in flight.
//     hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_width_probe tools/probes/gather_width_probe.hip && /tmp/gather_width_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int R> struct Vec;
template <> struct Vec<1> { using type = uint16_t; };
template <> struct Vec<2> { using type = uint32_t; };
template <> struct Vec<4> { using type = uint2; };
template <> struct Vec<8> { using type = uint4; };

__device__ __forceinline__ uint32_t fold(uint16_t v) { return v; }
__device__ __forceinline__ uint32_t fold(uint32_t v) { return v; }
__device__ __forceinline__ uint32_t fold(uint2 v) { return v.x ^ v.y; }
__device__ __forceinline__ uint32_t fold(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

// grid (frames / 64, slit rows / (64 R), shift groups); 256 threads
template <int R, int SC, int BATCH>
__global__ __launch_bounds__(256) void k_gather(const uint16_t* __restrict__ stack, int n_frames, int64_t fstride, int width, int x0, uint32_t* out) {
    using V = typename Vec<R>::type;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int y = (blockIdx.y * 64 + lane) * R;                 // first slit row of this lane = file column (the rotation's mirror left out)
    const int s0 = blockIdx.z * SC;
    uint32_t off[SC], offr[SC];
#pragma unroll
    for (int s = 0; s < SC; ++s) {
        off[s] = (uint32_t)(((int64_t)(x0 + s0 + s) * width + y) * 2);      // file row x0 + shift, this lane's columns
        offr[s] = off[s] + (uint32_t)width * 2;                              // the right sample: the next file row
    }
    uint32_t acc = 0;
    for (int cb = wave; cb < 64; cb += 4 * BATCH) {
        V lv[BATCH][SC], rv[BATCH][SC];
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
            const int k = blockIdx.x * 64 + cb + 4 * i;
            const char* f = reinterpret_cast<const char*>(stack + (int64_t)(k < n_frames ? k : 0) * fstride);
#pragma unroll
            for (int s = 0; s < SC; ++s) {
                lv[i][s] = *reinterpret_cast<const V*>(f + off[s]);
                rv[i][s] = *reinterpret_cast<const V*>(f + offr[s]);
            }
        }
#pragma unroll
        for (int i = 0; i < BATCH; ++i)
#pragma unroll
            for (int s = 0; s < SC; ++s) acc ^= fold(lv[i][s]) ^ fold(rv[i][s]);
    }
    out[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 256 + threadIdx.x] = acc;
}

template <int R, int SC, int BATCH>
void run(const uint16_t* stack, int n, int64_t fstride, int width, int ih, int S, uint32_t* out, const char* what) {
    dim3 grid((n + 63) / 64, ih / (64 * R), (S + SC - 1) / SC);
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        CHECK(hipEventRecord(a, 0));
        hipLaunchKernelGGL((k_gather<R, SC, BATCH>), grid, dim3(256), 0, 0, stack, n, fstride, width, 80, out);
        CHECK(hipEventRecord(b, 0));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (rep > 0 && ms < best) best = ms;
    }
    const double bytes = (double)n * grid.z * SC * 2 * ih * 2;      // what the loads ask for (shared file rows counted by every group that reads them)
    const double distinct = (double)n * (S + 1) * ih * 2;
    printf("%-44s %7.1f us  %5.2f TB/s asked, %5.2f TB/s of distinct bytes; %u workgroups, %d B a lane-load, %.0f KB in flight a wave\n", what, best * 1e3,
           bytes / (best * 1e-3) / 1e12, distinct / (best * 1e-3) / 1e12, grid.x * grid.y * grid.z, 2 * R, 64.0 * 2 * R * SC * 2 * BATCH / 1024);
}

// Round 6: every file row of the band ONCE (what k_extract_band asks for): a lane owns R consecutive slit rows and loads its 2 R
// bytes of each of the NR file rows of a frame; 8 waves a workgroup, FPW frames a wave, all FPW x NR loads in flight.
// grid (frames / (8 FPW), slit rows / (64 R)); the band of NR = 8 rows (a group of seven shifts) or 22 (all of them at once)
template <int R, int NR, int FPW>
__global__ __launch_bounds__(512) void k_band_once(const uint16_t* __restrict__ stack, int n_frames, int64_t fstride, int width, int x0, uint32_t* out) {
    using V = typename Vec<R>::type;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int y = (blockIdx.y * 64 + lane) * R;
    const int g0 = blockIdx.z * (NR - 1);
    uint32_t voff[NR];
#pragma unroll
    for (int d = 0; d < NR; ++d) voff[d] = (uint32_t)(((int64_t)(x0 + g0 + d) * width + y) * 2);
    V v[FPW][NR];
#pragma unroll
    for (int i = 0; i < FPW; ++i) {
        const int k = (blockIdx.x * 8 + wave) * FPW + i;
        const char* f = reinterpret_cast<const char*>(stack + (int64_t)(k < n_frames ? k : 0) * fstride);
#pragma unroll
        for (int d = 0; d < NR; ++d) v[i][d] = *reinterpret_cast<const V*>(f + voff[d]);
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < FPW; ++i)
#pragma unroll
        for (int d = 0; d < NR; ++d) acc ^= fold(v[i][d]);
    out[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 512 + threadIdx.x] = acc;
}

template <int R, int NR, int FPW>
void run_once(const uint16_t* stack, int n, int64_t fstride, int width, int ih, int S, uint32_t* out, const char* what) {
    dim3 grid((n + 8 * FPW - 1) / (8 * FPW), ih / (64 * R), (S + NR - 2) / (NR - 1));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        CHECK(hipEventRecord(a, 0));
        hipLaunchKernelGGL((k_band_once<R, NR, FPW>), grid, dim3(512), 0, 0, stack, n, fstride, width, 80, out);
        CHECK(hipEventRecord(b, 0));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (rep > 0 && ms < best) best = ms;
    }
    const double distinct = (double)n * (S + 1) * ih * 2;
    printf("%-52s %7.1f us  %5.2f TB/s of distinct bytes; %u workgroups, %d B a lane-load, %d loads in flight a wave\n", what, best * 1e3,
           distinct / (best * 1e-3) / 1e12, grid.x * grid.y * grid.z, 2 * R, FPW * NR);
}

int main() {
    const int n = 2000, height = 200, width = 2000, S = 21;
    const int64_t fstride = 802816 / 2;
    uint16_t* stack;
    uint32_t* out;
    CHECK(hipMalloc(&stack, (size_t)n * fstride * 2));
    CHECK(hipMalloc(&out, 64 << 20));
    CHECK(hipMemset(stack, 1, (size_t)n * fstride * 2));
    (void)height;
    const int ih = 1536;        // a multiple of 64 x 8 (of the 2000 slit rows: the same rows for every variant)
    run<1, 4, 8>(stack, n, fstride, width, ih, S, out, "1 row a lane, SC 4, batch 8 (k_extract)");
    run<1, 4, 16>(stack, n, fstride, width, ih, S, out, "1 row a lane, SC 4, batch 16");
    run<2, 4, 8>(stack, n, fstride, width, ih, S, out, "2 rows a lane, SC 4, batch 8");
    run<4, 4, 8>(stack, n, fstride, width, ih, S, out, "4 rows a lane, SC 4, batch 8");
    run<4, 4, 4>(stack, n, fstride, width, ih, S, out, "4 rows a lane, SC 4, batch 4");
    run<4, 2, 8>(stack, n, fstride, width, ih, S, out, "4 rows a lane, SC 2, batch 8");
    run<8, 4, 4>(stack, n, fstride, width, ih, S, out, "8 rows a lane, SC 4, batch 4");
    run<8, 2, 8>(stack, n, fstride, width, ih, S, out, "8 rows a lane, SC 2, batch 8");
    run<8, 2, 4>(stack, n, fstride, width, ih, S, out, "8 rows a lane, SC 2, batch 4");
    run<8, 2, 2>(stack, n, fstride, width, ih, S, out, "8 rows a lane, SC 2, batch 2");
    run<8, 1, 8>(stack, n, fstride, width, ih, S, out, "8 rows a lane, SC 1, batch 8");
    run<8, 11, 1>(stack, n, fstride, width, ih, S, out, "8 rows a lane, SC 11, batch 1");
    // round 6: every row of a group of seven shifts once (8 rows; 24 loads per slit row and frame), lane width 2 .. 16 bytes
    run_once<1, 8, 8>(stack, n, fstride, width, ih, S, out, "once: 1 row a lane (128 B a wave-load), 8 x 8");
    run_once<2, 8, 8>(stack, n, fstride, width, ih, S, out, "once: 2 rows a lane (256 B), 8 x 8");
    run_once<4, 8, 4>(stack, n, fstride, width, ih, S, out, "once: 4 rows a lane (512 B), 8 x 4");
    run_once<4, 8, 8>(stack, n, fstride, width, ih, S, out, "once: 4 rows a lane (512 B), 8 x 8");
    run_once<8, 8, 2>(stack, n, fstride, width, ih, S, out, "once: 8 rows a lane (1 KiB), 8 x 2");
    run_once<8, 8, 4>(stack, n, fstride, width, ih, S, out, "once: 8 rows a lane (1 KiB), 8 x 4");
    run_once<1, 22, 2>(stack, n, fstride, width, ih, S, out, "once: 1 row a lane, all 22 rows x 2 frames");
    run_once<8, 22, 1>(stack, n, fstride, width, ih, S, out, "once: 8 rows a lane, all 22 rows x 1 frame");
    return 0;
}
