// k_products8's memory pattern without its arithmetic: per lane one 16-byte load from each of two arrays and one 16-byte store to each of
// three, 21 "disks" of 2000 x 2096 u16 (the C4 launch), nontemporal or plain stores, the inputs warm in the caches or not.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/five_streams.hip -o tools/probes/five_streams && tools/probes/five_streams
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int __attribute__((ext_vector_type(4))) u32x4;
template <bool NT, int WORK>
__global__ __launch_bounds__(256) void k5(const u32x4* __restrict__ a, const u32x4* __restrict__ b, u32x4* __restrict__ o0, u32x4* __restrict__ o1,
                                         u32x4* __restrict__ o2, size_t per_disk) {
    const size_t i = (size_t)blockIdx.z * per_disk + (size_t)blockIdx.x * 256 + threadIdx.x;
    if ((size_t)blockIdx.x * 256 + threadIdx.x >= per_disk) return;
    u32x4 x = a[i], y = b[i];
#pragma unroll
    for (int k = 0; k < WORK; ++k) { x = x * 1664525u + y; y = y * 22695477u + x; }      // dependent integer work: WORK x 8 instructions
    const u32x4 p = x, q = y, r = x ^ y;
    if (NT) { __builtin_nontemporal_store(p, o0 + i); __builtin_nontemporal_store(q, o1 + i); __builtin_nontemporal_store(r, o2 + i); }
    else { o0[i] = p; o1[i] = q; o2[i] = r; }
}
template <bool NT, int WORK> float run(const u32x4* a, const u32x4* b, u32x4* o0, u32x4* o1, u32x4* o2, size_t per_disk, int disks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid((unsigned)((per_disk + 255) / 256), 1, disks);
    k5<NT, WORK><<<grid, 256>>>(a, b, o0, o1, o2, per_disk); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) k5<NT, WORK><<<grid, 256>>>(a, b, o0, o1, o2, per_disk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 100.f;      // us per launch
}
int main() {
    const int disks = 21; const size_t per_disk = 2000ull * 2096 / 8;       // 16-byte vectors
    const size_t n = per_disk * disks;
    u32x4 *a, *b, *o0, *o1, *o2;
    hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&o0, n * 16); hipMalloc(&o1, n * 16); hipMalloc(&o2, n * 16);
    hipMemset(a, 1, n * 16); hipMemset(b, 2, n * 16);
    const double mb = n * 16 * 5 / 1e6;
    printf("two reads + three writes of %.0f MB each, %.0f MB in all\n", n * 16 / 1e6, mb);
    float t;
    t = run<true, 0>(a, b, o0, o1, o2, per_disk, disks);  printf("nontemporal stores, no work : %7.1f us  %.2f TB/s\n", t, mb / t / 1e6 * 1e6 / 1e6);
    t = run<false, 0>(a, b, o0, o1, o2, per_disk, disks); printf("plain stores, no work       : %7.1f us  %.2f TB/s\n", t, mb / t);
    t = run<true, 8>(a, b, o0, o1, o2, per_disk, disks);  printf("nontemporal, 64 instructions: %7.1f us  %.2f TB/s\n", t, mb / t);
    t = run<true, 40>(a, b, o0, o1, o2, per_disk, disks); printf("nontemporal, 320 instructions: %6.1f us  %.2f TB/s\n", t, mb / t);
    t = run<true, 80>(a, b, o0, o1, o2, per_disk, disks); printf("nontemporal, 640 instructions: %6.1f us  %.2f TB/s\n", t, mb / t);
    return 0;
}
