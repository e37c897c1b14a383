// How many DISTINCT cache lines a CU's vector cache (TCP) takes per clock when every lane of a gather asks for a line of its own --
// the access pattern of k_clahe_interp_vm's LUT reads (one 8-byte entry group per pixel out of a 512 KiB value-major table that sits
// in L2).  Each lane issues `per_lane` independent 8-byte loads at pseudo-random 8-byte slots of a table of `table_kib` KiB; the
// variants: all 64 lanes in distinct lines / 2, 4, 8, 16 lanes sharing a line (neighbouring pixels with neighbouring values); a table
// far larger than a CU's vector cache (every line comes from L2: 64 bytes a clock a CU) and one that fits it.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/tcp_line_rate.hip -o tools/probes/tcp_line_rate && tools/probes/tcp_line_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int PER_LANE>
__global__ __launch_bounds__(256) void k_gather(const uint2* __restrict__ table, uint32_t slots_mask, int share_log2, int rounds, uint32_t* out) {
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    uint32_t acc = 0;
    for (int r = 0; r < rounds; ++r) {
        uint2 q[PER_LANE];
#pragma unroll
        for (int j = 0; j < PER_LANE; ++j) {
            // lanes that share a line: the same line index, different 8-byte slots of its 16
            const uint32_t grp = (tid >> share_log2), within = tid & ((1u << share_log2) - 1u);
            const uint32_t line = mix(grp * 977u + (uint32_t)(r * PER_LANE + j) * 0x9e3779b9u) & (slots_mask >> 4);
            q[j] = table[(line << 4) | (within & 15u)];
        }
#pragma unroll
        for (int j = 0; j < PER_LANE; ++j) acc += q[j].x ^ q[j].y;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int run(int table_kib);
int main() {
    run(512);        // L2 resident, far beyond a CU's 32 KiB of vector cache
    run(8);          // resident in every CU's vector cache
    return 0;
}
int run(int table_kib) {
    const uint32_t slots = table_kib * 1024 / 8;
    uint2* table; uint32_t* out;
    hipMalloc(&table, (size_t)slots * 8); hipMalloc(&out, 4);
    hipMemset(table, 1, (size_t)slots * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate / 1e6;
    printf("CUs %d, clock %.2f GHz, table %d KiB\n", cus, ghz, table_kib);
    const int blocks = cus * 8 * 4, rounds = 64;
    for (int share = 0; share <= 4; ++share) {
        k_gather<8><<<blocks, 256>>>(table, slots - 1, share, 4, out);
        hipDeviceSynchronize();
        hipEventRecord(a);
        k_gather<8><<<blocks, 256>>>(table, slots - 1, share, rounds, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double wave_loads = (double)blocks * 4 * rounds * 8;
        const double lines = wave_loads * (64 >> share);
        const double cyc = ms * 1e-3 * ghz * 1e9;
        printf("%2d lanes a line: %8.1f us  %6.2f cycles a wave load per CU  %5.2f lines a clock a CU  (%.2f T lane-loads/s)\n", 1 << share, ms * 1e3,
               cyc * cus / wave_loads, lines / cyc / cus, wave_loads * 64 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
