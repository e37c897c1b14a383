// How good is the hardware's double reciprocal (v_rcp_f64)?  csrc/fast_log.h takes ONE Newton step from it inside log_ratio_u16 and
// needs 2^-20 or better for the quotient of two 16-bit pixels to stay exact.  Prints the largest relative error over the products
// b (a + b) of all pixel pairs on a grid, and over 2^26 random doubles in [1, 2).
//     hipcc --offload-arch=gfx950 -O2 -o /tmp/rcp_f64_error tools/probes/rcp_f64_error.hip && /tmp/rcp_f64_error
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>

__global__ void k_err(double* worst, int mode) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    double x;
    if (mode == 0) {
        const uint32_t a = 1 + (uint32_t)((i * 2654435761ull) % 65535u), b = 1 + (uint32_t)(((i >> 7) * 40503ull + i) % 65535u);
        x = (double)b * ((double)a + (double)b);
    } else {
        uint64_t s = i * 0x9e3779b97f4a7c15ull + 12345;
        s ^= s >> 29; s *= 0xbf58476d1ce4e5b9ull; s ^= s >> 32;
        x = 1.0 + (double)(s >> 11) / 9007199254740992.0;
    }
    const double y = __builtin_amdgcn_rcp(x);
    const double e = fabs(fma(-x, y, 1.0));               // |1 - x y| = the relative error of y, exactly
    double m = e;
    for (int d = 32; d >= 1; d >>= 1) { const double o = __shfl_xor(m, d); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned long long*>(worst), (unsigned long long)__double_as_longlong(m));
}

int main() {
    double* d;
    hipMalloc(&d, 8);
    for (int mode = 0; mode < 2; ++mode) {
        hipMemset(d, 0, 8);
        hipLaunchKernelGGL(k_err, dim3(1 << 18), dim3(256), 0, 0, d, mode);
        double w;
        hipMemcpy(&w, d, 8, hipMemcpyDeviceToHost);
        printf("%s: worst relative error of v_rcp_f64 %.3e = 2^%.2f\n", mode == 0 ? "b (a + b) of 16-bit pixel pairs" : "random doubles in [1, 2)", w, log2(w));
    }
    return 0;
}
