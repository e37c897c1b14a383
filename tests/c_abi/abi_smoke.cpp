// A caller of the C ABI that has never heard of Python or torch: HIP runtime + libshg_hip.so only.
// Sums and maxima of a small synthetic 16-bit stack (pass A), columns along a straight line (pass B) and a
// row-scaling pass, each checked against a CPU loop.  Exit code 0 = all equal.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "shg_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define CHECK_SHG(x) do { int rc_ = (x); if (rc_ != 0) { printf("shg error %d: %s\n", rc_, shg_last_error_string()); return 3; } } while (0)

int main() {
    if (shg_abi_version() != SHG_ABI_VERSION) { printf("ABI mismatch\n"); return 1; }
    const int64_t n = 37, h = 24, w = 160;                 // Width > Height: a rotated file, ih = 160, iw = 24
    const int64_t npix = h * w, ih = w, iw = h;
    std::vector<uint16_t> frames(n * npix);
    uint32_t state = 12345u;
    for (auto& v : frames) { state = state * 1664525u + 1013904223u; v = (uint16_t)(state >> 16); }

    uint16_t* d_stack; uint64_t* d_sum; uint16_t* d_max; void* d_ws;
    const size_t ws_bytes = shg_accumulate_workspace_bytes(n, h, w, 2);
    CHECK_HIP(hipMalloc(&d_stack, frames.size() * 2));
    CHECK_HIP(hipMalloc(&d_sum, npix * 8));
    CHECK_HIP(hipMalloc(&d_max, npix * 2));
    CHECK_HIP(hipMalloc(&d_ws, ws_bytes));
    CHECK_HIP(hipMemcpy(d_stack, frames.data(), frames.size() * 2, hipMemcpyHostToDevice));
    CHECK_SHG(shg_accumulate_sum_max(d_stack, n, h, w, 2, 0, d_sum, d_max, d_ws, ws_bytes, nullptr));
    std::vector<uint64_t> sum(npix); std::vector<uint16_t> mx(npix);
    CHECK_HIP(hipMemcpy(sum.data(), d_sum, npix * 8, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(mx.data(), d_max, npix * 2, hipMemcpyDeviceToHost));
    for (int64_t p = 0; p < npix; ++p) {
        uint64_t s = 0; uint16_t m = 0;
        for (int64_t k = 0; k < n; ++k) { const uint16_t v = frames[k * npix + p]; s += v; m = v > m ? v : m; }
        if (s != sum[p] || m != mx[p]) { printf("pass A differs at pixel %lld\n", (long long)p); return 4; }
    }

    // pass B: one shift, line at column 7 + 0.25 for every slit row: disk[y][k] = trunc(img[y][7]*0.75 + img[y][8]*0.25)
    std::vector<int32_t> ind(ih, 7); std::vector<double> lw(ih, 0.75), rw(ih, 0.25);
    int32_t* d_ind; double *d_lw, *d_rw; uint16_t* d_disk;
    CHECK_HIP(hipMalloc(&d_ind, ih * 4)); CHECK_HIP(hipMalloc(&d_lw, ih * 8)); CHECK_HIP(hipMalloc(&d_rw, ih * 8));
    CHECK_HIP(hipMalloc(&d_disk, ih * n * 2));
    CHECK_HIP(hipMemcpy(d_ind, ind.data(), ih * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_lw, lw.data(), ih * 8, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_rw, rw.data(), ih * 8, hipMemcpyHostToDevice));
    CHECK_SHG(shg_extract_columns(d_stack, n, h, w, 2, 0, d_ind, d_lw, d_rw, 1, d_disk, n, ih * n, n, 0, 0, nullptr));
    std::vector<uint16_t> disk(ih * n);
    CHECK_HIP(hipMemcpy(disk.data(), d_disk, ih * n * 2, hipMemcpyDeviceToHost));
    for (int64_t y = 0; y < ih; ++y)
        for (int64_t k = 0; k < n; ++k) {
            // rotated file: img[y][x] = raw[x][Width - 1 - y]  (video_reader.py:119-120)
            const double l = frames[k * npix + 7 * w + (w - 1 - y)], r = frames[k * npix + 8 * w + (w - 1 - y)];
            const uint16_t want = (uint16_t)(int)(l * 0.75 + r * 0.25);
            if (disk[y * n + k] != want) { printf("pass B differs at (%lld, %lld)\n", (long long)y, (long long)k); return 5; }
        }
    (void)iw;

    // row scaling of the disk
    std::vector<double> c(ih);
    for (int64_t y = 0; y < ih; ++y) c[y] = 0.5 + 0.01 * (double)y;
    double* d_c; uint16_t* d_out;
    CHECK_HIP(hipMalloc(&d_c, ih * 8)); CHECK_HIP(hipMalloc(&d_out, ih * n * 2));
    CHECK_HIP(hipMemcpy(d_c, c.data(), ih * 8, hipMemcpyHostToDevice));
    CHECK_SHG(shg_scale_rows_u16(d_disk, ih, n, n, d_c, nullptr, d_out, n, nullptr));
    std::vector<uint16_t> out(ih * n);
    CHECK_HIP(hipMemcpy(out.data(), d_out, ih * n * 2, hipMemcpyDeviceToHost));
    for (int64_t y = 0; y < ih; ++y)
        for (int64_t k = 0; k < n; ++k) {
            double v = (double)disk[y * n + k] * c[y];
            v = v > 65535.0 ? 65535.0 : v;
            if (out[y * n + k] != (uint16_t)(int)v) { printf("scale_rows differs at (%lld, %lld)\n", (long long)y, (long long)k); return 6; }
        }

    // an argument error comes back as a status and a message, not an exception
    if (shg_accumulate_sum_max(nullptr, n, h, w, 2, 0, d_sum, d_max, d_ws, ws_bytes, nullptr) == 0) return 7;
    printf("C ABI smoke OK (abi %d): %s\n", shg_abi_version(), "pass A, pass B, scale_rows equal the CPU loops");
    return 0;
}
