// How fast can several host threads feed one GPU, each through its own stream?  launch_rate [kernel_us]
//   part 1: T threads launch an empty kernel 20000 times each -> aggregate launches per second (a process-wide lock in the
//           runtime shows as a rate that does not grow with T);
//   part 2: the same with a kernel that spins for `kernel_us` on ONE workgroup -> do kernels of different streams overlap?
//           (wall time against the sum of kernel times)
//   part 3: a stream synchronise after every 10 launches (the shape of a stage composite).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

__global__ void k_empty() {}
__global__ void k_spin(long long cycles) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const double kernel_us = argc > 1 ? atof(argv[1]) : 10.0;
    const long long cycles = (long long)(kernel_us * 100.0);     // wall_clock64 ticks at 100 MHz
    const int threads_list[] = {1, 2, 4, 8};
    for (int part = 1; part <= 3; ++part) {
        for (int T : threads_list) {
            std::vector<hipStream_t> st(T);
            for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
            const int n = part == 1 ? 20000 : 4000;
            hipDeviceSynchronize();
            const double t0 = now();
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t] {
                    hipSetDevice(0);
                    for (int i = 0; i < n; ++i) {
                        if (part == 1) k_empty<<<1, 64, 0, st[t]>>>();
                        else k_spin<<<1, 64, 0, st[t]>>>(cycles);
                        if (part == 3 && i % 10 == 9) hipStreamSynchronize(st[t]);
                    }
                    hipStreamSynchronize(st[t]);
                });
            for (auto& x : th) x.join();
            const double dt = now() - t0;
            if (part == 1)
                printf("empty kernels    T=%d: %.2f M launches/s aggregate, %.2f us per launch per thread\n", T, T * n / dt / 1e6, dt / n * 1e6);
            else
                printf("%s T=%d: wall %.1f ms, kernel time per stream %.1f ms -> concurrency %.2f of %d\n", part == 2 ? "spin kernels    " : "spin + sync / 10",
                       T, dt * 1e3, n * kernel_us / 1e3, T * n * kernel_us / 1e6 / dt, T);
            for (auto& s : st) hipStreamDestroy(s);
        }
    }
    return 0;
}
