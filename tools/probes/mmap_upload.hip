// Can the DMA engines read the page cache directly?  mmap a 1.6 GB file in /dev/shm, hipHostRegister the mapping (whole, or in
// chunks from several threads), copy it to the GPU from there, unregister -- each step timed -- against the product's way
// (pread into pinned buffers, then DMA: 47-50 GB/s, 34 ms per file).   mmap_upload [file_MB] [threads]
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const size_t mb = argc > 1 ? atol(argv[1]) : 1600;
    const int threads = argc > 2 ? atoi(argv[2]) : 8;
    const size_t bytes = mb << 20;
    const char* path = "/dev/shm/shg_mmap_probe.bin";
    {
        int fd = open(path, O_CREAT | O_RDWR | O_TRUNC, 0600);
        std::vector<char> block(16 << 20, 7);
        for (size_t off = 0; off < bytes; off += block.size()) (void)!write(fd, block.data(), block.size());
        close(fd);
    }
    void* dev = nullptr;
    hipMalloc(&dev, bytes);
    std::vector<hipStream_t> st(threads);
    for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int flags_i = 0; flags_i < 2; ++flags_i) {
        const unsigned flags = flags_i == 0 ? hipHostRegisterReadOnly : hipHostRegisterDefault;
        for (int pass = 0; pass < 3; ++pass) {
            int fd = open(path, O_RDONLY);
            const double t0 = now();
            void* map = mmap(nullptr, bytes, flags_i == 0 ? PROT_READ : (PROT_READ | PROT_WRITE), flags_i == 0 ? MAP_SHARED : MAP_PRIVATE, fd, 0);
            if (map == MAP_FAILED) { perror("mmap"); return 1; }
            const double t1 = now();
            // register in `threads` chunks from as many threads
            std::vector<std::thread> th;
            std::vector<int> rc(threads, 0);
            const size_t chunk = (bytes / threads + 4095) / 4096 * 4096;
            for (int t = 0; t < threads; ++t)
                th.emplace_back([&, t] {
                    const size_t off = t * chunk;
                    if (off >= bytes) return;
                    const size_t len = off + chunk <= bytes ? chunk : bytes - off;
                    rc[t] = (int)hipHostRegister((char*)map + off, len, flags);
                });
            for (auto& x : th) x.join();
            const double t2 = now();
            int bad = 0;
            for (int r : rc) bad |= r;
            if (bad) {
                printf("flags %s: hipHostRegister failed (%s)\n", flags_i == 0 ? "ReadOnly/MAP_SHARED" : "Default/MAP_PRIVATE", hipGetErrorString((hipError_t)bad));
                (void)hipGetLastError();
                munmap(map, bytes);
                close(fd);
                break;
            }
            for (int t = 0; t < threads; ++t) {
                const size_t off = t * chunk;
                if (off >= bytes) continue;
                const size_t len = off + chunk <= bytes ? chunk : bytes - off;
                hipMemcpyAsync((char*)dev + off, (char*)map + off, len, hipMemcpyHostToDevice, st[t]);
            }
            for (auto& s : st) hipStreamSynchronize(s);
            const double t3 = now();
            for (int t = 0; t < threads; ++t) {
                const size_t off = t * chunk;
                if (off < bytes) hipHostUnregister((char*)map + off);
            }
            const double t4 = now();
            munmap(map, bytes);
            close(fd);
            printf("%s, %d chunks: mmap %.1f ms, register %.1f ms, copy %.1f ms (%.1f GB/s), unregister %.1f ms -> file in HBM after %.1f ms = %.1f GB/s\n",
                   flags_i == 0 ? "ReadOnly / MAP_SHARED " : "Default / MAP_PRIVATE", threads, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3,
                   bytes / (t3 - t2) / 1e9, (t4 - t3) * 1e3, (t4 - t0) * 1e3, bytes / (t4 - t0) / 1e9);
        }
    }
    unlink(path);
    return 0;
}
