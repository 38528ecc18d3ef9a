// tools/pin_bench.hip -- measurement aid: what the file reader's ring of page-locked blocks costs a COLD process (16 x hipHostMalloc(64 MiB),
// each by the reader thread that first fills the slot), one after the other and from 16 threads at once; and what pread from a
// memory-resident file delivers into such blocks with 4 / 8 / 16 threads.   usage: pin_bench [file]
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/pin_bench tools/pin_bench.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fcntl.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
    const size_t BLK = (size_t)64 << 20; const int N = 16;
    double t0 = now_ms(); (void)hipSetDevice(0); (void)hipFree(0); printf("runtime init %.1f ms\n", now_ms() - t0);
    std::vector<void *> a(N, nullptr), b(N, nullptr);
    t0 = now_ms();
    for (int k = 0; k < N; ++k) { double t1 = now_ms(); if (hipHostMalloc(&a[k], BLK, hipHostMallocDefault) != hipSuccess) return 1; if (k < 3 || k == N - 1) printf("  hipHostMalloc(64 MiB) #%d: %.1f ms\n", k, now_ms() - t1); }
    printf("16 x hipHostMalloc(64 MiB), one after the other: %.1f ms\n", now_ms() - t0);
    t0 = now_ms();
    { std::vector<std::thread> th; for (int k = 0; k < N; ++k) th.emplace_back([&, k] { (void)hipSetDevice(0); (void)hipHostMalloc(&b[k], BLK, hipHostMallocDefault); }); for (auto &t : th) t.join(); }
    printf("16 x hipHostMalloc(64 MiB), 16 threads at once: %.1f ms\n", now_ms() - t0);
    if (argc > 1) {
        const int fd = open(argv[1], O_RDONLY); struct stat sb;
        if (fd < 0 || fstat(fd, &sb) != 0) { printf("cannot open %s\n", argv[1]); return 1; }
        const size_t nblk = (size_t)sb.st_size / BLK;
        for (int pass = 0; pass < 2; ++pass)
            for (int nt : {4, 8, 16}) {
                t0 = now_ms();
                std::vector<std::thread> th;
                for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { for (size_t blk = (size_t)t; blk < nblk; blk += (size_t)nt) { size_t got = 0; while (got < BLK) { ssize_t r = pread(fd, (char *)a[t] + got, BLK - got, (off_t)(blk * BLK + got)); if (r <= 0) break; got += (size_t)r; } } });
                for (auto &t : th) t.join();
                const double ms = now_ms() - t0;
                printf("pass %d: pread of %.1f GB into page-locked blocks with %2d threads: %.0f ms = %.1f GB/s\n", pass, (double)(nblk * BLK) / 1e9, nt, ms, (double)(nblk * BLK) / ms / 1e6);
            }
        close(fd);
    }
    return 0;
}
