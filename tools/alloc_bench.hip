// tools/alloc_bench.hip -- measurement aid (not part of the product): what a cold process pays for device memory on the
// MI355X box, and which way of getting it is cheapest.  Decides how pfbwt_hip.hip sizes / grows its workspace.
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/alloc_bench tools/alloc_bench.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s: %s\n", #x, hipGetErrorString(e_)); fflush(stdout); } } while (0)

__global__ void k_touch(uint4 *p, size_t n16) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; const size_t st = (size_t)gridDim.x * blockDim.x; for (; i < n16; i += st) p[i] = make_uint4(1, 2, 3, 4); }

int main(int argc, char **argv)
{
    const size_t GB = (size_t)1 << 30;
    double t0 = now_ms();
    CK(hipSetDevice(0)); CK(hipFree(0));
    printf("runtime init %.1f ms\n", now_ms() - t0);
    if (argc >= 3 && !strcmp(argv[1], "fresh")) {      // a fresh process takes <GB> at once (hipMalloc) or in 2 GiB VMM chunks ("freshvmm"), holds it <hold> s, exits
        const size_t g = (size_t)atoll(argv[2]);
        void *p = nullptr; t0 = now_ms(); hipError_t e = hipMalloc(&p, g * GB);
        printf("fresh process: hipMalloc %zu GB: %s %.1f ms\n", g, hipGetErrorString(e), now_ms() - t0); fflush(stdout);
        if (argc >= 4 && e == hipSuccess) { t0 = now_ms(); CK(hipFree(p)); printf("  hipFree %.1f ms\n", now_ms() - t0); void *q = nullptr; t0 = now_ms(); e = hipMalloc(&q, (size_t)atoll(argv[3]) * GB); printf("  then hipMalloc %s GB: %s %.1f ms\n", argv[3], hipGetErrorString(e), now_ms() - t0); }
        return 0;
    }
    if (argc >= 2 && !strcmp(argv[1], "vmmcopy")) {      // do runtime copies / fills work across the boundary of separately created + mapped handles?
#define MUST(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ABORT %s: %s\n", #x, hipGetErrorString(e_)); fflush(stdout); return 1; } } while (0)
        const size_t CH = (size_t)64 << 20;
        hipMemAllocationProp prop; memset(&prop, 0, sizeof prop);
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
        void *va = nullptr; MUST(hipMemAddressReserve(&va, 64 * GB, (size_t)2 << 30, nullptr, 0));
        printf("reserve 64 GiB asking for 2 GiB alignment: va=%p (%s)\n", va, ((uintptr_t)va & (((size_t)2 << 30) - 1)) ? "NOT aligned" : "aligned");
        hipMemAccessDesc ad; memset(&ad, 0, sizeof ad); ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
        // which (offset, size) combinations does hipMemSetAccess take?  (nothing touches these ranges)
        struct { size_t off, sz; } probe[] = {{1 * GB, 256 << 20}, {1 * GB + (256 << 20), 128 << 20}, {2 * GB + (128 << 20), 256 << 20}, {3 * GB, 192 << 20}, {4 * GB + (64 << 20), 2 * GB}};
        for (auto &pr : probe) {
            hipMemGenericAllocationHandle_t hh; MUST(hipMemCreate(&hh, pr.sz, &prop, 0));
            hipError_t em = hipMemMap((char *)va + pr.off, pr.sz, 0, hh, 0);
            hipError_t ea = em == hipSuccess ? hipMemSetAccess((char *)va + pr.off, pr.sz, &ad, 1) : em;
            printf("piece of %4zu MiB at offset %5zu MiB: map %s, access %s\n", pr.sz >> 20, pr.off >> 20, hipGetErrorString(em), hipGetErrorString(ea)); (void)hipGetLastError();
            if (em == hipSuccess) (void)hipMemUnmap((char *)va + pr.off, pr.sz);
            (void)hipMemRelease(hh);
        }
        // eight uniform 64 MiB pieces at multiples of their size
        char *reg = (char *)va + 16 * GB;
        for (int i = 0; i < 8; ++i) { hipMemGenericAllocationHandle_t hh; MUST(hipMemCreate(&hh, CH, &prop, 0)); MUST(hipMemMap(reg + i * CH, CH, 0, hh, 0)); MUST(hipMemSetAccess(reg + i * CH, CH, &ad, 1)); }
        hipStream_t s; MUST(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        const size_t nb = 3 * CH; uint8_t *hp = nullptr, *hq = nullptr; MUST(hipHostMalloc((void **)&hp, nb, 0)); MUST(hipHostMalloc((void **)&hq, nb, 0));
        for (size_t i = 0; i < nb; ++i) hp[i] = (uint8_t)(i * 7 + (i >> 20));
        char *d = reg + CH / 2;                       // [d, d + nb) spans four pieces, all mapped and accessible
        hipError_t e;
        e = hipMemcpyAsync(d, hp, nb, hipMemcpyHostToDevice, s); printf("H2D across pieces: %s", hipGetErrorString(e)); if (e != hipSuccess) return 1; e = hipStreamSynchronize(s); printf(" / sync %s\n", hipGetErrorString(e)); if (e != hipSuccess) return 1;
        memset(hq, 0, nb); e = hipMemcpyAsync(hq, d, nb, hipMemcpyDeviceToHost, s); printf("D2H across pieces: %s", hipGetErrorString(e)); if (e != hipSuccess) return 1; e = hipStreamSynchronize(s); printf(" / sync %s  data %s\n", hipGetErrorString(e), memcmp(hp, hq, nb) ? "DIFFER" : "equal"); if (e != hipSuccess) return 1;
        void *plain = nullptr; MUST(hipMalloc(&plain, nb));
        e = hipMemcpyAsync(plain, d, nb, hipMemcpyDeviceToDevice, s); printf("D2D vmm(across) -> hipMalloc: %s", hipGetErrorString(e)); if (e != hipSuccess) return 1; e = hipStreamSynchronize(s); printf(" / sync %s\n", hipGetErrorString(e)); if (e != hipSuccess) return 1;
        memset(hq, 0, nb); MUST(hipMemcpy(hq, plain, nb, hipMemcpyDeviceToHost)); printf("   data %s\n", memcmp(hp, hq, nb) ? "DIFFER" : "equal");
        e = hipMemsetAsync(d, 0x5A, nb, s); printf("memset across pieces: %s", hipGetErrorString(e)); if (e != hipSuccess) return 1; e = hipStreamSynchronize(s); printf(" / sync %s\n", hipGetErrorString(e)); if (e != hipSuccess) return 1;
        MUST(hipMemcpy(hq, d, nb, hipMemcpyDeviceToHost)); size_t bad = 0; for (size_t i = 0; i < nb; ++i) bad += hq[i] != 0x5A; printf("   memset bytes wrong: %zu\n", bad);
        e = hipMemcpyAsync(d, plain, nb, hipMemcpyDeviceToDevice, s); printf("D2D hipMalloc -> vmm(across): %s", hipGetErrorString(e)); if (e != hipSuccess) return 1; e = hipStreamSynchronize(s); printf(" / sync %s\n", hipGetErrorString(e)); if (e != hipSuccess) return 1;
        MUST(hipMemcpy(hq, d, nb, hipMemcpyDeviceToHost)); printf("   data %s\n", memcmp(hp, hq, nb) ? "DIFFER" : "equal");
        e = hipMemcpyAsync(reg + 5 * CH - 1000, d, 2000 + CH, hipMemcpyDeviceToDevice, s); printf("D2D vmm -> vmm, both across: %s", hipGetErrorString(e)); if (e != hipSuccess) return 1; e = hipStreamSynchronize(s); printf(" / sync %s\n", hipGetErrorString(e)); if (e != hipSuccess) return 1;
        e = hipMemcpy2DAsync(d + 5, 1000 + 10, hp, 1000, 1000, (nb - CH) / 1010, hipMemcpyHostToDevice, s); printf("2D H2D across pieces: %s", hipGetErrorString(e)); if (e != hipSuccess) return 1; e = hipStreamSynchronize(s); printf(" / sync %s\n", hipGetErrorString(e)); if (e != hipSuccess) return 1;
        e = hipMemset2DAsync(d + 5, 1010, 'A', 10, (nb - CH) / 1010, s); printf("2D memset across pieces: %s", hipGetErrorString(e)); if (e != hipSuccess) return 1; e = hipStreamSynchronize(s); printf(" / sync %s\n", hipGetErrorString(e)); if (e != hipSuccess) return 1;
        hipPointerAttribute_t at; e = hipPointerGetAttributes(&at, d); printf("hipPointerGetAttributes(vmm): %s type %d\n", hipGetErrorString(e), e == hipSuccess ? (int)at.type : -1); (void)hipGetLastError();
        // cost of many small pieces
        char *reg2 = (char *)va + 32 * GB;
        t0 = now_ms(); int nsm = 0; for (int i = 0; i < 256; ++i) { hipMemGenericAllocationHandle_t hh; MUST(hipMemCreate(&hh, CH, &prop, 0)); MUST(hipMemMap(reg2 + i * CH, CH, 0, hh, 0)); MUST(hipMemSetAccess(reg2 + i * CH, CH, &ad, 1)); ++nsm; }
        printf("%d pieces of 64 MiB created+mapped: %.2f ms each\n", nsm, (now_ms() - t0) / (nsm ? nsm : 1));
        t0 = now_ms(); k_touch<<<4096, 256, 0, s>>>((uint4 *)reg2, (size_t)nsm * CH / 16); MUST(hipStreamSynchronize(s)); printf("touch 16 GiB of 64 MiB pieces: %.1f ms\n", now_ms() - t0);
        t0 = now_ms(); k_touch<<<4096, 256, 0, s>>>((uint4 *)reg2, (size_t)nsm * CH / 16); MUST(hipStreamSynchronize(s)); printf("again: %.1f ms\n", now_ms() - t0);
        char *reg3 = (char *)va + 48 * GB;
        t0 = now_ms(); nsm = 0; for (int i = 0; i < 512; ++i) { hipMemGenericAllocationHandle_t hh; MUST(hipMemCreate(&hh, (size_t)2 << 20, &prop, 0)); MUST(hipMemMap(reg3 + (size_t)i * (2 << 20), (size_t)2 << 20, 0, hh, 0)); MUST(hipMemSetAccess(reg3 + (size_t)i * (2 << 20), (size_t)2 << 20, &ad, 1)); ++nsm; }
        printf("%d pieces of 2 MiB created+mapped: %.3f ms each\n", nsm, (now_ms() - t0) / (nsm ? nsm : 1));
        t0 = now_ms(); k_touch<<<4096, 256, 0, s>>>((uint4 *)reg3, (size_t)nsm * (2 << 20) / 16); MUST(hipStreamSynchronize(s)); printf("touch 1 GiB of 2 MiB pieces: %.2f ms\n", now_ms() - t0);
        t0 = now_ms(); k_touch<<<4096, 256, 0, s>>>((uint4 *)reg3, (size_t)nsm * (2 << 20) / 16); MUST(hipStreamSynchronize(s)); printf("again: %.2f ms\n", now_ms() - t0);
        return 0;
    }
    if (argc >= 3 && !strcmp(argv[1], "freshvmm")) {
        const size_t g = (size_t)atoll(argv[2]);
        hipMemAllocationProp prop; memset(&prop, 0, sizeof prop);
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
        void *va = nullptr; CK(hipMemAddressReserve(&va, 512 * GB, 0, nullptr, 0));
        hipMemAccessDesc ad; memset(&ad, 0, sizeof ad); ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
        double tall = now_ms(), worst = 0; std::vector<hipMemGenericAllocationHandle_t> hs;
        for (size_t i = 0; i < g / 2; ++i) {
            hipMemGenericAllocationHandle_t h; t0 = now_ms();
            if (hipMemCreate(&h, 2 * GB, &prop, 0) != hipSuccess) { printf("create failed at chunk %zu\n", i); break; }
            CK(hipMemMap((char *)va + i * 2 * GB, 2 * GB, 0, h, 0)); CK(hipMemSetAccess((char *)va + i * 2 * GB, 2 * GB, &ad, 1));
            const double t = now_ms() - t0; if (t > worst) worst = t; if (t > 50) printf("  chunk %zu: %.1f ms\n", i, t);
            hs.push_back(h);
        }
        printf("fresh process: %zu VMM chunks of 2 GB: %.1f ms total, worst chunk %.1f ms\n", hs.size(), now_ms() - tall, worst); fflush(stdout);
        hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        t0 = now_ms(); k_touch<<<4096, 256, 0, s>>>((uint4 *)va, hs.size() * 2 * GB / 16); CK(hipStreamSynchronize(s)); printf("  touch all: %.1f ms\n", now_ms() - t0);
        // move the physical chunk of slot 0 to the first unmapped slot (what a two-ended arena does instead of freeing)
        t0 = now_ms(); CK(hipMemUnmap(va, 2 * GB)); CK(hipMemMap((char *)va + hs.size() * 2 * GB, 2 * GB, 0, hs[0], 0)); CK(hipMemSetAccess((char *)va + hs.size() * 2 * GB, 2 * GB, &ad, 1));
        printf("  remap of one chunk: %.2f ms\n", now_ms() - t0);
        t0 = now_ms(); k_touch<<<4096, 256, 0, s>>>((uint4 *)((char *)va + hs.size() * 2 * GB), 2 * GB / 16); CK(hipStreamSynchronize(s)); printf("  touch remapped: %.1f ms\n", now_ms() - t0);
        return 0;
    }
    size_t fr = 0, tot = 0; CK(hipMemGetInfo(&fr, &tot)); printf("free %.1f GB of %.1f GB\n", fr / 1e9, tot / 1e9);
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    // 1. hipMalloc by size, cold then again
    const size_t sizes[] = {1, 4, 16, 64, 128, 200};
    for (int rep = 0; rep < 2; ++rep)
        for (size_t g : sizes) {
            void *p = nullptr; t0 = now_ms(); hipError_t e = hipMalloc(&p, g * GB); double ta = now_ms() - t0;
            if (e != hipSuccess) { printf("hipMalloc %zu GB failed\n", g); (void)hipGetLastError(); continue; }
            t0 = now_ms(); k_touch<<<4096, 256, 0, s>>>((uint4 *)p, g * GB / 16); CK(hipStreamSynchronize(s)); double tt = now_ms() - t0;
            t0 = now_ms(); k_touch<<<4096, 256, 0, s>>>((uint4 *)p, g * GB / 16); CK(hipStreamSynchronize(s)); double tt2 = now_ms() - t0;
            t0 = now_ms(); CK(hipFree(p)); double tf = now_ms() - t0;
            printf("rep %d hipMalloc %3zu GB: alloc %8.1f ms (%.1f ms/GB)  first touch %7.1f ms  second %7.1f ms  free %7.1f ms\n", rep, g, ta, ta / g, tt, tt2, tf); fflush(stdout);
        }
    // 2. virtual memory management: reserve once, map 2 GiB chunks
    {
        hipMemAllocationProp prop; memset(&prop, 0, sizeof prop);
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
        size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
        printf("VMM granularity %zu\n", gran);
        void *va = nullptr; t0 = now_ms(); hipError_t e = hipMemAddressReserve(&va, 256 * GB, 0, nullptr, 0);
        printf("reserve 256 GB: %s %.2f ms\n", hipGetErrorString(e), now_ms() - t0);
        if (e == hipSuccess) {
            for (size_t chunk : {(size_t)2 * GB, (size_t)16 * GB}) {
                std::vector<hipMemGenericAllocationHandle_t> hs;
                const int nch = chunk == 2 * GB ? 8 : 4;
                size_t off = 0;
                for (int i = 0; i < nch; ++i) {
                    hipMemGenericAllocationHandle_t h; t0 = now_ms();
                    e = hipMemCreate(&h, chunk, &prop, 0); double tc = now_ms() - t0;
                    if (e != hipSuccess) { printf("hipMemCreate failed %s\n", hipGetErrorString(e)); break; }
                    t0 = now_ms(); CK(hipMemMap((char *)va + off, chunk, 0, h, 0)); double tm = now_ms() - t0;
                    hipMemAccessDesc ad; memset(&ad, 0, sizeof ad); ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
                    t0 = now_ms(); CK(hipMemSetAccess((char *)va + off, chunk, &ad, 1)); double ts = now_ms() - t0;
                    t0 = now_ms(); k_touch<<<4096, 256, 0, s>>>((uint4 *)((char *)va + off), chunk / 16); CK(hipStreamSynchronize(s)); double tt = now_ms() - t0;
                    printf("VMM chunk %2zu GB #%d: create %7.1f map %6.2f access %7.1f touch %6.1f ms (%.1f ms/GB)\n", chunk / GB, i, tc, tm, ts, tt, (tc + tm + ts) / (chunk / GB)); fflush(stdout);
                    hs.push_back(h); off += chunk;
                }
                t0 = now_ms();
                size_t o2 = 0; for (auto h : hs) { CK(hipMemUnmap((char *)va + o2, chunk)); CK(hipMemRelease(h)); o2 += chunk; }
                printf("VMM unmap+release %zu chunks: %.1f ms\n", hs.size(), now_ms() - t0);
            }
            CK(hipMemAddressFree(va, 256 * GB));
        }
    }
    // 3. pinned host memory + H2D bandwidth, alone and while another thread allocates 64 GB
    {
        const size_t hb = 2 * GB;
        void *hp = nullptr; t0 = now_ms(); CK(hipHostMalloc(&hp, hb, hipHostMallocDefault)); printf("hipHostMalloc 2 GB: %.1f ms\n", now_ms() - t0);
        void *pg = malloc(hb); memset(pg, 1, hb);
        t0 = now_ms(); hipError_t e = hipHostRegister(pg, hb, hipHostRegisterDefault); printf("hipHostRegister 2 GB: %s %.1f ms\n", hipGetErrorString(e), now_ms() - t0);
        if (e == hipSuccess) { t0 = now_ms(); CK(hipHostUnregister(pg)); printf("hipHostUnregister: %.1f ms\n", now_ms() - t0); }
        void *d = nullptr; CK(hipMalloc(&d, hb));
        memset(hp, 2, hb);
        for (int rep = 0; rep < 2; ++rep) { t0 = now_ms(); CK(hipMemcpyAsync(d, hp, hb, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); double t = now_ms() - t0; printf("H2D pinned 2 GB: %.1f ms = %.1f GB/s\n", t, hb / t / 1e6); }
        for (int rep = 0; rep < 2; ++rep) { t0 = now_ms(); CK(hipMemcpyAsync(hp, d, hb, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); double t = now_ms() - t0; printf("D2H pinned 2 GB: %.1f ms = %.1f GB/s\n", t, hb / t / 1e6); }
        { t0 = now_ms(); CK(hipMemcpy(d, pg, hb, hipMemcpyHostToDevice)); double t = now_ms() - t0; printf("H2D pageable 2 GB: %.1f ms = %.1f GB/s\n", t, hb / t / 1e6); }
        { t0 = now_ms(); memcpy(hp, pg, hb); double t = now_ms() - t0; printf("host memcpy 2 GB (1 thread): %.1f ms = %.1f GB/s\n", t, hb / t / 1e6); }
        { t0 = now_ms(); std::vector<std::thread> th; const int nt = 8; for (int i = 0; i < nt; ++i) th.emplace_back([=] { memcpy((char *)hp + hb / nt * i, (char *)pg + hb / nt * i, hb / nt); }); for (auto &x : th) x.join();
          double t = now_ms() - t0; printf("host memcpy 2 GB (8 threads): %.1f ms = %.1f GB/s\n", t, hb / t / 1e6); }
        // bidirectional: H2D on one stream, D2H on another
        { hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking)); void *hp2 = nullptr, *d2 = nullptr; CK(hipHostMalloc(&hp2, hb, hipHostMallocDefault)); CK(hipMalloc(&d2, hb));
          t0 = now_ms(); CK(hipMemcpyAsync(d, hp, hb, hipMemcpyHostToDevice, s)); CK(hipMemcpyAsync(hp2, d2, hb, hipMemcpyDeviceToHost, s2)); CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2));
          double t = now_ms() - t0; printf("H2D + D2H concurrently, 2 GB each: %.1f ms = %.1f GB/s per direction\n", t, hb / t / 1e6); CK(hipHostFree(hp2)); CK(hipFree(d2)); CK(hipStreamDestroy(s2)); }
        // H2D while another thread runs a 64 GB hipMalloc
        {
            void *big = nullptr; double talloc = 0;
            std::thread th([&] { CK(hipSetDevice(0)); double a = now_ms(); hipError_t e2 = hipMalloc(&big, 64 * GB); talloc = now_ms() - a; if (e2 != hipSuccess) big = nullptr; });
            double tsum = 0; int cnt = 0; const double tb = now_ms();
            while (now_ms() - tb < 2500 && cnt < 40) { t0 = now_ms(); CK(hipMemcpyAsync(d, hp, hb / 4, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); double t = now_ms() - t0; tsum += t; ++cnt; if (cnt <= 12) printf("  H2D 0.5 GB during hipMalloc(64 GB): %.1f ms = %.1f GB/s\n", t, hb / 4 / t / 1e6); }
            th.join();
            printf("64 GB hipMalloc in a second thread took %.1f ms; %d copies of 0.5 GB meanwhile, mean %.1f GB/s\n", talloc, cnt, hb / 4 * cnt / tsum / 1e6);
            if (big) CK(hipFree(big));
        }
        CK(hipFree(d)); CK(hipHostFree(hp)); free(pg);
    }
    // 4. stream-ordered pool
    {
        void *p = nullptr; t0 = now_ms(); hipError_t e = hipMallocAsync(&p, 64 * GB, s); CK(hipStreamSynchronize(s));
        printf("hipMallocAsync 64 GB: %s %.1f ms\n", hipGetErrorString(e), now_ms() - t0);
        if (e == hipSuccess) { t0 = now_ms(); CK(hipFreeAsync(p, s)); CK(hipStreamSynchronize(s)); printf("hipFreeAsync: %.1f ms\n", now_ms() - t0);
            t0 = now_ms(); e = hipMallocAsync(&p, 64 * GB, s); CK(hipStreamSynchronize(s)); printf("hipMallocAsync 64 GB again: %s %.1f ms\n", hipGetErrorString(e), now_ms() - t0); if (e == hipSuccess) { CK(hipFreeAsync(p, s)); CK(hipStreamSynchronize(s)); } }
    }
    return 0;
}
