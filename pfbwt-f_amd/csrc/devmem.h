// pfbwt-f_amd/csrc/devmem.h -- device memory of a context: address ranges reserved up front, physical HBM committed on
// demand (HIP virtual memory management).  Why: measured on the MI355X box (tools/alloc_bench.hip, profiles/r03a_alloc_*):
// the driver wipes freed VRAM in the background at ~40 GB/s and an allocation that does not fit into the clean part waits
// for the whole pending wipe -- 100 GB cost 0.2 ms on an idle card and 3-6 s right after another process (or this one)
// released memory; a 240 GB slab costs 1.3 s in page-table work even on an idle card.  So a context (1) never frees and
// re-allocates while it lives (no doubling + copy of the text buffer, no re-sized slab), and (2) only ever commits what its
// stages really touch (S-32G: ~100 GB instead of a 240 GB slab), in pieces of at least 64 MiB.
// No counterpart in the reference (its containers are std::vector / mmap, include/file_wrappers.hpp).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstring>
#include <vector>

namespace pfp {

constexpr size_t VM_MIN_CHUNK = (size_t)2 << 20, VM_MAX_CHUNK = (size_t)2 << 30;

// One address range, committed in pieces of ONE size (a power of two between 2 MiB and 2 GiB, at most ~256 pieces per range)
// that sit at multiples of that size -- the layout the runtime was seen to accept for hipMemSetAccess; pieces of mixed sizes
// at arbitrary 64 MiB offsets were rejected ("invalid argument", profiles/r03a_alloc_vmmcopy.log).
struct VmRegion {
    char *base = nullptr; size_t va_bytes = 0, chunk = 0; int device = 0;
    bool vmm = false;                                     // false: one plain hipMalloc (the runtime offers no virtual memory management)
    std::vector<hipMemGenericAllocationHandle_t> handle;  // per piece (valid where mapped[] is set)
    std::vector<uint8_t> mapped;
    size_t committed = 0, lo_edge = 0, hi_edge = 0;       // [0, lo_edge) and [hi_edge, va_bytes) are known to be committed

    bool live() const { return base != nullptr; }
    // reserve `bytes` of address space (nothing committed yet); hipSuccess or the runtime's error
    hipError_t reserve(size_t bytes, int dev)
    {
        destroy();
        device = dev;
        size_t ch = VM_MIN_CHUNK;
        while (ch < VM_MAX_CHUNK && ch * 256 < bytes) ch <<= 1;
        const size_t va = (bytes + ch - 1) / ch * ch;
        void *p = nullptr;
        hipError_t e = hipMemAddressReserve(&p, va, ch, nullptr, 0);
        if (e == hipSuccess) {
            base = (char *)p; va_bytes = va; chunk = ch; vmm = true; mapped.assign(va / ch, 0); handle.assign(va / ch, hipMemGenericAllocationHandle_t()); lo_edge = 0; hi_edge = va; committed = 0;
            return hipSuccess;
        }
        (void)hipGetLastError();
        e = hipMalloc(&p, va);                            // fallback: committed as a whole
        if (e != hipSuccess) { (void)hipGetLastError(); return e; }
        base = (char *)p; va_bytes = va; chunk = va; vmm = false; committed = va; lo_edge = va; hi_edge = 0;
        return hipSuccess;
    }
    // make [off0, off1) accessible.  Returns false when the device has no memory left (what was committed stays committed).
    bool commit(size_t off0, size_t off1)
    {
        if (off1 > va_bytes) return false;
        if (!vmm || off0 >= off1 || off1 <= lo_edge || off0 >= hi_edge) return true;
        const size_t s0 = off0 / chunk, s1 = (off1 + chunk - 1) / chunk;
        for (size_t s = s0; s < s1; ++s) {
            if (mapped[s]) continue;
            hipMemAllocationProp prop; memset(&prop, 0, sizeof prop);
            prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = device;
            hipMemGenericAllocationHandle_t h;
            if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
            if (hipMemMap(base + s * chunk, chunk, 0, h, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipMemRelease(h); return false; }
            hipMemAccessDesc ad; memset(&ad, 0, sizeof ad); ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
            if (hipMemSetAccess(base + s * chunk, chunk, &ad, 1) != hipSuccess) { (void)hipGetLastError(); (void)hipMemUnmap(base + s * chunk, chunk); (void)hipMemRelease(h); return false; }
            handle[s] = h; mapped[s] = 1; committed += chunk;
        }
        while (lo_edge < va_bytes && mapped[lo_edge / chunk]) lo_edge += chunk;
        while (hi_edge > 0 && mapped[hi_edge / chunk - 1]) hi_edge -= chunk;
        return true;
    }
    void destroy()
    {
        if (!base) return;
        if (vmm) {
            for (size_t s = 0; s < mapped.size(); ++s) if (mapped[s]) { (void)hipMemUnmap(base + s * chunk, chunk); (void)hipMemRelease(handle[s]); }
            (void)hipMemAddressFree(base, va_bytes);
        } else (void)hipFree(base);
        handle.clear(); mapped.clear(); base = nullptr; va_bytes = 0; chunk = 0; committed = 0; lo_edge = hi_edge = 0; vmm = false;
    }
    // end of the piece that holds byte `off` (copies by the runtime are issued piece by piece)
    size_t piece_end(size_t off) const { return vmm ? (off / chunk + 1) * chunk : va_bytes; }
    bool holds(const void *p) const { return base && (const char *)p >= base && (const char *)p < base + va_bytes; }
};

} // namespace pfp
