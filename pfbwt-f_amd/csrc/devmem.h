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
#include <mutex>
#include <vector>

namespace pfp {

constexpr size_t VM_MIN_CHUNK = (size_t)2 << 20, VM_MAX_CHUNK = (size_t)2 << 30;

// Address ranges are never given back to the runtime while the process lives: a destroyed region unmaps and releases its HBM and
// parks its (empty) range here; a later reservation of a fitting size on the same device maps new pieces INTO that range.  Why
// (MI355X box, round 3): a range that was freed with hipMemAddressFree and handed out again by hipMemAddressReserve at the same
// address faulted in the kernels of the next owner -- a context created after a large one was destroyed (tools/shard_sim.py: the
// second rank's parse, deterministically), a workspace that was re-reserved larger (test_sharded_build_rccl_world1).  Unmapping
// and mapping again inside a range that stays reserved is what allocators with expandable segments do all day and is not
// affected.  Address space is plentiful (a process would have to park thousands of card-sized ranges to run out).
struct VmRangePool {
    struct Range { char *base; size_t va_bytes, chunk; int device; };
    std::mutex mu; std::vector<Range> parked;
    static VmRangePool &get() { static VmRangePool p; return p; }
    bool take(size_t va, size_t chunk, int device, Range *out)
    {
        std::lock_guard<std::mutex> g(mu);
        size_t best = (size_t)-1;
        for (size_t i = 0; i < parked.size(); ++i)
            if (parked[i].device == device && parked[i].chunk == chunk && parked[i].va_bytes >= va && parked[i].va_bytes <= 2 * va + chunk &&
                (best == (size_t)-1 || parked[i].va_bytes < parked[best].va_bytes)) best = i;
        if (best == (size_t)-1) return false;
        *out = parked[best]; parked.erase(parked.begin() + (long)best);
        return true;
    }
    void park(const Range &r) { std::lock_guard<std::mutex> g(mu); parked.push_back(r); }
};

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
        size_t va = (bytes + ch - 1) / ch * ch;
        void *p = nullptr;
        VmRangePool::Range pr;
        hipError_t e = hipSuccess;
        if (VmRangePool::get().take(va, ch, dev, &pr)) { p = pr.base; va = pr.va_bytes; }      // an empty range parked by a destroyed region
        else e = hipMemAddressReserve(&p, va, ch, nullptr, 0);
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
            VmRangePool::get().park({base, va_bytes, chunk, device});      // the range itself stays reserved (see VmRangePool)
        } else (void)hipFree(base);
        handle.clear(); mapped.clear(); base = nullptr; va_bytes = 0; chunk = 0; committed = 0; lo_edge = hi_edge = 0; vmm = false;
    }
    // end of the piece that holds byte `off` (copies by the runtime are issued piece by piece)
    size_t piece_end(size_t off) const { return vmm ? (off / chunk + 1) * chunk : va_bytes; }
    bool holds(const void *p) const { return base && (const char *)p >= base && (const char *)p < base + va_bytes; }
};

} // namespace pfp
