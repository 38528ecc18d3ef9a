// pfbwt-f_amd/csrc/prims.h -- device-wide primitives written for wave64 / 256-thread workgroups:
// exclusive/inclusive scans, LSD radix sort of (key, u32 value) pairs, stream compaction.
// No counterpart in the reference (it is sequential C/C++); these are the building blocks the
// suffix sorters and the emission are expressed in on the GPU.
#pragma once
#include "common.h"

namespace pfp {

// ------------------------------------------------------------------------------------------------
// workgroup scan helpers (256 threads = 4 waves)
template <typename T> __device__ __forceinline__ T wave_incl_sum(T v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { T y = __shfl_up(v, d); if (lane >= d) v += y; }
    return v;
}
template <typename T> __device__ __forceinline__ T wave_incl_max(T v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { T y = __shfl_up(v, d); if (lane >= d && y > v) v = y; }
    return v;
}

// exclusive sum over the 256 threads of the block; *total = block sum. lds: >= 4 entries.
template <typename T> __device__ __forceinline__ T block_excl_sum(T v, T *lds, T *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T inc = wave_incl_sum(v);
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    T base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < BLOCK / WAVE; ++i) { T s = lds[i]; if (i < wave) base += s; tot += s; }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}
template <typename T> __device__ __forceinline__ T block_incl_max(T v, T *lds, T *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T inc = wave_incl_max(v);
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    T base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < BLOCK / WAVE; ++i) { T s = lds[i]; if (i < wave && s > base) base = s; if (s > tot) tot = s; }
    __syncthreads();
    *total = tot;
    return inc > base ? inc : base;
}

// ------------------------------------------------------------------------------------------------
// device-wide scan: OP 0 = exclusive sum, OP 1 = inclusive max.  Tile = 256 threads x SCAN_ITEMS.
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = BLOCK * SCAN_ITEMS;

template <typename T, int OP> __global__ __launch_bounds__(BLOCK) void k_scan_reduce(const T *in, uint64_t n, T *partial)
{
    __shared__ T lds[4];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    T acc = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        uint64_t i = base + k;
        T v = i < n ? in[i] : (T)0;
        if (OP == 0) acc += v; else acc = v > acc ? v : acc;
    }
    T tot;
    if (OP == 0) (void)block_excl_sum(acc, lds, &tot); else (void)block_incl_max(acc, lds, &tot);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// single workgroup: scans `cnt` partials in place (exclusive for both ops: carry-in of each tile)
template <typename T, int OP> __global__ __launch_bounds__(BLOCK) void k_scan_spine(T *partial, uint64_t cnt, T *grand_total)
{
    __shared__ T lds[4];
    T carry = 0;
    for (uint64_t base = 0; base < cnt; base += SCAN_TILE) {
        T v[SCAN_ITEMS]; T acc = 0;
        const uint64_t b = base + (uint64_t)threadIdx.x * SCAN_ITEMS;
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) {
            v[k] = (b + k) < cnt ? partial[b + k] : (T)0;
            if (OP == 0) acc += v[k]; else acc = v[k] > acc ? v[k] : acc;
        }
        T tot, pre;
        if (OP == 0) pre = block_excl_sum(acc, lds, &tot);
        else { // exclusive max = inclusive max of the previous thread
            T inc = block_incl_max(acc, lds, &tot);
            // derive exclusive: max over threads < me.  Recompute with a shifted value.
            T sh = __shfl_up(inc, 1);
            __shared__ T wl[4];
            if ((threadIdx.x & 63) == 63) wl[threadIdx.x >> 6] = inc;
            __syncthreads();
            pre = (threadIdx.x & 63) ? sh : ((threadIdx.x >> 6) ? wl[(threadIdx.x >> 6) - 1] : (T)0);
            __syncthreads();
        }
        T run = OP == 0 ? carry + pre : (carry > pre ? carry : pre);
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) {
            if ((b + k) < cnt) partial[b + k] = run;
            if (OP == 0) run += v[k]; else run = v[k] > run ? v[k] : run;
        }
        if (OP == 0) carry += tot; else carry = tot > carry ? tot : carry;
        __syncthreads();
    }
    if (threadIdx.x == 0 && grand_total) *grand_total = carry;
}

template <typename T, int OP> __global__ __launch_bounds__(BLOCK) void k_scan_apply(const T *in, T *out, uint64_t n, const T *partial)
{
    __shared__ T lds[4];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    T v[SCAN_ITEMS]; T acc = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        uint64_t i = base + k;
        v[k] = i < n ? in[i] : (T)0;
        if (OP == 0) acc += v[k]; else acc = v[k] > acc ? v[k] : acc;
    }
    T tot;
    const T carry = partial[blockIdx.x];
    if (OP == 0) {
        T run = carry + block_excl_sum(acc, lds, &tot);
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) { uint64_t i = base + k; if (i < n) out[i] = run; run += v[k]; }
    } else {
        // inclusive max: need the max over all previous threads (exclusive) as the running start
        T inc = block_incl_max(acc, lds, &tot);
        __shared__ T wl[4];
        T sh = __shfl_up(inc, 1);
        if ((threadIdx.x & 63) == 63) wl[threadIdx.x >> 6] = inc;
        __syncthreads();
        T pre = (threadIdx.x & 63) ? sh : ((threadIdx.x >> 6) ? wl[(threadIdx.x >> 6) - 1] : (T)0);
        T run = carry > pre ? carry : pre;
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) { uint64_t i = base + k; run = v[k] > run ? v[k] : run; if (i < n) out[i] = run; }
    }
}

// out may alias in.  d_total (device, optional) receives the grand total.  Scratch from arena hi.
template <typename T, int OP> inline int device_scan(pfp_ctx *c, const T *in, T *out, uint64_t n, T *d_total)
{
    if (n == 0) { if (d_total) PFP_HIP(c, hipMemsetAsync(d_total, 0, sizeof(T), c->stream)); return PFP_OK; }
    const size_t mk = c->arena.mark_hi();
    const unsigned nb = nblocks(n, SCAN_TILE);
    T *partial; PFP_ALLOC_HI(c, partial, T, nb);
    PFP_LAUNCH(c, K_SCAN_REDUCE, n * sizeof(T), (k_scan_reduce<T, OP>), nb, in, n, partial);
    PFP_LAUNCH(c, K_SCAN_SPINE, nb * sizeof(T) * 2, (k_scan_spine<T, OP>), 1, partial, (uint64_t)nb, d_total);
    PFP_LAUNCH(c, K_SCAN_APPLY, n * sizeof(T) * 2, (k_scan_apply<T, OP>), nb, in, out, n, (const T *)partial);
    c->arena.release_hi(mk);
    return PFP_OK;
}

// ------------------------------------------------------------------------------------------------
// LSD radix sort, 8-bit digits, stable.  Tile = 4 waves x 16 rounds x 64 lanes = 4096 pairs; every
// wave owns a contiguous 1024-pair slice so that order inside the tile is wave-major.
constexpr int RS_ITEMS = 16;
constexpr int RS_TILE = BLOCK * RS_ITEMS;
constexpr int RS_RADIX = 256;

struct BitRange { int lo, hi; };

// ---- single-pass ("onesweep") variant: chained scan with decoupled look-back -------------------------
// One upfront kernel histograms every digit position (keys read once); then each pass is ONE kernel:
// a workgroup takes a ticket (tile id in arrival order, so a tile only ever waits for tiles that are
// already running), counts its digits, publishes the tile aggregate, walks back over its predecessors'
// status words until it meets an inclusive prefix, publishes its own inclusive prefix and scatters.
// Status word = flag (2 bits) | count (62 bits) in ONE 8-byte word written by one agent-scope relaxed
// atomic store and read by agent-scope relaxed atomic loads (L1-bypassing): flag and value travel
// together, so no fence is needed (MI355X_MICROARCH.md "Valid forms": granule needs no ordering).
constexpr int OS_MAX_PASSES = 8;
constexpr unsigned long long OS_FLAG_AGG = 1ULL << 62, OS_FLAG_PREFIX = 2ULL << 62, OS_VAL_MASK = (1ULL << 62) - 1ULL;
struct OsShifts { int shift[OS_MAX_PASSES]; int npass; };

template <typename K> __global__ __launch_bounds__(BLOCK) void k_radix_hist_all(const K *keys, uint64_t n, OsShifts sh, unsigned long long *ghist /*[npass][256]*/)
{
    __shared__ uint32_t h[OS_MAX_PASSES][RS_RADIX];
    for (int p = 0; p < sh.npass; ++p) h[p][threadIdx.x] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * RS_TILE + threadIdx.x;
#pragma unroll 4
    for (int it = 0; it < RS_ITEMS; ++it) {
        const uint64_t i = base + (uint64_t)it * BLOCK;
        if (i < n) { const K k = keys[i]; for (int p = 0; p < sh.npass; ++p) atomicAdd(&h[p][(unsigned)(k >> sh.shift[p]) & (RS_RADIX - 1)], 1u); }
    }
    __syncthreads();
    for (int p = 0; p < sh.npass; ++p) { const uint32_t v = h[p][threadIdx.x]; if (v) atomicAdd(&ghist[(size_t)p * RS_RADIX + threadIdx.x], (unsigned long long)v); }
}
// exclusive scan of each pass' 256 global counts -> first output index of every digit value
__global__ __launch_bounds__(BLOCK) void k_radix_bases(unsigned long long *ghist, int npass)
{
    __shared__ unsigned long long lds[4];
    for (int p = 0; p < npass; ++p) {
        unsigned long long tot;
        const unsigned long long v = ghist[(size_t)p * RS_RADIX + threadIdx.x];
        const unsigned long long e = block_excl_sum(v, lds, &tot);
        ghist[(size_t)p * RS_RADIX + threadIdx.x] = e;
    }
}

template <typename K, int ITEMS, int WIN> __global__ __launch_bounds__(BLOCK) void k_radix_onesweep(const K *keys, const uint32_t *vals, K *okeys, uint32_t *ovals, uint64_t n, int shift,
                                                                              const unsigned long long *gbase /*[256]*/, unsigned long long *status /*[tiles][256]*/,
                                                                              uint32_t *ticket, uint32_t *stuck, uint32_t ntiles, int ablate /*timing experiments only*/)
{
    constexpr int TILE = BLOCK * ITEMS;
    uint32_t static_tile = blockIdx.x;
    __shared__ uint32_t wh[BLOCK / WAVE][RS_RADIX];      // per-wave digit counts, then per-wave local cursors
    __shared__ unsigned long long gdelta[RS_RADIX];      // global index of tile-sorted element j with digit d = gdelta[d] + j
    __shared__ K skeys[TILE];                            // the tile in digit order: global stores become contiguous runs
    __shared__ uint32_t svals[TILE];
    __shared__ uint32_t red[4];
    __shared__ uint32_t s_tile;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lt = lane ? (~0ULL >> (64 - lane)) : 0ULL;
    // Persistent workgroups: each takes tiles by ticket until none is left.
    for (;;) {
        if (threadIdx.x == 0) { if (ablate & 4) { s_tile = static_tile; } else s_tile = atomicAdd(ticket, 1u); }
        static_tile += gridDim.x;
#pragma unroll
        for (int w = 0; w < BLOCK / WAVE; ++w) wh[w][threadIdx.x] = 0;
        __syncthreads();
        const uint32_t tile = s_tile;
        if (tile >= ntiles) break;                       // uniform: every wave leaves here
        K k[ITEMS]; uint32_t v[ITEMS];
        const uint64_t tbase = (uint64_t)tile * TILE;
        const uint32_t tile_n = (n - tbase) < (uint64_t)TILE ? (uint32_t)(n - tbase) : (uint32_t)TILE;
        const uint64_t base = tbase + (uint64_t)wave * (ITEMS * WAVE) + lane;
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const uint64_t i = base + (uint64_t)it * WAVE;
            if (i < n) { k[it] = keys[i]; v[it] = vals[i]; atomicAdd(&wh[wave][(unsigned)(k[it] >> shift) & (RS_RADIX - 1)], 1u); }
            else { k[it] = 0; v[it] = 0; }
        }
        __syncthreads();
        {   // thread d owns digit value d
            const unsigned d = threadIdx.x;
            uint32_t cw[BLOCK / WAVE]; uint32_t total = 0;
#pragma unroll
            for (int w = 0; w < BLOCK / WAVE; ++w) { cw[w] = wh[w][d]; total += cw[w]; }
            unsigned long long *mine = status + (size_t)tile * RS_RADIX + d;
            unsigned long long excl = 0;
            if (tile == 0 || (ablate & 1)) {
                __hip_atomic_store(mine, OS_FLAG_PREFIX | (unsigned long long)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                __hip_atomic_store(mine, OS_FLAG_AGG | (unsigned long long)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                uint32_t t = tile; bool done = false;
                while (t > 0 && !done) {
                    unsigned long long sv[WIN];
                    const int cntw = t < (uint32_t)WIN ? (int)t : WIN;
#pragma unroll
                    for (int j = 0; j < WIN; ++j)
                        sv[j] = j < cntw ? __hip_atomic_load(status + (size_t)(t - 1 - j) * RS_RADIX + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ULL;
#pragma unroll
                    for (int j = 0; j < WIN; ++j) {
                        if (j < cntw && !done) {
                            unsigned long long x = sv[j];
                            uint32_t spins = 0;
                            while ((x >> 62) == 0) {      // predecessor holds a ticket but has not published yet
                                __builtin_amdgcn_s_sleep(8);
                                x = __hip_atomic_load(status + (size_t)(t - 1 - j) * RS_RADIX + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (++spins > (1u << 22)) { atomicAdd(stuck, 1u); x = OS_FLAG_PREFIX; break; }   // never expected; bounds the wait
                            }
                            excl += x & OS_VAL_MASK;
                            if ((x >> 62) != 1) done = true;   // inclusive prefix met
                        }
                    }
                    t -= (uint32_t)cntw;
                }
                __hip_atomic_store(mine, OS_FLAG_PREFIX | (excl + (unsigned long long)total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            uint32_t tt;
            const uint32_t locbase = block_excl_sum(total, red, &tt);     // first tile-sorted index of digit d
            gdelta[d] = gbase[d] + excl - (unsigned long long)locbase;
            uint32_t run = locbase;
#pragma unroll
            for (int w = 0; w < BLOCK / WAVE; ++w) { wh[w][d] = run; run += cw[w]; }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            if (ablate & 8) break;
            const uint64_t i = base + (uint64_t)it * WAVE;
            const bool valid = i < n;
            const unsigned d = (unsigned)(k[it] >> shift) & (RS_RADIX - 1);
            unsigned long long peers = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                unsigned long long m = __ballot((d >> b) & 1);
                peers &= ((d >> b) & 1) ? m : ~m;
            }
            const int leader = valid ? (__ffsll((long long)peers) - 1) : lane;
            uint32_t old = 0;
            if (valid && lane == leader) { old = wh[wave][d]; wh[wave][d] = old + (uint32_t)__popcll(peers); }
            old = __shfl(old, leader);
            if (valid) { const uint32_t li = old + (uint32_t)__popcll(peers & lt); skeys[li] = k[it]; svals[li] = v[it]; }
        }
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < tile_n; j += BLOCK) {
            if (ablate & 2) break;
            const K kk = skeys[j];
            const unsigned long long pos = gdelta[(unsigned)(kk >> shift) & (RS_RADIX - 1)] + j;
            okeys[pos] = kk; ovals[pos] = svals[j];
        }
        __syncthreads();                                  // LDS is reused by the next tile
    }
}

// Sorts n pairs by the key bits in `ranges` (least significant range first).  Buffers (k0,v0) hold
// the input; (k1,v1) are scratch of the same size.  On return *rk,*rv point at the sorted arrays.
template <typename K> inline int radix_sort_pairs(pfp_ctx *c, K *k0, uint32_t *v0, K *k1, uint32_t *v1, uint64_t n,
                                                  const BitRange *ranges, int nranges, K **rk, uint32_t **rv)
{
    *rk = k0; *rv = v0;
    if (n <= 1) return PFP_OK;
    const size_t mk = c->arena.mark_hi();
    static int os_items = 0;   // pairs per thread of the scatter tile: 16 (4096-pair tiles) or 32 (8192)
    if (!os_items) { const char *e = getenv("PFP_OS_ITEMS"); os_items = 15; (void)e; }
    const int items = 15;   // 3840-pair tiles: 52 KiB of LDS -> three workgroups per CU
    const unsigned nb = nblocks(n, (uint64_t)BLOCK * items);
    const unsigned nbh = nblocks(n, RS_TILE);
    static int os_grid = 0;    // persistent workgroups of the scatter kernel (default 3 per CU on a 256-CU device)
    if (!os_grid) { const char *e = getenv("PFP_OS_GRID"); os_grid = (e && atoi(e) > 0) ? atoi(e) : 768; }
    const unsigned grid = nb < (unsigned)os_grid ? nb : (unsigned)os_grid;
    const int os_ablate = c->debug_ablate;   // only pfp_debug_sort sets this (timing experiments on throw-away data)
    OsShifts sh; sh.npass = 0;
    for (int r = 0; r < nranges; ++r) for (int s = ranges[r].lo; s < ranges[r].hi; s += 8) { if (sh.npass == OS_MAX_PASSES) return PFP_E_ARG; sh.shift[sh.npass++] = s; }
    for (int p = sh.npass; p < OS_MAX_PASSES; ++p) sh.shift[p] = 0;
    unsigned long long *ghist, *status; uint32_t *ctl;
    PFP_ALLOC_HI(c, ghist, unsigned long long, (size_t)OS_MAX_PASSES * RS_RADIX);
    PFP_ALLOC_HI(c, status, unsigned long long, (size_t)nb * RS_RADIX);
    PFP_ALLOC_HI(c, ctl, uint32_t, 2 * OS_MAX_PASSES + 2);   // ticket per pass, then the shared "stuck" counter
    PFP_HIP(c, hipMemsetAsync(ghist, 0, sizeof(unsigned long long) * OS_MAX_PASSES * RS_RADIX, c->stream));
    PFP_HIP(c, hipMemsetAsync(ctl, 0, sizeof(uint32_t) * (2 * OS_MAX_PASSES + 2), c->stream));
    PFP_LAUNCH(c, K_RADIX_HIST, n * sizeof(K), (k_radix_hist_all<K>), nbh, (const K *)k0, n, sh, ghist);
    PFP_LAUNCH(c, K_SCAN_SPINE, sh.npass * 4096, k_radix_bases, 1, ghist, sh.npass);
    K *src = k0, *dst = k1; uint32_t *sv = v0, *dv = v1;
    for (int p = 0; p < sh.npass; ++p) {
        PFP_HIP(c, hipMemsetAsync(status, 0, sizeof(unsigned long long) * (size_t)nb * RS_RADIX, c->stream));
        PFP_LAUNCH(c, K_RADIX_SCATTER, n * 2 * (sizeof(K) + 4), (k_radix_onesweep<K, 15, 4>), grid, (const K *)src, (const uint32_t *)sv, dst, dv, n, sh.shift[p],
                   (const unsigned long long *)(ghist + (size_t)p * RS_RADIX), status, ctl + p, ctl + 2 * OS_MAX_PASSES, nb, os_ablate & 15);
        K *tk = src; src = dst; dst = tk; uint32_t *tv = sv; sv = dv; dv = tv;
    }
    *rk = src; *rv = sv;
    c->arena.release_hi(mk);
    return PFP_OK;
}

// ------------------------------------------------------------------------------------------------
// compaction: out[j] = in[i] for the i with flag[i] != 0, order preserved; *d_count = kept.
__global__ __launch_bounds__(BLOCK) void k_compact_scatter(const uint32_t *in, const uint32_t *flag, const uint32_t *pos, uint64_t n, uint32_t *out)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n && flag[i]) out[pos[i]] = in ? in[i] : (uint32_t)i;
}
// in == nullptr compacts the indices themselves.  pos_scratch: n entries.
inline int device_compact(pfp_ctx *c, const uint32_t *in, const uint32_t *flag, uint64_t n, uint32_t *out, uint32_t *pos_scratch, uint32_t *d_count)
{
    PFP_TRY((device_scan<uint32_t, 0>(c, flag, pos_scratch, n, d_count)));
    if (n) PFP_LAUNCH(c, K_COMPACT, n * 12, k_compact_scatter, nblocks(n, BLOCK), in, flag, (const uint32_t *)pos_scratch, n, out);
    return PFP_OK;
}

inline int d2h_u32(pfp_ctx *c, const uint32_t *d, uint32_t *h)
{
    PFP_HIP(c, hipMemcpyAsync(h, d, 4, hipMemcpyDeviceToHost, c->stream));
    PFP_HIP(c, hipStreamSynchronize(c->stream));
    return PFP_OK;
}

} // namespace pfp
