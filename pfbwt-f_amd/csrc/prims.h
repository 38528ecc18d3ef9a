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
#ifndef PFP_RS_ITEMS
#define PFP_RS_ITEMS 15
#endif
constexpr int RS_ITEMS = PFP_RS_ITEMS;   // 15: 3840-pair tiles, 52 KiB of LDS in the scatter kernel -> three workgroups per CU
constexpr int RS_TILE = BLOCK * RS_ITEMS;
constexpr int RS_RADIX = 256;

struct BitRange { int lo, hi; };

// One pass = four launches, no inter-workgroup waiting:
//   k_seg_hist     G workgroups, workgroup b counts the digits of its SEGMENT (a contiguous run of tiles)
//   k_seg_colscan  256 workgroups: exclusive scan down every digit column of the G x 256 table, column totals
//   k_seg_dbase    1 workgroup: exclusive scan of the 256 totals
//   k_seg_scatter  G workgroups: workgroup b walks its segment tile by tile with running per-digit cursors
//                  (first output index of digit d in segment b = dbase[d] + column prefix[b][d]); every tile is
//                  put in digit order in LDS first so that the global stores are contiguous runs.
// A single-pass chained-scan ("onesweep") variant was measured slower on MI355X at this tile size: the
// look-back over 256 digit counters per tile costs as much as the data movement.
constexpr int SEG_MAX_GRID = 4096;

template <typename K> __global__ __launch_bounds__(BLOCK) void k_seg_hist(const K *keys, uint64_t n, int shift, uint32_t tiles_per_seg, uint32_t *seg /*[G][256]*/)
{
    __shared__ uint32_t h[RS_RADIX];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t s0 = (uint64_t)blockIdx.x * tiles_per_seg * RS_TILE;
    uint64_t s1 = s0 + (uint64_t)tiles_per_seg * RS_TILE; if (s1 > n) s1 = n;
    constexpr int U = 8;                                  // independent loads in flight per thread
    const int lane = threadIdx.x & 63;
    for (uint64_t base = s0; base < s1; base += (uint64_t)U * BLOCK) {       // uniform trip count: the wave votes below
        K kk[U]; bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const uint64_t i = base + (uint64_t)u * BLOCK + threadIdx.x; ok[u] = i < s1; kk[u] = ok[u] ? keys[i] : (K)0; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // sorted or low-entropy input (a run of N, the later passes) gives a wave 64 equal digits: one add instead of 64
            // serialised LDS atomics on the same counter
            const unsigned d = (unsigned)(kk[u] >> shift) & (RS_RADIX - 1);
            const unsigned long long act = __ballot(ok[u]);
            if (!act) continue;
            const int first = __ffsll((long long)act) - 1;
            const unsigned d0 = __shfl(d, first);
            if (__ballot(ok[u] && d == d0) == act) { if (lane == first) atomicAdd(&h[d0], (uint32_t)__popcll(act)); }
            else if (ok[u]) atomicAdd(&h[d], 1u);
        }
    }
    __syncthreads();
    seg[(size_t)blockIdx.x * RS_RADIX + threadIdx.x] = h[threadIdx.x];
}
// workgroup d: seg[b][d] := sum_{b' < b} seg[b'][d];  total[d] = column sum
__global__ __launch_bounds__(BLOCK) void k_seg_colscan(uint32_t *seg, uint32_t G, unsigned long long *total)
{
    __shared__ uint32_t red[4];
    const unsigned d = blockIdx.x;
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < G; b0 += BLOCK) {
        const uint32_t b = b0 + threadIdx.x;
        const uint32_t v = b < G ? seg[(size_t)b * RS_RADIX + d] : 0u;
        uint32_t tot;
        const uint32_t e = block_excl_sum(v, red, &tot);
        if (b < G) seg[(size_t)b * RS_RADIX + d] = carry + e;
        carry += tot;
    }
    if (threadIdx.x == 0) total[d] = carry;
}
__global__ __launch_bounds__(BLOCK) void k_seg_dbase(unsigned long long *total)
{
    __shared__ unsigned long long lds[4];
    unsigned long long tot;
    const unsigned long long e = block_excl_sum(total[threadIdx.x], lds, &tot);
    total[threadIdx.x] = e;
}

constexpr uint32_t SEG_SMALL_MAX = 32;
// few segments (small sorts): one workgroup does the column prefixes and the digit bases
__global__ __launch_bounds__(BLOCK) void k_seg_small(uint32_t *seg, uint32_t G, unsigned long long *total)
{
    __shared__ unsigned long long lds[4];
    const unsigned d = threadIdx.x;
    uint32_t v[SEG_SMALL_MAX];                           // all loads first: the stores below must not serialise them
#pragma unroll
    for (uint32_t b = 0; b < SEG_SMALL_MAX; ++b) v[b] = b < G ? seg[(size_t)b * RS_RADIX + d] : 0u;
    uint32_t run = 0;
#pragma unroll
    for (uint32_t b = 0; b < SEG_SMALL_MAX; ++b) { if (b < G) seg[(size_t)b * RS_RADIX + d] = run; run += v[b]; }
    unsigned long long tot;
    total[d] = block_excl_sum((unsigned long long)run, lds, &tot);
}

// lanes of the wave that hold the same 8-bit digit as this one (two 32-bit halves, AND-ed into plo / phi): per bit a
// sign-extended copy of it, one ballot, two xnor, two and -- the select-and-mask chain the compiler makes of
// "peers &= bit ? m : ~m" on 64-bit values is twice as long, and these sorts are VALU-bound
__device__ __forceinline__ void same_digit_lanes(uint32_t d, uint32_t &plo, uint32_t &phi)
{
#pragma unroll
    for (int bb = 0; bb < 8; ++bb) {
        const uint32_t neg = (uint32_t)(((int32_t)(d << (31 - bb))) >> 31);
        const unsigned long long m = __builtin_amdgcn_ballot_w64(neg != 0u);
        plo &= ~((uint32_t)m ^ neg); phi &= ~((uint32_t)(m >> 32) ^ neg);
    }
}

template <typename K, bool STAGE> __global__ __launch_bounds__(BLOCK) void k_seg_scatter(const K *keys, const uint32_t *vals, K *okeys, uint32_t *ovals, uint64_t n, int shift,
                                                                           uint32_t tiles_per_seg, const uint32_t *seg /*[G][256] column prefixes*/, const unsigned long long *dbase)
{
    constexpr int ITEMS = RS_ITEMS, TILE = RS_TILE;
    __shared__ uint32_t wh[BLOCK / WAVE][RS_RADIX];      // per-wave digit counts, then per-wave local cursors
    __shared__ unsigned long long gdelta[RS_RADIX];      // global index of tile-sorted element j with digit d = gdelta[d] + j
    __shared__ K skeys[STAGE ? TILE : 1];
    __shared__ uint32_t svals[STAGE ? TILE : 1];
    __shared__ uint32_t red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long cursor = dbase[threadIdx.x] + seg[(size_t)blockIdx.x * RS_RADIX + threadIdx.x];   // thread d: next output index of digit d
    // software pipeline: the pairs of tile t+1 are requested while tile t is ranked, staged and stored
    K kn[ITEMS]; uint32_t vn[ITEMS];
    {
        const uint64_t b0 = (uint64_t)blockIdx.x * tiles_per_seg * TILE + (uint64_t)wave * (ITEMS * WAVE) + lane;
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) { const uint64_t i = b0 + (uint64_t)it * WAVE; if (i < n) { kn[it] = keys[i]; vn[it] = vals[i]; } else { kn[it] = 0; vn[it] = 0; } }
    }
    for (uint32_t tl = 0; tl < tiles_per_seg; ++tl) {
        const uint64_t tbase = ((uint64_t)blockIdx.x * tiles_per_seg + tl) * TILE;
        if (tbase >= n) break;                            // uniform
        const uint32_t tile_n = (n - tbase) < (uint64_t)TILE ? (uint32_t)(n - tbase) : (uint32_t)TILE;
#pragma unroll
        for (int w = 0; w < BLOCK / WAVE; ++w) wh[w][threadIdx.x] = 0;
        __syncthreads();
        K k[ITEMS]; uint32_t v[ITEMS]; uint16_t rk[ITEMS];      // rk: rank among the wave's earlier pairs with the same digit
        const uint64_t base = tbase + (uint64_t)wave * (ITEMS * WAVE) + lane;
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const uint64_t i = base + (uint64_t)it * WAVE;
            k[it] = kn[it]; v[it] = vn[it];
            // count and rank in one step: the lanes holding the same digit are found with 8 votes, their leader advances
            // the wave's counter once (no LDS atomics, no serialisation on equal digits)
            const bool valid = i < n;
            const unsigned d = (unsigned)(k[it] >> shift) & (RS_RADIX - 1);
            const unsigned long long vm = __builtin_amdgcn_ballot_w64(valid);
            uint32_t plo = (uint32_t)vm, phi = (uint32_t)(vm >> 32);
            same_digit_lanes(d, plo, phi);
            const int leader = !valid ? lane : plo ? __builtin_ctz(plo) : 32 + __builtin_ctz(phi);
            uint32_t old = 0;
            if (valid && lane == leader) { old = wh[wave][d]; wh[wave][d] = old + (uint32_t)__builtin_popcount(plo) + (uint32_t)__builtin_popcount(phi); }
            old = __shfl(old, leader);
            rk[it] = (uint16_t)(old + __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u)));
        }
        if (tl + 1 < tiles_per_seg) {
            const uint64_t nb = base + TILE;
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) { const uint64_t i = nb + (uint64_t)it * WAVE; if (i < n) { kn[it] = keys[i]; vn[it] = vals[i]; } else { kn[it] = 0; vn[it] = 0; } }
        }
        __syncthreads();
        {   // thread d owns digit value d
            const unsigned d = threadIdx.x;
            uint32_t cw[BLOCK / WAVE]; uint32_t total = 0;
#pragma unroll
            for (int w = 0; w < BLOCK / WAVE; ++w) { cw[w] = wh[w][d]; total += cw[w]; }
            uint32_t tt;
            const uint32_t locbase = block_excl_sum(total, red, &tt);     // first tile-sorted index of digit d
            gdelta[d] = cursor - (unsigned long long)locbase;
            cursor += total;
            uint32_t run = locbase;
#pragma unroll
            for (int w = 0; w < BLOCK / WAVE; ++w) { wh[w][d] = run; run += cw[w]; }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const uint64_t i = base + (uint64_t)it * WAVE;
            const bool valid = i < n;
            const unsigned d = (unsigned)(k[it] >> shift) & (RS_RADIX - 1);
            if (valid) {
                const uint32_t li = wh[wave][d] + rk[it];             // wave's first index for this digit + rank inside the wave
                if (STAGE) { skeys[li] = k[it]; svals[li] = v[it]; }
                else { const unsigned long long pos = gdelta[d] + li; okeys[pos] = k[it]; ovals[pos] = v[it]; }
            }
        }
        __syncthreads();
        if (STAGE) {
            for (uint32_t j = threadIdx.x; j < tile_n; j += BLOCK) {
                const K kk = skeys[j];
                const unsigned long long pos = gdelta[(unsigned)(kk >> shift) & (RS_RADIX - 1)] + j;
                okeys[pos] = kk; ovals[pos] = svals[j];
            }
            __syncthreads();                              // LDS is reused by the next tile
        }
    }
}

// ---- one tile or less: every pass inside ONE workgroup, the pairs never leave LDS/registers -----------------
// (late doubling rounds and small parses sort a few hundred to a few thousand pairs; four launches per
// pass would be pure launch latency)
constexpr int SMALL_MAX_PASSES = 8;
struct PassList { int shift[SMALL_MAX_PASSES]; int npass; };
template <typename K> __global__ __launch_bounds__(BLOCK) void k_small_sort(const K *keys, const uint32_t *vals, K *okeys, uint32_t *ovals, uint32_t n, PassList pl)
{
    constexpr int ITEMS = RS_ITEMS, TILE = RS_TILE;
    __shared__ uint32_t wh[BLOCK / WAVE][RS_RADIX];
    __shared__ K skeys[TILE];
    __shared__ uint32_t svals[TILE];
    __shared__ uint32_t red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lt = lane ? (~0ULL >> (64 - lane)) : 0ULL;
    const uint32_t base = (uint32_t)wave * (ITEMS * WAVE) + lane;
    for (uint32_t j = threadIdx.x; j < n; j += BLOCK) { skeys[j] = keys[j]; svals[j] = vals[j]; }
    __syncthreads();
    for (int p = 0; p < pl.npass; ++p) {
        const int shift = pl.shift[p];
        K k[ITEMS]; uint32_t v[ITEMS];
#pragma unroll
        for (int w = 0; w < BLOCK / WAVE; ++w) wh[w][threadIdx.x] = 0;
        __syncthreads();
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const uint32_t i = base + (uint32_t)it * WAVE;
            if (i < n) { k[it] = skeys[i]; v[it] = svals[i]; atomicAdd(&wh[wave][(unsigned)(k[it] >> shift) & (RS_RADIX - 1)], 1u); }
            else { k[it] = 0; v[it] = 0; }
        }
        __syncthreads();
        {
            const unsigned d = threadIdx.x;
            uint32_t cw[BLOCK / WAVE]; uint32_t total = 0;
#pragma unroll
            for (int w = 0; w < BLOCK / WAVE; ++w) { cw[w] = wh[w][d]; total += cw[w]; }
            uint32_t tt;
            uint32_t run = block_excl_sum(total, red, &tt);
#pragma unroll
            for (int w = 0; w < BLOCK / WAVE; ++w) { wh[w][d] = run; run += cw[w]; }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const uint32_t i = base + (uint32_t)it * WAVE;
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
    }
    for (uint32_t j = threadIdx.x; j < n; j += BLOCK) { okeys[j] = skeys[j]; ovals[j] = svals[j]; }
}

// Sorts n pairs by the key bits in `ranges` (least significant range first).  Buffers (k0,v0) hold
// the input; (k1,v1) are scratch of the same size.  On return *rk,*rv point at the sorted arrays.
template <typename K> inline int radix_sort_pairs(pfp_ctx *c, K *k0, uint32_t *v0, K *k1, uint32_t *v1, uint64_t n,
                                                  const BitRange *ranges, int nranges, K **rk, uint32_t **rv)
{
    *rk = k0; *rv = v0;
    if (n <= 1) return PFP_OK;
    if (n <= (uint64_t)RS_TILE) {
        PassList pl; pl.npass = 0;
        for (int r = 0; r < nranges; ++r) for (int s = ranges[r].lo; s < ranges[r].hi; s += 8) { if (pl.npass == SMALL_MAX_PASSES) return PFP_E_ARG; pl.shift[pl.npass++] = s; }
        for (int p = pl.npass; p < SMALL_MAX_PASSES; ++p) pl.shift[p] = 0;
        PFP_LAUNCH(c, K_RADIX_SCATTER, n * 2 * (sizeof(K) + 4), (k_small_sort<K>), 1, (const K *)k0, (const uint32_t *)v0, k1, v1, (uint32_t)n, pl);
        *rk = k1; *rv = v1;
        return PFP_OK;
    }
    const size_t mk = c->arena.mark_hi();
    const uint32_t ntiles = nblocks(n, RS_TILE);
    const int seg_grid = (c->tun.seg_grid > 0 && c->tun.seg_grid <= SEG_MAX_GRID) ? c->tun.seg_grid : 768;   // workgroups per pass (segments)
    const int seg_stage = c->tun.seg_stage;
    const uint32_t tps = (ntiles + (uint32_t)seg_grid - 1) / (uint32_t)seg_grid;   // tiles per segment
    const uint32_t G = (ntiles + tps - 1) / tps;
    uint32_t *seg; unsigned long long *total;
    PFP_ALLOC_HI(c, seg, uint32_t, (size_t)G * RS_RADIX);
    PFP_ALLOC_HI(c, total, unsigned long long, RS_RADIX);
    K *src = k0, *dst = k1; uint32_t *sv = v0, *dv = v1;
    for (int r = 0; r < nranges; ++r) {
        for (int shift = ranges[r].lo; shift < ranges[r].hi; shift += 8) {
            PFP_LAUNCH(c, K_RADIX_HIST, n * sizeof(K), (k_seg_hist<K>), G, (const K *)src, n, shift, tps, seg);
            if (G <= SEG_SMALL_MAX) PFP_LAUNCH(c, K_SCAN_SPINE, (uint64_t)G * RS_RADIX * 8, k_seg_small, 1, seg, G, total);
            else {
                PFP_LAUNCH(c, K_SCAN_SPINE, (uint64_t)G * RS_RADIX * 8, k_seg_colscan, RS_RADIX, seg, G, total);
                PFP_LAUNCH(c, K_SCAN_SPINE, RS_RADIX * 16, k_seg_dbase, 1, total);
            }
            if (seg_stage) PFP_LAUNCH(c, K_RADIX_SCATTER, n * 2 * (sizeof(K) + 4), (k_seg_scatter<K, true>), G, (const K *)src, (const uint32_t *)sv, dst, dv, n, shift, tps,
                       (const uint32_t *)seg, (const unsigned long long *)total);
            else PFP_LAUNCH(c, K_RADIX_SCATTER, n * 2 * (sizeof(K) + 4), (k_seg_scatter<K, false>), G, (const K *)src, (const uint32_t *)sv, dst, dv, n, shift, tps,
                       (const uint32_t *)seg, (const unsigned long long *)total);
            K *tk = src; src = dst; dst = tk; uint32_t *tv = sv; sv = dv; dv = tv;
        }
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
