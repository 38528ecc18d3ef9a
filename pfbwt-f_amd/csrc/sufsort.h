// pfbwt-f_amd/csrc/sufsort.h -- suffix sorting on the device by prefix doubling with radix sort
// and active-set filtering.  Two uses:
//   * the parse (integer alphabet): stands in for sacak_int, gsa/gsacak.c:2499-2502 -> SACA_K :1397-1526,
//     called at include/pfparser.hpp:425;
//   * the dictionary (bytes, words ended by EndOfWord): stands in for gsacak + LCP,
//     gsa/gsacak.c:2504-2524 -> gSACA_K_LCP :1649-1929, called at include/pfbwt.hpp:211.
// The reference's induced sorting is inherently sequential; doubling is the data-parallel route.
// Dictionary semantics: suffixes are compared up to and including their EndOfWord, so byte-identical
// suffixes of different words end in ONE class.  The emission only asks "is glcp[j] >= suff_len"
// (pfbwt.hpp:137), i.e. "same class" -- the class head slot kept in `rank` replaces the LCP array.
#pragma once
#include "prims.h"

namespace pfp {

// 4-bit codes that keep the byte order 0 < 1 < 2 < '-' < A < C < G < N < T
__device__ __forceinline__ uint32_t dict_code(uint32_t c)
{
    return c <= 2 ? c : (c == '-') ? 3u : (c == 'A') ? 4u : (c == 'C') ? 5u : (c == 'G') ? 6u : (c == 'N') ? 7u : 8u;
}

constexpr int DK_CHARS = 16;          // characters per initial key
constexpr int DK_PER_THREAD = 16;     // suffixes per thread
constexpr int DK_TILE = BLOCK * DK_PER_THREAD;

// keys[x] = the first 16 characters of suffix x read as a 16-digit base-9 number (digits = dict_code, 0 after
// the terminator): order-preserving like 4-bit packing, but 9^16 < 2^51, i.e. 7 radix passes instead of 8.
constexpr int DK_KEY_BITS = 51;
__global__ __launch_bounds__(BLOCK) void k_dict_init_keys(const uint8_t *D, uint64_t dsize, uint64_t *keys, uint32_t *vals)
{
    __shared__ uint8_t tile[DK_TILE + DK_CHARS];
    const uint64_t t0 = (uint64_t)blockIdx.x * DK_TILE;
    for (uint32_t i = threadIdx.x; i < DK_TILE + DK_CHARS; i += BLOCK) {
        const uint64_t x = t0 + i;
        tile[i] = x < dsize ? (uint8_t)dict_code(D[x]) : (uint8_t)0;
    }
    __syncthreads();
    // consecutive threads take consecutive positions (conflict-free LDS reads, coalesced stores)
#pragma unroll 1
    for (int k = 0; k < DK_PER_THREAD; ++k) {
        const uint32_t l = (uint32_t)k * BLOCK + threadIdx.x;
        uint64_t key = 0; bool stop = false;
#pragma unroll
        for (int j = 0; j < DK_CHARS; ++j) {
            const uint32_t cc = tile[l + j];
            key = key * 9u + (stop ? 0u : cc);
            stop = stop || cc <= 1;                      // the terminator itself is part of the key, nothing after it
        }
        const uint64_t x = t0 + l;
        if (x < dsize) { keys[x] = key; vals[x] = (uint32_t)x; }
    }
}

// parse keys: (S[x], S[x+1]) with S = ranks + [0]  (pfparser.hpp:407-410)
__global__ __launch_bounds__(BLOCK) void k_int_init_keys(const uint32_t *S, uint64_t N, uint64_t *keys, uint32_t *vals)
{
    const uint64_t x = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (x >= N) return;
    keys[x] = ((uint64_t)S[x] << 32) | (x + 1 < N ? S[x + 1] : 0u);
    vals[x] = (uint32_t)x;
}

// after a sort of the active list: head flags, SA write-back, head slot for the max-scan
__global__ __launch_bounds__(BLOCK) void k_ss_heads(const uint64_t *keys, const uint32_t *vals, const uint32_t *slots /*nullable: identity*/, uint64_t na,
                                                    uint32_t *SA, uint32_t *head, uint32_t *headslot)
{
    const uint64_t a = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (a >= na) return;
    const uint32_t slot = slots ? slots[a] : (uint32_t)a;
    const uint32_t hd = (a == 0 || keys[a] != keys[a - 1]) ? 1u : 0u;
    SA[slot] = vals[a];
    head[a] = hd;
    headslot[a] = hd ? slot : 0u;
}
// rj[x] = { rank: slot of the class head of suffix x, jump: end of its covered prefix } -- one 8-byte record so
// that the doubling step costs two random gathers (x and jump[x]) instead of four
__global__ __launch_bounds__(BLOCK) void k_ss_write_rank(const uint32_t *vals, const uint32_t *newrank, uint64_t na, uint2 *rj)
{
    const uint64_t a = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (a < na) reinterpret_cast<uint32_t *>(rj)[2 * (uint64_t)vals[a]] = newrank[a];
}
// rank and jump of a re-sorted suffix in ONE 8-byte record write (the new jump travelled with the pair through the sort)
__global__ __launch_bounds__(BLOCK) void k_ss_write_rank_jump(const uint32_t *vals, const uint32_t *newrank, const uint32_t *nj, uint64_t na, uint2 *rj)
{
    const uint64_t a = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (a < na) rj[vals[a]] = make_uint2(newrank[a], nj[a]);
}
__global__ __launch_bounds__(BLOCK) void k_ss_init_rj(const uint32_t *vals, const uint32_t *newrank, uint64_t N, uint32_t h0, uint2 *rj)
{
    const uint64_t a = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (a >= N) return;
    const uint64_t x = vals[a];
    rj[x] = make_uint2(newrank[a], (uint32_t)(x + h0 < N ? x + h0 : N));
}
// keep[a] = 1 while the class of element a still has to be refined.
// ws/wordid != nullptr selects dictionary semantics: a class whose covered prefix [x, jump[x]) already
// contains the terminator is a group of byte-identical suffixes and is final.
__global__ __launch_bounds__(BLOCK) void k_ss_flag_active(const uint32_t *vals, const uint32_t *head, uint64_t na, const uint2 *rj, const uint32_t *jump_sorted /*nullable: jump of vals[a]*/,
                                                          const uint64_t *dict_init_keys /*nullable: first call of a dictionary sort*/, const uint32_t *ws, const uint32_t *wordid, uint32_t *keep)
{
    const uint64_t a = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (a >= na) return;
    const bool single = head[a] && (a + 1 == na || head[a + 1]);
    bool fin = single;
    if (!fin && dict_init_keys) {
        // after the initial sort the covered prefix is the 16 characters of the key: it contains the word's terminator exactly
        // when the last base-9 digit is 0 (padding behind a terminator) or the terminator itself (k_dict_init_keys) -- no gathers
        fin = dict_init_keys[a] % 9u <= 1u;
    } else if (!fin && ws) {
        const uint32_t x = vals[a];
        const uint32_t term = ws[wordid[x] + 1] - 1u;     // offset of the EndOfWord of x's word
        fin = (jump_sorted ? jump_sorted[a] : reinterpret_cast<const uint32_t *>(rj)[2 * (uint64_t)x + 1]) > term;
    }
    keep[a] = fin ? 0u : 1u;
}
// Doubling with pointer jumping: rank[x] orders suffix x by its covered prefix [x, jump[x]); the next
// key is (rank[x], rank[jump[x]]) and the covered prefix grows to [x, jump[jump[x]]).
// Run round (M != nullptr, first refinement of a byte text): a suffix whose first RUN_MIN characters
// are one repeated character c sits inside a run c^d; ordering such suffixes by plain doubling takes
// log2(d) rounds with the whole run active (a 10 Mbp run of N: 20 rounds x 10 M suffixes).  Instead the
// whole run is consumed at once: c^d a... is ordered among the suffixes starting with c by
// t = d if a < c, 2^32-1-d if a > c  (a = first character after the run), and jump = x + d.
constexpr uint32_t RUN_MIN = DK_CHARS;
__global__ __launch_bounds__(BLOCK) void k_ss_make_keys(const uint32_t *slots, const uint32_t *SA, const uint2 *rj, uint64_t na, uint64_t N,
                                                        const uint8_t *D, const uint32_t *M, int lowbits, uint64_t *keys, uint32_t *vals, uint32_t *nj)
{
    const uint64_t a = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (a >= na) return;
    const uint32_t x = SA[slots[a]];
    const uint2 P = rj[x];
    uint32_t low, nx;
    bool run = false;
    if (M) {
        const uint32_t ip = (uint32_t)(N - 1 - x);
        const uint32_t d = ip - M[ip] + 1u;                 // length of the run of D[x] that starts at x
        if (d >= RUN_MIN && D[x] > EndOfWord) {
            const uint64_t e = (uint64_t)x + d;
            const uint8_t nxt = e < N ? D[e] : (uint8_t)0;
            low = nxt < D[x] ? d : 0xFFFFFFFFu - d;
            nx = (uint32_t)(e < N ? e : N);
            run = true;
        }
    }
    if (!run) {
        const uint32_t y = P.y;
        const uint2 Q = y < N ? rj[y] : make_uint2(0u, (uint32_t)N);
        low = Q.x; nx = Q.y;
    }
    keys[a] = ((uint64_t)P.x << lowbits) | low;      // (rank, refinement) packed tightly: 2*bits(N) key bits
    vals[a] = x;
    nj[a] = nx;
}
__global__ __launch_bounds__(BLOCK) void k_ss_apply_jump(const uint32_t *vals, const uint32_t *nj, uint64_t na, uint2 *rj)
{
    const uint64_t a = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (a < na) reinterpret_cast<uint32_t *>(rj)[2 * (uint64_t)vals[a] + 1] = nj[a];
}
// i' = N-1-x: g[i'] = i' where x is the last position of a run of equal bytes, else 0.  The inclusive
// max-scan M of g gives, for every x, the nearest run end at or after x: runlen(x) = i' - M[i'] + 1.
__global__ __launch_bounds__(BLOCK) void k_ss_runend_marks(const uint8_t *D, uint64_t N, uint32_t *g)
{
    const uint64_t ip = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (ip >= N) return;
    const uint64_t x = N - 1 - ip;
    g[ip] = (x + 1 == N || D[x] != D[x + 1]) ? (uint32_t)ip : 0u;
}

// ---- sort of a doubling round ------------------------------------------------------------------------------------
// The active list is in slot order: the members of a class (equal rank = equal high key part) are already contiguous,
// only the order INSIDE every class is unknown.  Classes are small on the inputs that have many rounds with everything
// active (a 1000-haplotype parse: ~1000 members per class), so a workgroup sorts whole classes inside LDS: the pairs
// cross HBM once per round instead of once per radix pass (8 passes of 32 B for 58-bit keys).  Workgroup j owns the
// classes that START in [j*CS_STEP, (j+1)*CS_STEP); they end where the first class of the next stripe starts.  A range
// longer than one tile (a class with thousands of members) is left alone; those pairs are collected afterwards and go
// through the ordinary radix sort.
constexpr uint32_t CS_STEP = RS_TILE / 2;
template <typename K> __global__ __launch_bounds__(BLOCK) void k_class_tile_sort(const K *keys, const uint32_t *vals, const uint32_t *nj, K *okeys, uint32_t *ovals, uint32_t *onj,
                                                                                 uint64_t na, int lowbits, uint32_t max_range, uint8_t *done, unsigned long long *nsorted)
{
    constexpr int ITEMS = RS_ITEMS, TILE = RS_TILE;
    __shared__ uint32_t wh[BLOCK / WAVE][RS_RADIX];
    __shared__ K skeys[TILE];
    __shared__ uint16_t sidx[TILE];                     // the payload inside LDS is the pair's index in the range; x and the new jump
    __shared__ uint32_t red[4];                          // are fetched from the (cache-resident) input range when the range is written out
    __shared__ uint32_t bound[2];
    const uint64_t w0 = (uint64_t)blockIdx.x * CS_STEP;
    if (threadIdx.x < 2) bound[threadIdx.x] = 0xFFFFFFFFu;
    __syncthreads();
    for (int which = 0; which < 2; ++which) {          // first class start at or after w0 (which = 0), w0 + CS_STEP (which = 1)
        const uint64_t b = w0 + (uint64_t)which * CS_STEP;
        for (uint32_t t = threadIdx.x; t < (uint32_t)TILE; t += BLOCK) {
            const uint64_t a = b + t;
            if (a > na) break;
            const bool st = a == na || a == 0 || (keys[a] >> lowbits) != (keys[a - 1] >> lowbits);
            if (st) { atomicMin(&bound[which], t); break; }
        }
    }
    __syncthreads();
    if (bound[0] >= CS_STEP) return;                      // no class starts in this stripe
    const uint64_t s = w0 + bound[0];
    if (s >= na) return;
    if (bound[1] == 0xFFFFFFFFu) return;                  // the last class of the stripe runs on for more than a tile
    const uint64_t e = w0 + CS_STEP + bound[1];
    if (e - s > (uint64_t)max_range) return;
    const uint32_t n = (uint32_t)(e - s);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lt = lane ? (~0ULL >> (64 - lane)) : 0ULL;
    const uint32_t nit = (n + BLOCK - 1) / BLOCK;                 // pairs per thread this range needs (ranges are ~half a tile)
    const uint32_t base = (uint32_t)wave * (nit * WAVE) + lane;   // wave w owns the contiguous pairs [w * nit * 64, (w + 1) * nit * 64)
    for (uint32_t j = threadIdx.x; j < n; j += BLOCK) { skeys[j] = keys[s + j]; sidx[j] = (uint16_t)j; }
    __syncthreads();
    const K lomask = lowbits >= 64 ? ~(K)0 : (((K)1 << lowbits) - 1);
    const K hmin = skeys[0] >> lowbits, hspan = (skeys[n - 1] >> lowbits) - hmin;     // high parts are already in order
    const int nlo = (lowbits + 7) / 8;
    int nhi = 0; while (nhi < 8 && (hspan >> (8 * nhi))) ++nhi;
    for (int p = 0; p < nlo + nhi; ++p) {
        K k[ITEMS]; uint16_t v[ITEMS]; unsigned dg[ITEMS];
#pragma unroll
        for (int w = 0; w < BLOCK / WAVE; ++w) wh[w][threadIdx.x] = 0;
        __syncthreads();
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            if ((uint32_t)it < nit) {                                  // uniform; no break: the loop must unroll (register arrays)
            const uint32_t i = base + (uint32_t)it * WAVE;
            const bool valid = i < n;
            k[it] = valid ? skeys[i] : (K)0; v[it] = valid ? sidx[i] : (uint16_t)0;
            const unsigned d = p < nlo ? (unsigned)((k[it] & lomask) >> (8 * p)) & (RS_RADIX - 1) : (unsigned)(((k[it] >> lowbits) - hmin) >> (8 * (p - nlo))) & (RS_RADIX - 1);
            unsigned long long peers = __ballot(valid);
#pragma unroll
            for (int bb = 0; bb < 8; ++bb) {
                unsigned long long m = __ballot((d >> bb) & 1);
                peers &= ((d >> bb) & 1) ? m : ~m;
            }
            const int leader = valid ? (__ffsll((long long)peers) - 1) : lane;
            uint32_t old = 0;
            if (valid && lane == leader) { old = wh[wave][d]; wh[wave][d] = old + (uint32_t)__popcll(peers); }
            old = __shfl(old, leader);
            dg[it] = d | ((old + (uint32_t)__popcll(peers & lt)) << 8);        // digit and rank inside the wave
            }
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
            if ((uint32_t)it < nit && i < n) { const uint32_t li = wh[wave][dg[it] & 255u] + (dg[it] >> 8); skeys[li] = k[it]; sidx[li] = v[it]; }
        }
        __syncthreads();
    }
    for (uint32_t j = threadIdx.x; j < n; j += BLOCK) {
        const uint32_t src = sidx[j];
        okeys[s + j] = skeys[j]; ovals[s + j] = vals[s + src]; onj[s + j] = nj[s + src]; done[s + j] = 1;
    }
    if (threadIdx.x == 0) atomicAdd(nsorted, (unsigned long long)n);
}
__global__ __launch_bounds__(BLOCK) void k_not_done(const uint8_t *done, uint64_t n, uint32_t *flag)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) flag[i] = done[i] ? 0u : 1u;
}
template <typename K> __global__ __launch_bounds__(BLOCK) void k_gather_pairs(const K *keys, const uint32_t *vals, const uint32_t *idx, uint64_t n, K *ok, uint32_t *ov)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) { ok[i] = keys[idx[i]]; ov[i] = vals[idx[i]]; }
}
template <typename K> __global__ __launch_bounds__(BLOCK) void k_scatter_pairs(const K *keys, const uint32_t *vals, const uint32_t *idx, uint64_t n, K *ok, uint32_t *ov)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) { ok[idx[i]] = keys[i]; ov[idx[i]] = vals[i]; }
}
__global__ __launch_bounds__(BLOCK) void k_gather_u32(const uint32_t *in, const uint32_t *idx, uint64_t n, uint32_t *out)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) out[i] = in[idx[i]];
}
__global__ __launch_bounds__(BLOCK) void k_gather_jump(const uint32_t *vals, const uint32_t *idx, uint64_t n, const uint2 *rj, uint32_t *onj)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) onj[idx[i]] = rj[vals[i]].y;
}
// Sorts the na pairs of a round by (high part, low part); result in *sk / *sv.  nj[a] is the new jump of pair a: on the
// LDS route it travels with the pair (*snj = its sorted copy, the caller writes rank and jump together); when the plain
// radix sort is taken the jumps are applied to rj here, before the pair order is lost (*snj = nullptr).
inline int class_segment_sort(pfp_ctx *c, uint64_t *k0, uint32_t *v0, const uint32_t *nj, uint64_t *k1, uint32_t *v1, uint32_t *onj, uint64_t na, int lowbits, int rbits,
                              uint2 *rj, uint64_t **sk, uint32_t **sv, uint32_t **snj)
{
    BitRange rr = {0, lowbits + rbits};
    const unsigned ga = nblocks(na, BLOCK);
    *snj = nullptr;
    static const long long min_na = getenv("PFP_CLASS_SORT_MIN") ? atoll(getenv("PFP_CLASS_SORT_MIN")) : 8ll * RS_TILE;   // < 0: never (tests: 1 = always)
    if (min_na < 0 || na < (uint64_t)min_na) {
        PFP_LAUNCH(c, K_SS_MAKE_KEYS, na * 12, k_ss_apply_jump, ga, (const uint32_t *)v0, nj, na, rj);
        return radix_sort_pairs<uint64_t>(c, k0, v0, k1, v1, na, &rr, 1, sk, sv);
    }
    static const uint32_t max_range = getenv("PFP_CLASS_SORT_MAXRANGE") ? (uint32_t)atoi(getenv("PFP_CLASS_SORT_MAXRANGE")) : (uint32_t)RS_TILE;   // tests: smaller, to reach the large-class route
    const size_t mk = c->arena.mark_hi();
    uint8_t *done; unsigned long long *d_ns;
    PFP_ALLOC_HI(c, done, uint8_t, na); PFP_ALLOC_HI(c, d_ns, unsigned long long, 1);
    PFP_HIP(c, hipMemsetAsync(done, 0, na, c->stream));
    PFP_HIP(c, hipMemsetAsync(d_ns, 0, 8, c->stream));
    PFP_LAUNCH(c, K_RADIX_SCATTER, na * 64, (k_class_tile_sort<uint64_t>), nblocks(na, CS_STEP), (const uint64_t *)k0, (const uint32_t *)v0, nj, k1, v1, onj, na, lowbits, max_range, done, d_ns);
    unsigned long long ns = 0;
    PFP_HIP(c, hipMemcpyAsync(&ns, d_ns, 8, hipMemcpyDeviceToHost, c->stream));
    PFP_HIP(c, hipStreamSynchronize(c->stream));
    const uint64_t nl = na - ns;
    static const bool verbose = getenv("PFP_VERBOSE") != nullptr;
    if (verbose) fprintf(stderr, "[pfbwt_hip]   class sort: %llu pairs, %llu in classes too large for one tile%s\n", (unsigned long long)na, (unsigned long long)nl, nl > na / 2 ? " -> plain radix sort" : "");
    if (nl > na / 2) {            // mostly large classes: one plain sort of everything (k0 / v0 / nj are still intact)
        c->arena.release_hi(mk);
        PFP_LAUNCH(c, K_SS_MAKE_KEYS, na * 12, k_ss_apply_jump, ga, (const uint32_t *)v0, nj, na, rj);
        return radix_sort_pairs<uint64_t>(c, k0, v0, k1, v1, na, &rr, 1, sk, sv);
    }
    if (nl) {                      // the pairs of the large classes: collect, sort, put back (their positions are whole classes in order)
        uint32_t *flag, *pos, *idx, *d_cnt, *lnj; uint64_t *lk0, *lk1; uint32_t *lv0, *lv1;
        PFP_ALLOC_HI(c, flag, uint32_t, na); PFP_ALLOC_HI(c, pos, uint32_t, na); PFP_ALLOC_HI(c, idx, uint32_t, nl); PFP_ALLOC_HI(c, d_cnt, uint32_t, 1);
        PFP_ALLOC_HI(c, lk0, uint64_t, nl); PFP_ALLOC_HI(c, lk1, uint64_t, nl); PFP_ALLOC_HI(c, lv0, uint32_t, nl); PFP_ALLOC_HI(c, lv1, uint32_t, nl); PFP_ALLOC_HI(c, lnj, uint32_t, nl);
        PFP_LAUNCH(c, K_COMPACT, na * 5, k_not_done, ga, (const uint8_t *)done, na, flag);
        PFP_TRY(device_compact(c, nullptr, flag, na, idx, pos, d_cnt));
        PFP_LAUNCH(c, K_COMPACT, nl * 28, (k_gather_pairs<uint64_t>), nblocks(nl, BLOCK), (const uint64_t *)k0, (const uint32_t *)v0, (const uint32_t *)idx, nl, lk0, lv0);
        PFP_LAUNCH(c, K_COMPACT, nl * 12, k_gather_u32, nblocks(nl, BLOCK), nj, (const uint32_t *)idx, nl, lnj);
        PFP_LAUNCH(c, K_SS_MAKE_KEYS, nl * 12, k_ss_apply_jump, nblocks(nl, BLOCK), (const uint32_t *)lv0, (const uint32_t *)lnj, nl, rj);   // their jumps go to rj now, the sort loses the pairing
        uint64_t *lsk; uint32_t *lsv;
        PFP_TRY(radix_sort_pairs<uint64_t>(c, lk0, lv0, lk1, lv1, nl, &rr, 1, &lsk, &lsv));
        PFP_LAUNCH(c, K_COMPACT, nl * 28, (k_scatter_pairs<uint64_t>), nblocks(nl, BLOCK), (const uint64_t *)lsk, (const uint32_t *)lsv, (const uint32_t *)idx, nl, k1, v1);
        PFP_LAUNCH(c, K_COMPACT, nl * 16, k_gather_jump, nblocks(nl, BLOCK), (const uint32_t *)lsv, (const uint32_t *)idx, nl, (const uint2 *)rj, onj);
    }
    c->arena.release_hi(mk);
    *sk = k1; *sv = v1; *snj = onj;
    return PFP_OK;
}

// Sorts the N suffixes described by (keys,vals) [already filled: keys = h0-character prefixes, vals = x].
// Outputs SA (slot -> x) and rank (x -> slot of its class head).  ws/wordid select dictionary semantics
// (see k_ss_flag_active); D != nullptr additionally enables the run round (byte texts).
// keys/vals and their twins k1/v1 (N entries each) are scratch owned by the caller.
inline int suffix_sort_doubling(pfp_ctx *c, uint64_t N, uint64_t *k0, uint32_t *v0, uint64_t *k1, uint32_t *v1,
                                const BitRange *init_ranges, int n_init_ranges, uint32_t h0,
                                const uint32_t *ws, const uint32_t *wordid, const uint8_t *D, uint32_t *SA, uint2 *rj, int *rounds_out)
{
    const size_t mk = c->arena.mark_hi();
    uint32_t *head, *aux, *slots, *slots2, *d_cnt, *nj, *onj = nullptr, *M = nullptr;
    PFP_ALLOC_HI(c, head, uint32_t, N);
    PFP_ALLOC_HI(c, aux, uint32_t, N);
    PFP_ALLOC_HI(c, slots, uint32_t, N);
    PFP_ALLOC_HI(c, slots2, uint32_t, N);
    PFP_ALLOC_HI(c, d_cnt, uint32_t, 1);
    uint64_t *sk; uint32_t *sv;
    PFP_TRY(radix_sort_pairs<uint64_t>(c, k0, v0, k1, v1, N, init_ranges, n_init_ranges, &sk, &sv));
    const unsigned gN = nblocks(N, BLOCK);
    PFP_LAUNCH(c, K_SS_HEADS, N * 24, k_ss_heads, gN, (const uint64_t *)sk, (const uint32_t *)sv, (const uint32_t *)nullptr, N, SA, head, aux);
    PFP_TRY((device_scan<uint32_t, 1>(c, aux, aux, N, nullptr)));
    PFP_LAUNCH(c, K_SS_WRITE_RANK, N * 16, k_ss_init_rj, gN, (const uint32_t *)sv, (const uint32_t *)aux, N, h0, rj);
    // first active list
    uint32_t *keep = aux; // aux is free again after write_rank
    PFP_LAUNCH(c, K_SS_FLAG_ACTIVE, N * 16, k_ss_flag_active, gN, (const uint32_t *)sv, (const uint32_t *)head, N, (const uint2 *)rj, (const uint32_t *)nullptr, ws ? (const uint64_t *)sk : (const uint64_t *)nullptr, ws, wordid, keep);
    PFP_TRY(device_compact(c, nullptr, keep, N, slots, slots2, d_cnt));
    uint32_t na = 0; PFP_TRY(d2h_u32(c, d_cnt, &na));
    int rounds = 1;
    const int rbits = bits_for(N);
    if (na > 0) {
        PFP_ALLOC_HI(c, nj, uint32_t, na); PFP_ALLOC_HI(c, onj, uint32_t, na);
        if (D) {   // run lengths for the run round
            PFP_ALLOC_HI(c, M, uint32_t, N);
            PFP_LAUNCH(c, K_SS_MAKE_KEYS, N * 5, k_ss_runend_marks, gN, D, N, M);
            PFP_TRY((device_scan<uint32_t, 1>(c, M, M, N, nullptr)));
        }
    }
    static const bool verbose = getenv("PFP_VERBOSE") != nullptr;
    while (na > 0) {
        if (verbose) fprintf(stderr, "[pfbwt_hip] suffix sort N=%llu round %d: %u active\n", (unsigned long long)N, rounds, na);
        if (rounds > 64) return PFP_E_CORRUPT; // cannot happen on well-formed input
        const unsigned ga = nblocks(na, BLOCK);
        const bool run_round = (M != nullptr && rounds == 1);
        // build keys for the active list into k0/v0 (previous contents are dead: SA/rank/jump hold the state)
        PFP_LAUNCH(c, K_SS_MAKE_KEYS, (uint64_t)na * 36, k_ss_make_keys, ga, (const uint32_t *)slots, (const uint32_t *)SA, (const uint2 *)rj, (uint64_t)na, N,
                   run_round ? D : (const uint8_t *)nullptr, run_round ? (const uint32_t *)M : (const uint32_t *)nullptr, run_round ? 32 : rbits, k0, v0, nj);
        uint32_t *snj = nullptr;
        PFP_TRY(class_segment_sort(c, k0, v0, nj, k1, v1, onj, na, run_round ? 32 : rbits, rbits, rj, &sk, &sv, &snj));
        PFP_LAUNCH(c, K_SS_HEADS, (uint64_t)na * 28, k_ss_heads, ga, (const uint64_t *)sk, (const uint32_t *)sv, (const uint32_t *)slots, (uint64_t)na, SA, head, aux);
        PFP_TRY((device_scan<uint32_t, 1>(c, aux, aux, na, nullptr)));
        if (snj) PFP_LAUNCH(c, K_SS_WRITE_RANK, (uint64_t)na * 20, k_ss_write_rank_jump, ga, (const uint32_t *)sv, (const uint32_t *)aux, (const uint32_t *)snj, (uint64_t)na, rj);
        else PFP_LAUNCH(c, K_SS_WRITE_RANK, (uint64_t)na * 12, k_ss_write_rank, ga, (const uint32_t *)sv, (const uint32_t *)aux, (uint64_t)na, rj);
        ++rounds;
        PFP_LAUNCH(c, K_SS_FLAG_ACTIVE, (uint64_t)na * 16, k_ss_flag_active, ga, (const uint32_t *)sv, (const uint32_t *)head, (uint64_t)na, (const uint2 *)rj, (const uint32_t *)snj, (const uint64_t *)nullptr, ws, wordid, keep);
        PFP_TRY(device_compact(c, slots, keep, na, slots2, head /*pos scratch*/, d_cnt));
        uint32_t *t = slots; slots = slots2; slots2 = t;
        PFP_TRY(d2h_u32(c, d_cnt, &na));
    }
    if (rounds_out) *rounds_out = rounds;
    c->arena.release_hi(mk);
    return PFP_OK;
}

} // namespace pfp
