// pfbwt-f_amd/csrc/sufsort.h -- suffix sorting on the device by prefix refinement (initial radix sort, then
// rounds that order every class of equal prefixes by the ranks of one -- or three: the covered prefix grows 4x --
// further prefixes, inside LDS) with active-set filtering.  Two uses:
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

constexpr int DK_CHARS = 17;          // characters per initial key: 9^17 < 2^54 still sorts in seven 8-bit passes (round 3; 16 before: on a random-like text of 3.4 G positions 18 % instead of 55 % of the suffixes share their key with another and enter the first round)
constexpr int DK_PER_THREAD = 16;     // suffixes per thread
constexpr int DK_TILE = BLOCK * DK_PER_THREAD;

// keys[x] = the first DK_CHARS characters of suffix x read as a base-9 number (digits = dict_code, 0 after
// the terminator): order-preserving like 4-bit packing, but 9^17 < 2^54, i.e. 7 radix passes for 17 characters (4-bit codes: 16 in 8).
constexpr int DK_KEY_BITS = 54;
__global__ __launch_bounds__(BLOCK) void k_dict_init_keys(const uint8_t *D, uint64_t dsize, uint64_t *keys, uint32_t *vals)
{
    __shared__ uint8_t tile[DK_TILE + DK_CHARS];
    __shared__ uint8_t prevc;
    const uint64_t t0 = (uint64_t)blockIdx.x * DK_TILE;
    for (uint32_t i = threadIdx.x; i < DK_TILE + DK_CHARS; i += BLOCK) {
        const uint64_t x = t0 + i;
        tile[i] = x < dsize ? (uint8_t)dict_code(D[x]) : (uint8_t)0;
    }
    if (threadIdx.x == 0) prevc = t0 ? (uint8_t)dict_code(D[t0 - 1]) : (uint8_t)EndOfWord;
    __syncthreads();
    // consecutive threads take consecutive positions (conflict-free LDS reads, coalesced stores)
#pragma unroll 1
    for (int k = 0; k < DK_PER_THREAD; ++k) {
        const uint32_t l = (uint32_t)k * BLOCK + threadIdx.x;
        uint64_t key = 0; bool stop = false; uint32_t off = DK_CHARS;
#pragma unroll
        for (int j = 0; j < DK_CHARS; ++j) {
            const uint32_t cc = tile[l + j];
            key = key * 9u + (stop ? 0u : cc);
            if (!stop && cc <= 1) off = (uint32_t)j + 1u;
            stop = stop || cc <= 1;                      // the terminator itself is part of the key, nothing after it
        }
        // bits 56..60: length of the prefix the key covers = jump offset (never past the byte behind the terminator)
        const uint64_t x = t0 + l;
        // bit 62 (outside every sort range): x starts a word -- the byte in front is an EndOfWord (or x == 0) and x is not the EndOfDict
        const uint64_t wstart = ((l ? tile[l - 1] : prevc) == EndOfWord && tile[l] != EndOfDict) ? 1ULL : 0ULL;
        if (x < dsize) { keys[x] = key | ((uint64_t)off << 56) | (wstart << 62); vals[x] = (uint32_t)x; }
    }
}

// The same for ANY byte alphabet (the gsacak drop-in called with strings that are not dictionaries of this parser): code[] maps the
// bytes that occur to 0 (terminator), 1 (separator), 2, 3, ... in byte order, sigma = number of codes, the key holds the first
// `chars` characters as a base-sigma number (sigma^chars < 2^51, chars <= 16).  One thread per suffix: this path is not timed.
__global__ __launch_bounds__(BLOCK) void k_dict_init_keys_any(const uint8_t *D, uint64_t dsize, const uint8_t *code, uint32_t sigma, uint32_t chars, uint64_t *keys, uint32_t *vals)
{
    const uint64_t x = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (x >= dsize) return;
    uint64_t key = 0; bool stop = false; uint32_t off = chars;
    for (uint32_t j = 0; j < chars; ++j) {
        const uint32_t cc = x + j < dsize ? code[D[x + j]] : 0u;
        key = key * sigma + (stop ? 0u : cc);
        if (!stop && cc <= 1) off = j + 1u;
        stop = stop || cc <= 1;
    }
    keys[x] = key | ((uint64_t)off << 56); vals[x] = (uint32_t)x;
}

// parse keys: (S[x], S[x+1]) with S = ranks + [0]  (pfparser.hpp:407-410)
__global__ __launch_bounds__(BLOCK) void k_int_init_keys(const uint32_t *S, uint64_t N, int symbits, int nsym, uint64_t *keys, uint32_t *vals)
{
    const uint64_t x = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (x >= N) return;
    uint64_t k = ((uint64_t)S[x] << symbits) | (x + 1 < N ? S[x + 1] : 0u);      // packed tightly: 2 * symbits key bits (5 radix passes for 20-bit symbols, not 6)
    if (nsym == 3) k = (k << symbits) | (x + 2 < N ? S[x + 2] : 0u);             // three symbols where they fit 64 bits
    keys[x] = k;
    vals[x] = (uint32_t)x;
}

// after the initial sort of all N suffixes: head flags, SA, head slot for the max-scan
__global__ __launch_bounds__(BLOCK) void k_ss_heads(const uint64_t *keys, const uint32_t *vals, uint64_t na, uint64_t keymask, uint32_t *SA, uint32_t *head, uint32_t *headslot, uint8_t *sflag /*nullable: per slot, bit 62 of its key (the suffix starts a word)*/)
{
    const uint64_t a = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (a >= na) return;
    if (sflag) sflag[a] = (uint8_t)((keys[a] >> 62) & 1ULL);
    const uint32_t hd = (a == 0 || (keys[a] & keymask) != (keys[a - 1] & keymask)) ? 1u : 0u;
    SA[a] = vals[a];
    head[a] = hd;
    headslot[a] = hd ? (uint32_t)a : 0u;
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

// longest run of equal bytes (from the scanned run-end marks M): without a long run the run round is skipped -- it is a K = 1 round
// for every suffix that is not inside a run, i.e. on a dictionary without runs it only delays the quadrupling rounds
__global__ __launch_bounds__(BLOCK) void k_max_run_length(const uint32_t *M, uint64_t N, uint32_t *out)
{
    __shared__ uint32_t red[4];
    // 16 positions per thread; a workgroup touches the result only when it would raise it (one atomic per workgroup on one address cost
    // 150 ms over the 3.4 G positions of S-3G)
    const uint64_t i0 = ((uint64_t)blockIdx.x * BLOCK + threadIdx.x) * 16;
    uint32_t mx = 0;
    for (int k = 0; k < 16; ++k) { const uint64_t ip = i0 + k; if (ip < N) { const uint32_t d = (uint32_t)ip - M[ip] + 1u; mx = d > mx ? d : mx; } }
    uint32_t tot;
    (void)block_incl_max(mx, red, &tot);
    if (threadIdx.x == 0 && tot > *(volatile uint32_t *)out) atomicMax(out, tot);
}
constexpr uint32_t RUN_ROUND_MIN_RUN = 256;      // a run of d equal bytes costs ~log2(d / 16) extra doubling rounds for its few suffixes; the run round costs one K = 1 round for ALL

// ---- state of the refinement -------------------------------------------------------------------------------------
//   SA[slot] = x                          suffixes in the order found so far; a class = a contiguous range of slots
//   int alphabet:  rank[x]                slot of the head of x's class (jumps are uniform there: the suffix one covered
//                                         prefix on is x + h)
//   dictionary:    rj[x] = {rank, jump}   jump = end of the prefix [x, jump) the rank orders x by; never past the first
//                                         byte after the word's EndOfWord, so "the covered prefix contains the
//                                         terminator" (the class is a group of identical suffixes, final) is
//                                         D[jump - 1] == EndOfWord
//   active list, in slot order:  aslot[a], arnk[a] = rank of its class (= slot of the class head), ajmp[a] (dictionary)
// One round = k_round (sort inside every class by the ranks of the next K covered prefixes, new heads, new ranks, SA;
// the pairs cross HBM once) + k_round_apply (ranks that changed are scattered to rank[] / rj[] only now -- a round must
// read the ranks of ONE state --, the list of classes that still have to be refined is compacted per stripe).
//
// K = 3 ("quadrupling"): what bounds a round is the random gather of the second key -- one 128-byte line per pair for 4
// useful bytes (PMC: 165 B of HBM traffic per pair) -- and the scatter of the new ranks; the LDS sort costs ~1.5 ms per
// 8-bit pass and 325 M pairs (VALU-bound: the eight ballots per item; measured: K = 1 rounds 14.3 ms for 6 passes, K = 3
// rounds 26.5 ms for 14), so three ranks per round cost less than two rounds of one and halve the scatters.  A pass in
// TEXT order (coalesced) therefore first builds T[y] = {rank of y, of the suffix one covered prefix behind y, of the one
// two behind; (dictionary) the jump behind the third}: ONE 16-byte gather per pair then orders the class by three further
// prefixes, the covered prefix grows 4x per round and the rounds (gathers, scatters, compactions) halve.  When few pairs
// are left the table is not worth a pass over all N positions: the kernel then follows the chain itself.
constexpr uint8_t RF_KEEP = 1, RF_CHANGED = 2, RF_DONE = 4;
#ifndef PFP_K3_ITEMS
#define PFP_K3_ITEMS 9
#endif
// pairs per thread of the class-sort tile: 15 -> 3840 pairs, 49 KiB of LDS with 8-byte keys (three workgroups per CU);
// 9 -> 2304 pairs with the 16-byte keys of K = 3 (50 KiB)
template <int K> struct RoundCfg { static constexpr int ITEMS = K == 1 ? RS_ITEMS : PFP_K3_ITEMS, TILE = BLOCK * ITEMS; static constexpr uint32_t STEP = TILE / 2; };

__global__ __launch_bounds__(BLOCK) void k_not_done(const uint8_t *done, uint64_t n, uint32_t *flag)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) flag[i] = done[i] ? 0u : 1u;
}
// first class start at or after position b of the active list (arnk ascends along the list: the end of a class is found
// by bisection, however many members it has)
__device__ __forceinline__ uint64_t class_start_at_or_after(const uint32_t *arnk, uint64_t na, uint64_t b)
{
    if (b == 0 || b >= na) return b < na ? b : na;
    const uint32_t g = arnk[b - 1];
    if (arnk[b] != g) return b;
    uint64_t lo = b, hi = na;                              // first position in [b, na) with arnk > g
    while (lo < hi) { const uint64_t mid = lo + ((hi - lo) >> 1); if (arnk[mid] <= g) lo = mid + 1; else hi = mid; }
    return lo;
}

// Text rounds (dictionaries of non-repetitive collections): the next TK_CHARS characters behind the covered prefix, read from the
// dictionary itself, order a class -- no rank of any other suffix is consulted, so the 8 bytes per dictionary offset of rj[] are
// neither scattered after the initial sort (S-3G: 3.4 G random 8-byte writes, 126 ms) nor after every round.  Same number system
// as the initial keys (nothing behind the terminator counts); *nx = end of the prefix the key covers.
constexpr int TK_CHARS = 10;                                   // 9^10 < 2^32: the key is the `low` word of a K = 1 round
__device__ __forceinline__ uint32_t text_key(const uint8_t *D, uint64_t N, uint64_t y, uint32_t *nx)
{
    // the characters y .. y+9 from three aligned 8-byte loads (measured on S-3G: ten byte loads per pair made the round 40 % slower
    // than the 8-byte rank gather it replaces).  The dictionary buffer is 8-byte aligned and ends in EndOfDict, behind which
    // nothing is looked at (stop), so what the loads pick up beyond it does not matter; loads past the buffer's slack are skipped.
    const uint64_t base = y & ~7ULL, lim = N + 16;
    const uint64_t w0 = *reinterpret_cast<const uint64_t *>(D + base);
    const uint64_t w1 = base + 16 <= lim ? *reinterpret_cast<const uint64_t *>(D + base + 8) : 0ULL;
    const uint64_t w2 = base + 24 <= lim ? *reinterpret_cast<const uint64_t *>(D + base + 16) : 0ULL;
    const unsigned sh = (unsigned)(y & 7ULL) * 8u;
    const uint64_t lo = sh ? (w0 >> sh) | (w1 << (64u - sh)) : w0;          // bytes y .. y+7
    const uint64_t hi = sh ? (w1 >> sh) | (w2 << (64u - sh)) : w1;          // bytes y+8 .. y+15
    uint32_t key = 0, off = TK_CHARS; bool stop = false;
#pragma unroll
    for (int j = 0; j < TK_CHARS; ++j) {
        const uint32_t cc = dict_code((uint32_t)((j < 8 ? lo >> (8 * j) : hi >> (8 * (j - 8))) & 0xFFu));
        key = key * 9u + (stop ? 0u : cc);
        if (!stop && cc <= 1) off = (uint32_t)j + 1u;
        stop = stop || cc <= 1;
    }
    const uint64_t e = y + off;
    *nx = (uint32_t)(e < N ? e : N);
    return key;
}
// the three ranks behind y (and the jump behind the third), straight from the state: the chain stops at the prefix that
// holds the word's terminator (nothing behind it is compared; both members of a tie stop at the same link)
template <bool DICT> __device__ __forceinline__ uint4 chain3(const uint32_t *rank, const uint2 *rj, uint64_t N, uint64_t y, uint32_t h, const uint8_t *D)
{
    if (!DICT) {
        const uint64_t y2 = y + h, y3 = y2 + h;
        return make_uint4(y < N ? rank[y] : 0u, y2 < N ? rank[y2] : 0u, y3 < N ? rank[y3] : 0u, 0u);
    }
    uint4 t = make_uint4(0u, 0u, 0u, (uint32_t)N);
    if (y >= N) return t;
    uint2 q = rj[y]; t.x = q.x; t.w = q.y;
    if (q.y >= N || D[q.y - 1] == EndOfWord) return t;
    q = rj[q.y]; t.y = q.x; t.w = q.y;
    if (q.y >= N || D[q.y - 1] == EndOfWord) return t;
    q = rj[q.y]; t.z = q.x; t.w = q.y;
    return t;
}
// T[y] for every position, in text order (the chain of a dictionary position stays inside its word: neighbouring lines)
template <bool DICT> __global__ __launch_bounds__(BLOCK) void k_round_table(const uint32_t *rank, const uint2 *rj, uint64_t N, uint32_t h, const uint8_t *D, uint4 *T)
{
    const uint64_t y = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (y < N) T[y] = chain3<DICT>(rank, rj, N, y, h, D);
}

// Workgroup j owns the classes that START in [j*STEP, (j+1)*STEP) of the active list; they end where the first class of
// the next stripe starts (at most one tile of pairs; a longer range -- a class with thousands of members -- is left
// alone: its flags stay 0 and the pairs go through the global radix sort, k_round_keys / k_round_finish).
// Sort keys in LDS: A = (class - first class of the range) << lowbits | first rank; K = 3: B = second rank << lowbits |
// third rank.
template <bool DICT, int K> __global__ __launch_bounds__(BLOCK) void k_round(const uint32_t *aslot, const uint32_t *arnk, const uint32_t *ajmp, uint64_t na, uint64_t N,
                                                                             uint32_t *SA, const uint32_t *rank, const uint2 *rj, const uint4 *T /*K = 3; null: follow the chain*/, uint32_t h, const uint8_t *D,
                                                                             const uint32_t *M /*run round (K = 1)*/, uint32_t run_min /*characters per initial key*/, int lowbits, uint32_t max_range, uint32_t *newr,
                                                                             uint32_t *xout /*the sorted suffixes, in list order*/, uint32_t *tnj, uint32_t *newj, uint8_t *flags, uint32_t *stripe_keep, unsigned long long *ndone,
                                                                             uint32_t *srank /*nullable: per SLOT, head slot of its class*/, uint8_t *sflag /*nullable: per slot flag that moves with its suffix*/, int textkeys)
{
    constexpr int ITEMS = RoundCfg<K>::ITEMS, TILE = RoundCfg<K>::TILE;
    constexpr uint32_t STEP = RoundCfg<K>::STEP;
    __shared__ uint32_t wh[BLOCK / WAVE][RS_RADIX];
    __shared__ uint64_t skeys[TILE];
    __shared__ uint64_t skeyb[K == 3 ? TILE : 1];
    __shared__ uint16_t sidx[TILE];                      // payload of the LDS sort: index of the pair in the range
    __shared__ uint16_t shp[TILE + 1];                   // position (in the sorted range) of the head of every pair's new class
    __shared__ unsigned long long red[4];
    __shared__ uint64_t bound[2];
    const uint64_t w0 = (uint64_t)blockIdx.x * STEP;
    if (threadIdx.x < 2) bound[threadIdx.x] = class_start_at_or_after(arnk, na, w0 + (uint64_t)threadIdx.x * STEP);
    __syncthreads();
    const uint64_t s = bound[0], e = bound[1];
    if (s >= w0 + STEP || s >= na) return;                // no class starts in this stripe
    if (e - s > (uint64_t)max_range) return;              // a class with more members than a tile holds
    const uint32_t n = (uint32_t)(e - s);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t nit = (n + BLOCK - 1) / BLOCK;
    const uint32_t base = (uint32_t)wave * (nit * WAVE) + lane;   // wave w owns the contiguous pairs [w * nit * 64, (w + 1) * nit * 64)
    const uint32_t hmin = arnk[s];
    // ---- keys.  Dependent gathers per pair (list entry -> SA[slot] -> ranks of the target): all of a thread's pairs go
    //      through each stage together, so that a tile pays the memory latencies once, not once per pair
    uint32_t xi[ITEMS];                                  // the suffixes of this thread's pairs (by position before the sort)
    {
        uint32_t sl[ITEMS], lowv[ITEMS], njv[ITEMS], yj[ITEMS];
        uint32_t r2[K == 3 ? ITEMS : 1], r3[K == 3 ? ITEMS : 1];
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const uint32_t j = threadIdx.x + (uint32_t)it * BLOCK;
            sl[it] = j < n ? aslot[s + j] : 0u;
            yj[it] = (DICT && j < n) ? ajmp[s + j] : 0u;
        }
        uint8_t fi[ITEMS];
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) { const uint32_t j = threadIdx.x + (uint32_t)it * BLOCK; xi[it] = j < n ? SA[sl[it]] : 0u; fi[it] = (DICT && sflag && j < n) ? sflag[sl[it]] : (uint8_t)0; }
        if (K == 3) {
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                const uint32_t j = threadIdx.x + (uint32_t)it * BLOCK;
                const uint64_t y = DICT ? (uint64_t)yj[it] : (uint64_t)xi[it] + h;
                uint4 t = make_uint4(0u, 0u, 0u, (uint32_t)N);
                if (j < n) { if (T) { if (y < N) t = T[y]; } else t = chain3<DICT>(rank, rj, N, y, h, D); }
                lowv[it] = t.x; r2[it] = t.y; r3[it] = t.z; njv[it] = t.w;
            }
        } else if (!DICT) {
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                const uint32_t j = threadIdx.x + (uint32_t)it * BLOCK;
                const uint64_t y = (uint64_t)xi[it] + h;
                lowv[it] = (j < n && y < N) ? rank[y] : 0u; njv[it] = 0;
            }
        } else if (!M) {
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                const uint32_t j = threadIdx.x + (uint32_t)it * BLOCK;
                if (textkeys) { uint32_t nx = (uint32_t)N; lowv[it] = (j < n && yj[it] < N) ? text_key(D, N, yj[it], &nx) : 0u; njv[it] = nx; }
                else { const uint2 Q = (j < n && yj[it] < N) ? rj[yj[it]] : make_uint2(0u, (uint32_t)N); lowv[it] = Q.x; njv[it] = Q.y; }
            }
        } else {
            // run round: c^d a... is ordered among the suffixes that start with c by t = d if a < c else 2^32-1-d, jump = x + d
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                const uint32_t j = threadIdx.x + (uint32_t)it * BLOCK;
                uint32_t low = 0, nx = 0;
                if (j < n) {
                    const uint32_t x = xi[it];
                    bool run = false;
                    const uint32_t ip = (uint32_t)(N - 1 - x);
                    const uint32_t d = ip - M[ip] + 1u;
                    if (d >= run_min && D[x] > EndOfWord) {
                        const uint64_t en = (uint64_t)x + d;
                        const uint8_t nxt = en < N ? D[en] : (uint8_t)0;
                        low = nxt < D[x] ? d : 0xFFFFFFFFu - d; nx = (uint32_t)(en < N ? en : N); run = true;
                    }
                    if (!run) {
                        const uint32_t y = yj[it];
                        if (textkeys) { nx = (uint32_t)N; low = y < N ? text_key(D, N, y, &nx) : 0u; }
                        else { const uint2 Q = y < N ? rj[y] : make_uint2(0u, (uint32_t)N); low = Q.x; nx = Q.y; }
                    }
                }
                lowv[it] = low; njv[it] = nx;
            }
        }
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const uint32_t j = threadIdx.x + (uint32_t)it * BLOCK;
            if (j < n) {
                if (DICT) tnj[s + j] = njv[it];
                skeys[j] = ((uint64_t)(arnk[s + j] - hmin) << lowbits) | lowv[it]; sidx[j] = (uint16_t)(j | ((uint32_t)fi[it] << 15));      // (a tile holds fewer than 2^15 pairs: bit 15 carries the suffix's flag through the sort)
                if (K == 3) skeyb[j] = ((uint64_t)r2[it] << lowbits) | r3[it];
            }
        }
    }
    __syncthreads();
    // ---- LSD radix sort of the range inside LDS: (K = 3: the two ranks of B,) the low part of A, then the span of the
    //      (already ordered) class part.  This loop is what bounds the kernel (VALU: ~1.5 ms per pass and 325 M pairs when
    //      every pair was predicated on "is it one of the n" and the digit's place was worked out per item), hence:
    //      the range is padded to whole waves with all-ones keys (they stay behind the real ones in a stable sort, so no
    //      item needs a validity test), the place of a pass's digit is three scalars, and the lanes that hold the same
    //      digit come from same_digit_lanes (prims.h).  Measured: 26.5 -> 24 ms for a K = 3 round over 325 M pairs.
    const uint64_t hspan = skeys[n - 1] >> lowbits;
    for (uint32_t j = n + threadIdx.x; j < nit * BLOCK; j += BLOCK) { skeys[j] = ~0ULL; sidx[j] = (uint16_t)j; if (K == 3) skeyb[j] = ~0ULL; }
    const int nb = K == 3 ? (2 * lowbits + 7) / 8 : 0;
    const int nlo = (lowbits + 7) / 8;
    int nhi = 0; while (nhi < 8 && (hspan >> (8 * nhi))) ++nhi;
    for (int p = 0; p < nb + nlo + nhi; ++p) {
        const bool use_b = p < nb;
        const int q = p - nb;
        const int sh = use_b ? 8 * p : q < nlo ? 8 * q : lowbits + 8 * (q - nlo);
        const uint32_t dmask = (!use_b && q < nlo && lowbits - 8 * q < 8) ? ((1u << (lowbits - 8 * q)) - 1u) : 255u;      // the top digit of the low part stops where the class part starts
        uint64_t k[ITEMS], kb[K == 3 ? ITEMS : 1]; uint16_t v[ITEMS]; unsigned dg[ITEMS];
#pragma unroll
        for (int w = 0; w < BLOCK / WAVE; ++w) wh[w][threadIdx.x] = 0;
        __syncthreads();
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            if ((uint32_t)it < nit) {                                  // uniform; no break: the loop must unroll (register arrays)
            const uint32_t i = base + (uint32_t)it * WAVE;
            k[it] = skeys[i]; v[it] = sidx[i];
            if (K == 3) kb[it] = skeyb[i];
            const uint32_t d = (uint32_t)(((K == 3 && use_b) ? kb[K == 3 ? it : 0] : k[it]) >> sh) & dmask;
            uint32_t plo = 0xFFFFFFFFu, phi = 0xFFFFFFFFu;            // lanes with the same digit
            same_digit_lanes(d, plo, phi);
            const int leader = plo ? __builtin_ctz(plo) : 32 + __builtin_ctz(phi);       // never empty: the lane itself
            uint32_t old = 0;
            if (lane == leader) { old = wh[wave][d]; wh[wave][d] = old + (uint32_t)__builtin_popcount(plo) + (uint32_t)__builtin_popcount(phi); }
            old = __shfl(old, leader);
            dg[it] = d | ((old + __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u))) << 8);        // digit and rank inside the wave
            }
        }
        __syncthreads();
        {
            const unsigned d = threadIdx.x;
            uint32_t cw[BLOCK / WAVE]; uint32_t total = 0;
#pragma unroll
            for (int w = 0; w < BLOCK / WAVE; ++w) { cw[w] = wh[w][d]; total += cw[w]; }
            uint32_t tt;
            uint32_t run = block_excl_sum(total, reinterpret_cast<uint32_t *>(red), &tt);
#pragma unroll
            for (int w = 0; w < BLOCK / WAVE; ++w) { wh[w][d] = run; run += cw[w]; }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            if ((uint32_t)it < nit) {
                const uint32_t li = wh[wave][dg[it] & 255u] + (dg[it] >> 8);
                skeys[li] = k[it]; sidx[li] = v[it];
                if (K == 3) skeyb[li] = kb[K == 3 ? it : 0];
            }
        }
        __syncthreads();
    }
    // ---- new classes: pair j heads a class iff its key differs from its predecessor's; head position of every pair by a
    //      max-scan (a thread scans a contiguous chunk, the chunk maxima go through the block scan)
    const uint32_t c0 = threadIdx.x * nit, c1 = (c0 + nit < n) ? c0 + nit : n;
    uint32_t runmax = 0;
    for (uint32_t j = c0; j < c1; ++j) {
        const bool hd = j == 0 || skeys[j] != skeys[j - 1] || (K == 3 && skeyb[j] != skeyb[j - 1]);
        runmax = hd ? j : runmax;
        shp[j] = (uint16_t)runmax;                       // exact only behind the first head of the chunk; fixed below
    }
    uint32_t tot;
    const uint32_t inc = block_incl_max(runmax, reinterpret_cast<uint32_t *>(red), &tot);
    uint32_t prev = __shfl_up(inc, 1);
    __shared__ uint32_t wmax[BLOCK / WAVE];
    if (lane == 63) wmax[wave] = inc;
    __syncthreads();
    if (lane == 0) prev = wave ? wmax[wave - 1] : 0u;     // maximum over the threads in front (0 for thread 0: pair 0 is a head)
    for (uint32_t j = c0; j < c1; ++j) { const uint32_t v = shp[j]; if (v >= prev && v != 0) break; shp[j] = (uint16_t)(v > prev ? v : prev); }
    if (threadIdx.x == 0) shp[n] = (uint16_t)n;           // sentinel: "the pair behind the last one heads a class"
    __syncthreads();
    // the keys are not needed any more: their LDS holds the suffixes now, indexed by position before the sort, so that the
    // output below does not gather SA a second time
    uint32_t *sx = reinterpret_cast<uint32_t *>(skeys);
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) { const uint32_t j = threadIdx.x + (uint32_t)it * BLOCK; if (j < n) sx[j] = xi[it]; }
    __syncthreads();
    // ---- output: all gathers first (the old SA of the range's slots is overwritten below; loads of all of a thread's pairs
    //      are in flight together), then the stores
    uint32_t xs[ITEMS], slo[ITEMS], nrv[ITEMS], njo[ITEMS], oldr[ITEMS]; uint8_t flg[ITEMS], fout[ITEMS]; unsigned long long keepn = 0;     // kept pairs per position stripe (the range touches at most three), 20 bits each
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const uint32_t j = threadIdx.x + (uint32_t)it * BLOCK;
        const bool ok = j < n;
        const uint32_t sraw = ok ? sidx[j] : 0u, src = sraw & 0x7FFFu, hp = ok ? shp[j] : 0u;
        fout[it] = (uint8_t)(sraw >> 15);
        xs[it] = ok ? sx[src] : 0u;
        slo[it] = ok ? aslot[s + j] : 0u;
        nrv[it] = ok ? aslot[s + hp] : 0u;
        oldr[it] = ok ? arnk[s + j] : 0u;
        njo[it] = (DICT && ok) ? tnj[s + src] : 0u;
        const bool single = ok && hp == j && shp[j + 1] == j + 1;
        flg[it] = ok ? (uint8_t)(RF_DONE | (single ? 0 : RF_KEEP)) : (uint8_t)0;
    }
    if (DICT) {
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            if (flg[it]) {
                const bool fin = njo[it] >= N || D[njo[it] - 1] == EndOfWord;     // the covered prefix now holds the word's terminator: a group of identical suffixes
                if (fin) flg[it] &= (uint8_t)~RF_KEEP;
                flg[it] |= RF_CHANGED;                                           // the jump changes every round
            }
        }
    } else {
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) if (flg[it] && nrv[it] != oldr[it]) flg[it] |= RF_CHANGED;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const uint32_t j = threadIdx.x + (uint32_t)it * BLOCK;
        if (j < n) {
            if (flg[it] & RF_KEEP) keepn += 1ULL << (20 * (uint32_t)((s + j) / STEP - blockIdx.x));
            if (DICT) newj[s + j] = njo[it];
            SA[slo[it]] = xs[it]; newr[s + j] = nrv[it]; flags[s + j] = flg[it]; xout[s + j] = xs[it];
            if (srank) srank[slo[it]] = nrv[it];
            if (DICT && sflag) sflag[slo[it]] = fout[it];
        }
    }
    unsigned long long kt;
    (void)block_excl_sum(keepn, red, &kt);
    if (threadIdx.x == 0) {
        for (int t = 0; t < 3; ++t) { const uint32_t cnt = (uint32_t)(kt >> (20 * t)) & 0xFFFFFu; if (cnt) atomicAdd(&stripe_keep[blockIdx.x + t], cnt); }
        atomicAdd(ndone, (unsigned long long)n);
    }
}

// pairs of the ranges k_round left alone (idx = their positions in the active list): keys for the global radix sort, the
// payload is the pair's index i in this subset; ux / tnj keep its suffix and new jump.  K = 3: the two ranks of kb are
// sorted first, then (stable) ka.
template <bool DICT, int K> __global__ __launch_bounds__(BLOCK) void k_round_keys(const uint32_t *idx, uint64_t nl, const uint32_t *aslot, const uint32_t *arnk, const uint32_t *ajmp, uint64_t N,
                                                                                  const uint32_t *SA, const uint32_t *rank, const uint2 *rj, const uint4 *T, uint32_t h, const uint8_t *D, const uint32_t *M, uint32_t run_min, int lowbits,
                                                                                  uint64_t *ka, uint64_t *kb, uint32_t *ux, uint32_t *tnj, const uint8_t *sflag, uint8_t *uf, int textkeys)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= nl) return;
    const uint32_t a = idx[i];
    const uint32_t x = SA[aslot[a]];
    if (DICT && sflag) uf[i] = sflag[aslot[a]];
    uint32_t low, nx = 0;
    if (K == 3) {
        const uint64_t y = DICT ? (uint64_t)ajmp[a] : (uint64_t)x + h;
        const uint4 t = T ? (y < N ? T[y] : make_uint4(0u, 0u, 0u, (uint32_t)N)) : chain3<DICT>(rank, rj, N, y, h, D);
        low = t.x; nx = t.w; kb[i] = ((uint64_t)t.y << lowbits) | t.z;
    } else if (!DICT) { const uint64_t y = (uint64_t)x + h; low = y < N ? rank[y] : 0u; }
    else {
        bool run = false;
        if (M) {
            const uint32_t ip = (uint32_t)(N - 1 - x);
            const uint32_t d = ip - M[ip] + 1u;
            if (d >= run_min && D[x] > EndOfWord) {
                const uint64_t en = (uint64_t)x + d;
                const uint8_t nxt = en < N ? D[en] : (uint8_t)0;
                low = nxt < D[x] ? d : 0xFFFFFFFFu - d; nx = (uint32_t)(en < N ? en : N); run = true;
            }
        }
        if (!run) {
            const uint32_t y = ajmp[a];
            if (textkeys) { nx = (uint32_t)N; low = y < N ? text_key(D, N, y, &nx) : 0u; }
            else { const uint2 Q = y < N ? rj[y] : make_uint2(0u, (uint32_t)N); low = Q.x; nx = Q.y; }
        }
    }
    if (DICT) tnj[a] = nx;
    ka[i] = ((uint64_t)arnk[a] << lowbits) | low; ux[i] = x;
}
__global__ __launch_bounds__(BLOCK) void k_copy_keys_iota(const uint64_t *src, uint64_t n, uint64_t *keys, uint32_t *vals)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) { keys[i] = src[i]; vals[i] = (uint32_t)i; }
}
// second sort of K = 3: the pairs, ordered by kb, get their ka as key (in place of the sorted kb)
__global__ __launch_bounds__(BLOCK) void k_gather_keys(const uint64_t *ka, const uint32_t *perm, uint64_t n, uint64_t *keys)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) keys[i] = ka[perm[i]];
}
__global__ __launch_bounds__(BLOCK) void k_round_subset_heads(const uint64_t *keys /*sorted ka*/, const uint64_t *kb /*by subset index; null for K = 1*/, const uint32_t *perm, uint64_t nl, uint32_t *headidx)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= nl) return;
    bool hd = i == 0 || keys[i] != keys[i - 1];
    if (!hd && kb) hd = kb[perm[i]] != kb[perm[i - 1]];
    headidx[i] = hd ? (uint32_t)i : 0u;
}
// the sorted subset goes back to its positions (whole classes, in order): same outputs as k_round
template <bool DICT> __global__ __launch_bounds__(BLOCK) void k_round_finish(const uint32_t *perm, const uint32_t *ux, const uint32_t *tnj, const uint32_t *headidx /*max-scanned*/, const uint32_t *idx, uint64_t nl, uint64_t N,
                                                                             const uint32_t *aslot, const uint32_t *arnk, uint32_t *SA, const uint8_t *D, uint32_t step,
                                                                             uint32_t *newr, uint32_t *xout, uint32_t *newj, uint8_t *flags, uint32_t *stripe_keep, uint32_t *srank, uint8_t *sflag, const uint8_t *uf)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= nl) return;
    const uint32_t a = idx[i], hi = headidx[i], src = perm[i], x = ux[src];
    const bool single = hi == (uint32_t)i && (i + 1 == nl || headidx[i + 1] == (uint32_t)(i + 1));
    const uint32_t nr = aslot[idx[hi]];
    bool keep = !single;
    uint8_t fl = RF_DONE;
    if (DICT) {
        const uint32_t nj = tnj[idx[src]];
        newj[a] = nj;
        keep = keep && !(nj >= N || D[nj - 1] == EndOfWord);
        fl |= RF_CHANGED;
    } else if (nr != arnk[a]) fl |= RF_CHANGED;
    if (keep) fl |= RF_KEEP;
    SA[aslot[a]] = x; newr[a] = nr; flags[a] = fl; xout[a] = x;
    if (srank) srank[aslot[a]] = nr;
    if (DICT && sflag) sflag[aslot[a]] = uf[src];
    if (keep) atomicAdd(&stripe_keep[a / step], 1u);
}

// second half of a round: ranks (and jumps) that changed go to rank[] / rj[] now, the pairs of classes that still have
// to be refined move to the next active list -- stripe by stripe, at the offsets the scan of stripe_keep gave
template <bool DICT, int K> __global__ __launch_bounds__(BLOCK) void k_round_apply(const uint32_t *aslot, uint64_t na, const uint32_t *xsorted, const uint32_t *newr, const uint32_t *newj,
                                                                                   const uint8_t *flags, const uint32_t *stripe_base, uint32_t *rank, uint2 *rj,
                                                                                   uint32_t *oslot, uint32_t *ornk, uint32_t *ojmp, unsigned long long *stat /*nullable (PFP_VERBOSE): ranks scattered*/)
{
    constexpr uint32_t STEP = RoundCfg<K>::STEP;
    constexpr int PER = (STEP + BLOCK - 1) / BLOCK;             // a thread owns PER consecutive pairs of the stripe: one block scan per stripe,
    __shared__ uint32_t red[4];                                  // the loads of all its pairs in flight together
    const uint64_t s = (uint64_t)blockIdx.x * STEP;
    const uint64_t e = s + STEP < na ? s + STEP : na;
    const uint64_t a0 = s + (uint64_t)threadIdx.x * PER;
    uint8_t fl[PER]; uint32_t slot[PER], nr[PER], nj[PER], x[PER], kept = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const uint64_t a = a0 + k;
        const bool ok = a < e;
        fl[k] = ok ? flags[a] : (uint8_t)0; slot[k] = ok ? aslot[a] : 0u; nr[k] = ok ? newr[a] : 0u; nj[k] = (DICT && ok) ? newj[a] : 0u;
        x[k] = ok ? xsorted[a] : 0u;
        kept += (fl[k] & RF_KEEP) ? 1u : 0u;
    }
    uint32_t tot;
    uint32_t o = stripe_base[blockIdx.x] + block_excl_sum(kept, red, &tot);
    uint32_t nch = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        if (fl[k] & RF_CHANGED) { if (DICT) { if (rj) rj[x[k]] = make_uint2(nr[k], nj[k]); } else rank[x[k]] = nr[k]; ++nch; }
        if (fl[k] & RF_KEEP) { oslot[o] = slot[k]; ornk[o] = nr[k]; if (DICT) ojmp[o] = nj[k]; ++o; }
    }
    if (stat && nch) atomicAdd(stat, (unsigned long long)nch);
}

// ---- first state after the initial sort of all N suffixes ---------------------------------------------------------
// keys / vals: sorted initial keys and their suffixes; rk[a] = head slot of a's class (max-scanned).  Writes rank[] /
// rj[].x and keep[a] = the class of a has to be refined.  Dictionary: the initial jump sits in bits 56..60 of the key
// (k_dict_init_keys) and the covered prefix holds the terminator exactly when the jump offset is below DK_CHARS or the
// 16th character is the terminator -- no gathers.
template <bool DICT> __global__ __launch_bounds__(BLOCK) void k_init_state(const uint64_t *keys, const uint32_t *vals, const uint32_t *head, const uint32_t *rk, uint64_t N, uint32_t sigma, uint32_t *rank, uint2 *rj, uint32_t *keep)
{
    const uint64_t a = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (a >= N) return;
    const uint32_t x = vals[a];
    if (DICT) { if (rj) rj[x] = make_uint2(rk[a], x + ((uint32_t)(keys[a] >> 56) & 31u)); } else rank[x] = rk[a];
    const bool single = head[a] && (a + 1 == N || head[a + 1]);
    bool fin = single;
    if (DICT && !fin) fin = (keys[a] & ((1ULL << 56) - 1)) % sigma <= 1u;   // last base-sigma digit: padding behind a terminator, or the terminator itself
    keep[a] = fin ? 0u : 1u;
}
template <bool DICT> __global__ __launch_bounds__(BLOCK) void k_init_active(const uint64_t *keys, const uint32_t *vals, const uint32_t *rk, const uint32_t *keep, const uint32_t *pos, uint64_t N,
                                                                            uint32_t *aslot, uint32_t *arnk, uint32_t *ajmp)
{
    const uint64_t a = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (a >= N || !keep[a]) return;
    const uint32_t o = pos[a];
    aslot[o] = (uint32_t)a; arnk[o] = rk[a];
    if (DICT) ajmp[o] = vals[a] + ((uint32_t)(keys[a] >> 56) & 31u);
}

struct RoundBufs {
    uint32_t *aslot[2], *arnk[2], *ajmp[2], *newr, *xout, *tnj, *newj, *M, *stripe, *lidx, *head, *keep, *pos, *d_cnt;
    uint8_t *flags; unsigned long long *d_done; uint4 *T; uint32_t *srank; uint8_t *sflag; int textkeys;
    uint64_t *k0, *k1; uint32_t *v0, *v1;
    uint32_t run_min;      // dictionary: characters per initial key (a suffix inside a run of at least that many equal bytes is ordered by the run round)
};

// one refinement round over the active list `cur` (na pairs) -> list cur ^ 1; *na_out = its length
template <bool DICT, int K> inline int suffix_sort_round(pfp_ctx *c, RoundBufs &b, int cur, uint32_t na, uint64_t N, uint64_t h, const uint8_t *D, bool run_round, int rbits, uint32_t max_range,
                                                         uint32_t *SA, uint32_t *rank, uint2 *rj, bool verbose, uint32_t *na_out)
{
    constexpr uint32_t STEP = RoundCfg<K>::STEP;
    const int lowbits = (run_round || b.textkeys) ? 32 : rbits;      // run tokens and text keys are 32-bit words
    const unsigned gs = nblocks(na, STEP);
    const uint32_t hh = (uint32_t)(h < N ? h : N);
    const uint32_t *Mr = run_round ? (const uint32_t *)b.M : (const uint32_t *)nullptr;
    PFP_HIP(c, hipMemsetAsync(b.flags, 0, na, c->stream));
    PFP_HIP(c, hipMemsetAsync(b.stripe, 0, ((size_t)gs + 3) * 4, c->stream));
    PFP_HIP(c, hipMemsetAsync(b.d_done, 0, 8, c->stream));
    const uint4 *T = nullptr;
    if (K == 3 && b.T && (uint64_t)na * 8 > N) {     // the table pays for itself when more than ~ N/9 pairs gather from it (28 B per position against two more 128-byte lines per pair)
        PFP_LAUNCH(c, K_SS_MAKE_KEYS, N * (DICT ? 40 : 28), (k_round_table<DICT>), nblocks(N, BLOCK), (const uint32_t *)rank, (const uint2 *)rj, N, hh, D, b.T);
        T = b.T;
    }
    // algorithmic bytes per active suffix: list entry 8 (+4 jump), SA[slot] 4 in + 4 out, the gathered
    // rank 4 (K = 3: 12; dictionary: + jump 4, + 1 terminator byte), new rank 4 (+ new jump 4 + 4 through scratch), flag 1
    PFP_LAUNCH(c, K_CLASS_SORT, (uint64_t)na * ((DICT ? 46 : 25) + (K == 3 ? 8 : 0)), (k_round<DICT, K>), gs, (const uint32_t *)b.aslot[cur], (const uint32_t *)b.arnk[cur], (const uint32_t *)b.ajmp[cur], (uint64_t)na, N, SA,
               (const uint32_t *)rank, (const uint2 *)rj, T, hh, D, Mr, b.run_min, lowbits, max_range, b.newr, b.xout, b.tnj, b.newj, b.flags, b.stripe, b.d_done, b.srank, b.sflag, b.textkeys);
    unsigned long long nd = 0;
    PFP_HIP(c, hipMemcpyAsync(&nd, b.d_done, 8, hipMemcpyDeviceToHost, c->stream));
    PFP_HIP(c, hipStreamSynchronize(c->stream));
    const uint64_t nl = na - nd;
    if (nl) {   // classes too large for a tile: collect their pairs, sort them globally, put them back
        if (verbose) fprintf(stderr, "[pfbwt_hip]   class sort: %u pairs, %llu in classes too large for one tile\n", na, (unsigned long long)nl);
        const size_t mk = c->arena.mark_hi();
        uint64_t *ka, *kb = nullptr; uint32_t *ux; uint8_t *uf = nullptr;
        PFP_ALLOC_HI(c, ka, uint64_t, nl); PFP_ALLOC_HI(c, ux, uint32_t, nl);
        if (DICT && b.sflag) PFP_ALLOC_HI(c, uf, uint8_t, nl);
        if (K == 3) PFP_ALLOC_HI(c, kb, uint64_t, nl);
        const unsigned ga = nblocks(na, BLOCK), gl = nblocks(nl, BLOCK);
        PFP_LAUNCH(c, K_COMPACT, (uint64_t)na * 5, k_not_done, ga, (const uint8_t *)b.flags, (uint64_t)na, b.keep);
        PFP_TRY(device_compact(c, nullptr, b.keep, na, b.lidx, b.pos, b.d_cnt));
        PFP_LAUNCH(c, K_SS_MAKE_KEYS, nl * 48, (k_round_keys<DICT, K>), gl, (const uint32_t *)b.lidx, nl, (const uint32_t *)b.aslot[cur], (const uint32_t *)b.arnk[cur], (const uint32_t *)b.ajmp[cur], N, (const uint32_t *)SA,
                   (const uint32_t *)rank, (const uint2 *)rj, T, hh, D, Mr, b.run_min, lowbits, ka, kb, ux, b.tnj, (const uint8_t *)b.sflag, uf, b.textkeys);
        uint64_t *lsk = b.k0; uint32_t *lsv = b.v0; uint64_t *alk = b.k1; uint32_t *alv = b.v1;
        if (K == 3) {
            PFP_LAUNCH(c, K_SS_MAKE_KEYS, nl * 20, k_copy_keys_iota, gl, (const uint64_t *)kb, nl, b.k0, b.v0);
            BitRange rb = {0, 2 * lowbits};
            PFP_TRY(radix_sort_pairs<uint64_t>(c, b.k0, b.v0, b.k1, b.v1, nl, &rb, 1, &lsk, &lsv));
            PFP_LAUNCH(c, K_SS_MAKE_KEYS, nl * 20, k_gather_keys, gl, (const uint64_t *)ka, (const uint32_t *)lsv, nl, lsk);
            alk = lsk == b.k0 ? b.k1 : b.k0; alv = lsv == b.v0 ? b.v1 : b.v0;
        } else PFP_LAUNCH(c, K_SS_MAKE_KEYS, nl * 20, k_copy_keys_iota, gl, (const uint64_t *)ka, nl, b.k0, b.v0);
        BitRange rr = {0, lowbits + rbits};      // (skipping the digits that are equal in all keys was tried: the classes of a large range differ in a byte or two of the class part, one pass of eight saved on S-chr22 for a reduction kernel + a host round trip: no gain)
        uint64_t *fsk; uint32_t *fsv;
        PFP_TRY(radix_sort_pairs<uint64_t>(c, lsk, lsv, alk, alv, nl, &rr, 1, &fsk, &fsv));
        PFP_LAUNCH(c, K_SS_HEADS, nl * 12, k_round_subset_heads, gl, (const uint64_t *)fsk, (const uint64_t *)kb, (const uint32_t *)fsv, nl, b.head);
        PFP_TRY((device_scan<uint32_t, 1>(c, b.head, b.head, nl, nullptr)));
        PFP_LAUNCH(c, K_SS_WRITE_RANK, nl * 40, (k_round_finish<DICT>), gl, (const uint32_t *)fsv, (const uint32_t *)ux, (const uint32_t *)b.tnj, (const uint32_t *)b.head, (const uint32_t *)b.lidx, nl, N,
                   (const uint32_t *)b.aslot[cur], (const uint32_t *)b.arnk[cur], SA, D, STEP, b.newr, b.xout, b.newj, b.flags, b.stripe, b.srank, b.sflag, (const uint8_t *)uf);
        c->arena.release_hi(mk);
    }
    PFP_TRY((device_scan<uint32_t, 0>(c, b.stripe, b.stripe, (uint64_t)gs, b.d_cnt)));
    PFP_LAUNCH(c, K_SS_WRITE_RANK, (uint64_t)na * (DICT ? 37 : 25), (k_round_apply<DICT, K>), gs, (const uint32_t *)b.aslot[cur], (uint64_t)na, (const uint32_t *)b.xout, (const uint32_t *)b.newr, (const uint32_t *)b.newj,
               (const uint8_t *)b.flags, (const uint32_t *)b.stripe, rank, rj, b.aslot[cur ^ 1], b.arnk[cur ^ 1], b.ajmp[cur ^ 1], verbose ? b.d_done : (unsigned long long *)nullptr);
    if (verbose) {
        unsigned long long ch = 0;
        PFP_HIP(c, hipMemcpyAsync(&ch, b.d_done, 8, hipMemcpyDeviceToHost, c->stream));
        PFP_HIP(c, hipStreamSynchronize(c->stream));
        fprintf(stderr, "[pfbwt_hip]   K=%d%s: ranks scattered this round: %llu (the round kernel handled %llu pairs)\n", K, T ? " (table)" : "", ch - nd, nd);
    }
    PFP_TRY(d2h_u32(c, b.d_cnt, na_out));
    return PFP_OK;
}

// Sorts the N suffixes described by (k0, v0) [keys = first characters, vals = x; dictionary keys carry the initial jump
// offset in bits 56..60].  Outputs SA (slot -> x) and, per x, the slot of its class head: rank[x] (int alphabet,
// DICT = false: symbols are compared one by one, the covered prefix after the initial sort is h0 symbols) or rj[x].x
// (dictionary, DICT = true: suffixes end at their EndOfWord, byte-identical suffixes stay one class; D enables the run
// round).  k0/v0 and their twins k1/v1 (N entries each) are scratch owned by the caller.
template <bool DICT> inline int suffix_sort_doubling(pfp_ctx *c, uint64_t N, uint64_t *k0, uint32_t *v0, uint64_t *k1, uint32_t *v1,
                                                    const BitRange *init_ranges, int n_init_ranges, uint32_t h0, const uint8_t *D, uint32_t *SA, uint32_t *rank, uint2 *rj, int *rounds_out,
                                                    uint32_t sigma = 9 /*dictionary: codes of the initial keys' number system*/, uint32_t *srank = nullptr /*N entries, optional: per slot, head slot of its class*/,
                                                    uint8_t *sflag = nullptr /*N bytes, optional (dictionary): bit 62 of every suffix's initial key, kept at the suffix's slot*/,
                                                    int text_mode = 0 /*dictionary: every round orders by the next characters of the text, rj is not used (may be null)*/,
                                                    int *converged = nullptr /*text mode: 0 when the rounds were given up (many long common prefixes): sort again without text_mode*/)
{
    const size_t mk = c->arena.mark_hi();
    RoundBufs b{};
    b.run_min = h0;
    uint32_t *aux;
    PFP_ALLOC_HI(c, b.head, uint32_t, N); if (srank) aux = srank; else PFP_ALLOC_HI(c, aux, uint32_t, N);
    PFP_ALLOC_HI(c, b.keep, uint32_t, N); PFP_ALLOC_HI(c, b.pos, uint32_t, N);
    b.srank = srank; b.sflag = sflag; b.textkeys = (DICT && text_mode) ? 1 : 0;
    if (converged) *converged = 1;
    PFP_ALLOC_HI(c, b.d_cnt, uint32_t, 4);
    uint64_t *sk; uint32_t *sv;
    PFP_TRY(radix_sort_pairs<uint64_t>(c, k0, v0, k1, v1, N, init_ranges, n_init_ranges, &sk, &sv));
    const unsigned gN = nblocks(N, BLOCK);
    uint64_t keymask = 0; for (int r = 0; r < n_init_ranges; ++r) for (int bb = init_ranges[r].lo; bb < init_ranges[r].hi; ++bb) keymask |= 1ULL << bb;
    PFP_LAUNCH(c, K_SS_HEADS, N * 24, k_ss_heads, gN, (const uint64_t *)sk, (const uint32_t *)sv, N, keymask, SA, b.head, aux, sflag);
    PFP_TRY((device_scan<uint32_t, 1>(c, aux, aux, N, nullptr)));
    PFP_LAUNCH(c, K_SS_WRITE_RANK, N * 24, (k_init_state<DICT>), gN, (const uint64_t *)sk, (const uint32_t *)sv, (const uint32_t *)b.head, (const uint32_t *)aux, N, sigma, rank, rj, b.keep);
    PFP_TRY((device_scan<uint32_t, 0>(c, b.keep, b.pos, N, b.d_cnt)));
    uint32_t na = 0; PFP_TRY(d2h_u32(c, b.d_cnt, &na));
    int rounds = 1;
    const int rbits = bits_for(N);
    const bool verbose = c->tun.verbose != 0;
    const int force_k = c->tun.sort_k;      // tests / A-B runs: 1 = plain doubling in every round, 3 = three ranks in every round but the run round
    if (na > 0) {
        // active lists (two sets, swapped every round) and the per-round outputs; every later list is shorter than the first
        for (int t = 0; t < 2; ++t) { PFP_ALLOC_HI(c, b.aslot[t], uint32_t, na); PFP_ALLOC_HI(c, b.arnk[t], uint32_t, na); if (DICT) PFP_ALLOC_HI(c, b.ajmp[t], uint32_t, na); }
        PFP_ALLOC_HI(c, b.newr, uint32_t, na); PFP_ALLOC_HI(c, b.xout, uint32_t, na); PFP_ALLOC_HI(c, b.flags, uint8_t, na); PFP_ALLOC_HI(c, b.d_done, unsigned long long, 1);
        if (DICT) { PFP_ALLOC_HI(c, b.tnj, uint32_t, na); PFP_ALLOC_HI(c, b.newj, uint32_t, na); }
        const uint64_t max_stripes = nblocks(na, RoundCfg<3>::STEP < RoundCfg<1>::STEP ? RoundCfg<3>::STEP : RoundCfg<1>::STEP) + 1;
        PFP_ALLOC_HI(c, b.stripe, uint32_t, max_stripes + 4);
        PFP_ALLOC_HI(c, b.lidx, uint32_t, na);
        PFP_LAUNCH(c, K_COMPACT, N * 24, (k_init_active<DICT>), gN, (const uint64_t *)sk, (const uint32_t *)sv, (const uint32_t *)aux, (const uint32_t *)b.keep, (const uint32_t *)b.pos, N, b.aslot[0], b.arnk[0], b.ajmp[0]);
        if (DICT && D) {   // run lengths for the run round
            PFP_ALLOC_HI(c, b.M, uint32_t, N);
            PFP_LAUNCH(c, K_SS_MAKE_KEYS, N * 5, k_ss_runend_marks, gN, D, N, b.M);
            PFP_TRY((device_scan<uint32_t, 1>(c, b.M, b.M, N, nullptr)));
            uint32_t maxrun = 0;
            PFP_HIP(c, hipMemsetAsync(b.d_cnt + 1, 0, 4, c->stream));
            PFP_LAUNCH(c, K_SS_MAKE_KEYS, N * 4, k_max_run_length, nblocks(N, 16 * BLOCK), (const uint32_t *)b.M, N, b.d_cnt + 1);
            PFP_TRY(d2h_u32(c, b.d_cnt + 1, &maxrun));
            if (maxrun < RUN_ROUND_MIN_RUN && !c->tun.force_run_round) b.M = nullptr;      // no run round
        }
        // k0 / v0 / k1 / v1 are free from here on: scratch of the large-class route
        b.k0 = k0; b.k1 = k1; b.v0 = v0; b.v1 = v1;
        const bool no_table = c->tun.sort_no_table != 0;                   // tests: the K = 3 rounds follow the chains themselves
        if (!b.textkeys && force_k != 1 && !no_table && (uint64_t)na * 8 > N && c->arena.hi - c->arena.lo > 16 * (size_t)N + 40 * (size_t)na + ((size_t)1 << 20))
            PFP_ALLOC_HI(c, b.T, uint4, N);          // optional: without room for it the K = 3 rounds follow the chains themselves
        const uint32_t max_range_env = c->tun.class_sort_maxrange;   // tests: smaller, to reach the large-class route
        int cur = 0; uint64_t h = h0;
        uint32_t na_before = 0;                 // length of the list the previous round started from (0: no previous round)
        while (na > 0) {
            if (verbose) fprintf(stderr, "[pfbwt_hip] suffix sort N=%llu round %d: %u active\n", (unsigned long long)N, rounds, na);
            if (b.textkeys && ((rounds >= 4 && (uint64_t)na * 64 > N) || rounds > 40)) {      // a text round adds TK_CHARS characters, a doubling round doubles: give up
                if (verbose) fprintf(stderr, "[pfbwt_hip] text rounds given up (%u suffixes still share %llu characters): rank-based rounds instead\n", na, (unsigned long long)(h0 + (uint64_t)TK_CHARS * (rounds - 1)));
                if (converged) *converged = 0;
                break;
            }
            if (rounds > 64) return PFP_E_CORRUPT; // cannot happen on well-formed input
            const bool run_round = (b.M != nullptr && rounds == 1);
            uint32_t nn = 0;
            // K = 3 pays while the classes keep splitting without dissolving (14 LDS passes instead of 6, but half the rounds);
            // once a round has resolved more than half of its pairs, most of the rest is decided by the FIRST further rank and
            // the two extra ranks are sorted for nothing: plain doubling (6 passes) from there on
            const bool resolving = na_before != 0 && (uint64_t)na * 2 < na_before;
            na_before = na;
            if (run_round || force_k == 1 || b.textkeys || (resolving && force_k != 3)) {
                const uint32_t mr = max_range_env ? max_range_env : (uint32_t)RoundCfg<1>::TILE;
                PFP_TRY((suffix_sort_round<DICT, 1>(c, b, cur, na, N, h, D, run_round, rbits, mr < (uint32_t)RoundCfg<1>::TILE ? mr : (uint32_t)RoundCfg<1>::TILE, SA, rank, rj, verbose, &nn)));
                h *= 2;
            } else {
                const uint32_t mr = max_range_env ? max_range_env : (uint32_t)RoundCfg<3>::TILE;
                PFP_TRY((suffix_sort_round<DICT, 3>(c, b, cur, na, N, h, D, false, rbits, mr < (uint32_t)RoundCfg<3>::TILE ? mr : (uint32_t)RoundCfg<3>::TILE, SA, rank, rj, verbose, &nn)));
                h *= 4;
            }
            na = nn; cur ^= 1; ++rounds;
        }
    }
    if (rounds_out) *rounds_out = rounds;
    c->arena.release_hi(mk);
    return PFP_OK;
}

} // namespace pfp
