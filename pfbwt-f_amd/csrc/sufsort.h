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
__global__ __launch_bounds__(BLOCK) void k_ss_flag_active(const uint32_t *vals, const uint32_t *head, uint64_t na, const uint2 *rj,
                                                          const uint32_t *ws, const uint32_t *wordid, uint32_t *keep)
{
    const uint64_t a = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (a >= na) return;
    const bool single = head[a] && (a + 1 == na || head[a + 1]);
    bool fin = single;
    if (!fin && ws) {
        const uint32_t x = vals[a];
        const uint32_t term = ws[wordid[x] + 1] - 1u;     // offset of the EndOfWord of x's word
        fin = reinterpret_cast<const uint32_t *>(rj)[2 * (uint64_t)x + 1] > term;
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

// Sorts the N suffixes described by (keys,vals) [already filled: keys = h0-character prefixes, vals = x].
// Outputs SA (slot -> x) and rank (x -> slot of its class head).  ws/wordid select dictionary semantics
// (see k_ss_flag_active); D != nullptr additionally enables the run round (byte texts).
// keys/vals and their twins k1/v1 (N entries each) are scratch owned by the caller.
inline int suffix_sort_doubling(pfp_ctx *c, uint64_t N, uint64_t *k0, uint32_t *v0, uint64_t *k1, uint32_t *v1,
                                const BitRange *init_ranges, int n_init_ranges, uint32_t h0,
                                const uint32_t *ws, const uint32_t *wordid, const uint8_t *D, uint32_t *SA, uint2 *rj, int *rounds_out)
{
    const size_t mk = c->arena.mark_hi();
    uint32_t *head, *aux, *slots, *slots2, *d_cnt, *nj, *M = nullptr;
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
    PFP_LAUNCH(c, K_SS_FLAG_ACTIVE, N * 16, k_ss_flag_active, gN, (const uint32_t *)sv, (const uint32_t *)head, N, (const uint2 *)rj, ws, wordid, keep);
    PFP_TRY(device_compact(c, nullptr, keep, N, slots, slots2, d_cnt));
    uint32_t na = 0; PFP_TRY(d2h_u32(c, d_cnt, &na));
    int rounds = 1;
    const int rbits = bits_for(N);
    if (na > 0) {
        PFP_ALLOC_HI(c, nj, uint32_t, na);
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
        PFP_LAUNCH(c, K_SS_MAKE_KEYS, (uint64_t)na * 12, k_ss_apply_jump, ga, (const uint32_t *)v0, (const uint32_t *)nj, (uint64_t)na, rj);
        BitRange rr = {0, (run_round ? 32 : rbits) + rbits};
        PFP_TRY(radix_sort_pairs<uint64_t>(c, k0, v0, k1, v1, na, &rr, 1, &sk, &sv));
        PFP_LAUNCH(c, K_SS_HEADS, (uint64_t)na * 28, k_ss_heads, ga, (const uint64_t *)sk, (const uint32_t *)sv, (const uint32_t *)slots, (uint64_t)na, SA, head, aux);
        PFP_TRY((device_scan<uint32_t, 1>(c, aux, aux, na, nullptr)));
        PFP_LAUNCH(c, K_SS_WRITE_RANK, (uint64_t)na * 12, k_ss_write_rank, ga, (const uint32_t *)sv, (const uint32_t *)aux, (uint64_t)na, rj);
        ++rounds;
        PFP_LAUNCH(c, K_SS_FLAG_ACTIVE, (uint64_t)na * 16, k_ss_flag_active, ga, (const uint32_t *)sv, (const uint32_t *)head, (uint64_t)na, (const uint2 *)rj, ws, wordid, keep);
        PFP_TRY(device_compact(c, slots, keep, na, slots2, head /*pos scratch*/, d_cnt));
        uint32_t *t = slots; slots = slots2; slots2 = t;
        PFP_TRY(d2h_u32(c, d_cnt, &na));
    }
    if (rounds_out) *rounds_out = rounds;
    c->arena.release_hi(mk);
    return PFP_OK;
}

} // namespace pfp
