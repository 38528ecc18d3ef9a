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

// keys[x] = first 16 characters of suffix x packed 4 bits each, zero after the terminator.
__global__ __launch_bounds__(BLOCK) void k_dict_init_keys(const uint8_t *D, uint64_t dsize, uint64_t *keys, uint32_t *vals)
{
    __shared__ uint8_t tile[DK_TILE + DK_CHARS];
    const uint64_t t0 = (uint64_t)blockIdx.x * DK_TILE;
    for (uint32_t i = threadIdx.x; i < DK_TILE + DK_CHARS; i += BLOCK) {
        const uint64_t x = t0 + i;
        tile[i] = x < dsize ? (uint8_t)dict_code(D[x]) : (uint8_t)0;
    }
    __syncthreads();
    const uint32_t l0 = threadIdx.x * DK_PER_THREAD;
    // raw rolling window over [l, l+16) and distance to the first terminator at or after l
    uint64_t raw = 0;
#pragma unroll
    for (int k = 0; k < DK_CHARS - 1; ++k) raw = (raw << 4) | tile[l0 + DK_PER_THREAD + k];
    // raw now holds the 15 codes of positions l0+16 .. l0+30 (used as the window slides backwards)
    int dt = DK_CHARS; // distance from position (l0+16) to its first terminator, capped
#pragma unroll
    for (int k = DK_CHARS - 1; k >= 0; --k) if (tile[l0 + DK_PER_THREAD + k] <= 1) dt = k;
    // slide backwards: position l = l0+15 .. l0
    uint64_t win = raw; // 15 codes: positions l+1 .. l+15 (for l = l0+15), in the low 60 bits
#pragma unroll
    for (int k = DK_PER_THREAD - 1; k >= 0; --k) {
        const uint32_t l = l0 + k;
        const uint32_t c = tile[l];
        const uint64_t full = ((uint64_t)c << 60) | win;         // codes of l .. l+15
        dt = (c <= 1) ? 0 : (dt + 1 > DK_CHARS ? DK_CHARS : dt + 1);
        // keep characters 0..dt (terminator included), zero the rest
        const uint64_t key = dt >= DK_CHARS - 1 ? full : (full & ~((1ULL << (4 * (DK_CHARS - 1 - dt))) - 1ULL));
        const uint64_t x = t0 + l;
        if (x < dsize) { keys[x] = key; vals[x] = (uint32_t)x; }
        win = full >> 4;
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
__global__ __launch_bounds__(BLOCK) void k_ss_write_rank(const uint32_t *vals, const uint32_t *newrank, uint64_t na, uint32_t *rank)
{
    const uint64_t a = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (a < na) rank[vals[a]] = newrank[a];
}
// keep[a] = 1 while the class of element a still has to be refined with depth h
// ws/wordid != nullptr selects dictionary semantics: a class whose common prefix already contains the
// terminator (suff_len < h) is a group of identical suffixes and is final.
__global__ __launch_bounds__(BLOCK) void k_ss_flag_active(const uint32_t *vals, const uint32_t *head, uint64_t na, uint32_t h,
                                                          const uint32_t *ws, const uint32_t *wordid, uint32_t *keep)
{
    const uint64_t a = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (a >= na) return;
    const bool single = head[a] && (a + 1 == na || head[a + 1]);
    bool fin = single;
    if (!fin && ws) {
        const uint32_t x = vals[a];
        const uint32_t sl = ws[wordid[x] + 1] - 1u - x;   // distance to the EndOfWord of x's word
        fin = sl < h;
    }
    keep[a] = fin ? 0u : 1u;
}
__global__ __launch_bounds__(BLOCK) void k_ss_make_keys(const uint32_t *slots, const uint32_t *SA, const uint32_t *rank, uint64_t na, uint64_t N, uint32_t h,
                                                        uint64_t *keys, uint32_t *vals)
{
    const uint64_t a = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (a >= na) return;
    const uint32_t x = SA[slots[a]];
    const uint64_t y = (uint64_t)x + h;
    keys[a] = ((uint64_t)rank[x] << 32) | (y < N ? rank[y] : 0u);
    vals[a] = x;
}

// Sorts the N suffixes described by (keys,vals) [already filled: keys = h0-character prefixes, vals = x].
// Outputs SA (slot -> x) and rank (x -> slot of its class head).  ws/wordid as in k_ss_flag_active.
// keys/vals and their twins k1/v1 (N entries each) are scratch owned by the caller.
inline int suffix_sort_doubling(pfp_ctx *c, uint64_t N, uint64_t *k0, uint32_t *v0, uint64_t *k1, uint32_t *v1,
                                const BitRange *init_ranges, int n_init_ranges, uint32_t h0,
                                const uint32_t *ws, const uint32_t *wordid, uint32_t *SA, uint32_t *rank, int *rounds_out)
{
    const size_t mk = c->arena.mark_hi();
    uint32_t *head, *aux, *slots, *slots2, *d_cnt;
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
    PFP_LAUNCH(c, K_SS_WRITE_RANK, N * 12, k_ss_write_rank, gN, (const uint32_t *)sv, (const uint32_t *)aux, N, rank);
    uint32_t h = h0;
    // first active list
    uint32_t *keep = aux; // aux is free again after write_rank
    PFP_LAUNCH(c, K_SS_FLAG_ACTIVE, N * 12, k_ss_flag_active, gN, (const uint32_t *)sv, (const uint32_t *)head, N, h, ws, wordid, keep);
    PFP_TRY(device_compact(c, nullptr, keep, N, slots, slots2, d_cnt));
    uint32_t na = 0; PFP_TRY(d2h_u32(c, d_cnt, &na));
    int rounds = 1;
    const int rbits = bits_for(N);
    while (na > 0) {
        if (h >= (1u << 31) || rounds > 40) return PFP_E_CORRUPT; // cannot happen on well-formed input
        const unsigned ga = nblocks(na, BLOCK);
        uint64_t *ak = (sk == k0) ? k1 : k0; uint32_t *av = (sv == v0) ? v1 : v0; // the buffers not holding the last result
        (void)ak; (void)av;
        // build keys for the active list into k0/v0 (previous contents are dead: SA/rank hold the state)
        PFP_LAUNCH(c, K_SS_MAKE_KEYS, (uint64_t)na * 28, k_ss_make_keys, ga, (const uint32_t *)slots, (const uint32_t *)SA, (const uint32_t *)rank, (uint64_t)na, N, h, k0, v0);
        BitRange rr[2] = {{0, rbits}, {32, 32 + rbits}};
        PFP_TRY(radix_sort_pairs<uint64_t>(c, k0, v0, k1, v1, na, rr, 2, &sk, &sv));
        PFP_LAUNCH(c, K_SS_HEADS, (uint64_t)na * 28, k_ss_heads, ga, (const uint64_t *)sk, (const uint32_t *)sv, (const uint32_t *)slots, (uint64_t)na, SA, head, aux);
        PFP_TRY((device_scan<uint32_t, 1>(c, aux, aux, na, nullptr)));
        PFP_LAUNCH(c, K_SS_WRITE_RANK, (uint64_t)na * 12, k_ss_write_rank, ga, (const uint32_t *)sv, (const uint32_t *)aux, (uint64_t)na, rank);
        h *= 2; ++rounds;
        PFP_LAUNCH(c, K_SS_FLAG_ACTIVE, (uint64_t)na * 12, k_ss_flag_active, ga, (const uint32_t *)sv, (const uint32_t *)head, (uint64_t)na, h, ws, wordid, keep);
        PFP_TRY(device_compact(c, slots, keep, na, slots2, head /*pos scratch*/, d_cnt));
        uint32_t *t = slots; slots = slots2; slots2 = t;
        PFP_TRY(d2h_u32(c, d_cnt, &na));
    }
    if (rounds_out) *rounds_out = rounds;
    c->arena.release_hi(mk);
    return PFP_OK;
}

} // namespace pfp
