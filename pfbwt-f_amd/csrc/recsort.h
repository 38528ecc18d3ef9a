// pfbwt-f_amd/csrc/recsort.h -- suffix array of an integer string S (the parse) by ONE level of prefix-free parsing of S itself,
// the recursion of the reference's SACA-K restated for a data-parallel machine.
//
// Reference: sacak_int -> SACA_K, gsa/gsacak.c:1397-1526: sort the LMS substrings (:852-874, induce :215-252), NAME them (nameSubstr
// :1165-1220), recurse on the string of names (level >= 1, :928-1145), then induce the order of all suffixes from the order of the
// sampled ones (getSAlms :1347-1361, putSuffix0 :97-111).  Called at include/pfparser.hpp:425 for the parse of a collection.
// Induced sorting is sequential.  What carries over is the shape -- sample, name, recurse on the names, derive the rest -- and the
// sampling that makes it data-parallel is the parser's own: prefix-free parsing (pfparser.hpp:335-352) applied to S.
//
//   * a symbol is a TRIGGER when a hash of its value is 0 mod p2 (the last symbol, the unique 0, always is); level-2 phrase j is
//     S[ps[j] .. ps[j+1]], two consecutive phrases share their trigger symbol (w = 1).  The strings S[i .. next trigger behind i]
//     of all positions i are a prefix-free set (a string ends in a trigger and has none inside), so
//         suffix(i) < suffix(i')  <=>  (string of i, rank of the sampled suffix that starts at its last symbol) compares smaller;
//   * the distinct phrases ("words", exact hash table like the text's own de-duplication, pfparser.hpp:595-601) are written one
//     behind the other, separated, into D2; ONE suffix sort of D2 (small: the collection is repetitive) names every string above
//     (class = run of equal strings in SA(D2)) and ranks the words; P2 = the word ranks of the phrases is the string of names;
//   * SA(P2) (recursive call; prefix doubling at the bottom) ranks the sampled suffixes;
//   * assembly: the rows of a class are the occurrences of its member words (inverted lists of P2), ordered by the rank of the
//     sampled suffix behind the occurrence: whole classes are gathered into LDS, sorted there by that ONE 32-bit key and stored.
//     This is generate_bwt_lcp's grouping (include/pfbwt.hpp:137-181) one level up, with integers for characters.
// On S-32G (325 M phrases of 1000 near-identical haplotypes) four refinement rounds over all 325 M suffixes become two over the 81 M
// sampled ones plus one LDS sort pass over the rows.  Inputs that do not shrink (alphabet ~ length, dictionaries that stay large, a
// phrase without trigger for > REC_MAX_PHRASE symbols) take the doubling route of sufsort.h, which stays the bottom of the recursion.
#pragma once
#include "sufsort.h"

namespace pfp {

constexpr uint32_t REC_MAX_PHRASE = 1024;      // longest level-2 phrase this route accepts (class heads compare strings symbol by symbol)
constexpr int RS_TRIG_PER = 16;                // symbols per thread of the trigger kernels
constexpr int RS_TRIG_TILE = BLOCK * RS_TRIG_PER;

__device__ __forceinline__ uint32_t rs_trigger(uint32_t s, uint32_t p2)
{
    uint32_t h = s * 0x9E3779B1u; h ^= h >> 15; h *= 0x85EBCA77u; h ^= h >> 13;
    return (h % p2 == 0u) ? 1u : 0u;
}
// trigger bits of the 16 positions i0 .. i0+15 (position 0 never starts a new phrase, the last position always ends one)
__device__ __forceinline__ uint32_t rs_trig_mask16(const uint32_t *S, uint64_t N, uint64_t i0, uint32_t p2)
{
    uint32_t m = 0;
    if (i0 + RS_TRIG_PER <= N) {
        const uint4 *q = reinterpret_cast<const uint4 *>(S + i0);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const uint4 a = q[v];
            m |= (rs_trigger(a.x, p2) | (rs_trigger(a.y, p2) << 1) | (rs_trigger(a.z, p2) << 2) | (rs_trigger(a.w, p2) << 3)) << (4 * v);
        }
    } else {
        for (int k = 0; k < RS_TRIG_PER; ++k) if (i0 + k < N) m |= rs_trigger(S[i0 + k], p2) << k;
    }
    if (i0 == 0) m &= ~1u;
    if (N - 1 >= i0 && N - 1 < i0 + RS_TRIG_PER) m |= 1u << (uint32_t)(N - 1 - i0);
    return m;
}
__global__ __launch_bounds__(BLOCK) void k_rs_trig_count(const uint32_t *S, uint64_t N, uint32_t p2, uint32_t *cnt)
{
    __shared__ uint32_t red[4];
    const uint64_t i0 = ((uint64_t)blockIdx.x * BLOCK + threadIdx.x) * RS_TRIG_PER;
    const uint32_t m = i0 < N ? rs_trig_mask16(S, N, i0, p2) : 0u;
    uint32_t tot;
    (void)block_excl_sum((uint32_t)__popc(m), red, &tot);
    if (threadIdx.x == 0) cnt[blockIdx.x] = tot;
}
// ps[0] = 0, ps[1 + t] = position of the t-th trigger (ascending): phrase j = S[ps[j] .. ps[j+1]]
__global__ __launch_bounds__(BLOCK) void k_rs_trig_write(const uint32_t *S, uint64_t N, uint32_t p2, const uint32_t *blockoff, uint32_t *ps)
{
    __shared__ uint32_t red[4];
    const uint64_t i0 = ((uint64_t)blockIdx.x * BLOCK + threadIdx.x) * RS_TRIG_PER;
    uint32_t m = i0 < N ? rs_trig_mask16(S, N, i0, p2) : 0u;
    uint32_t tot;
    uint32_t o = 1u + blockoff[blockIdx.x] + block_excl_sum((uint32_t)__popc(m), red, &tot);
    while (m) { const int b = __ffs((int)m) - 1; ps[o++] = (uint32_t)(i0 + (uint32_t)b); m &= m - 1u; }
    if (blockIdx.x == 0 && threadIdx.x == 0) ps[0] = 0u;
}
__global__ __launch_bounds__(BLOCK) void k_rs_max_phrase(const uint32_t *ps, uint64_t k, uint32_t *out)
{
    __shared__ uint32_t red[4];
    const uint64_t j = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t len = j < k ? ps[j + 1] - ps[j] + 1u : 0u;
    uint32_t tot;
    (void)block_incl_max(len, red, &tot);
    if (threadIdx.x == 0 && tot > *(volatile uint32_t *)out) atomicMax(out, tot);
}

// ---- exact de-duplication of the level-2 phrases (the std::map of pfparser.hpp:69-70, 595-601, for strings of integers) ----------
// table entry = hash tag << 32 | (position of the representative phrase's first symbol + 1); 0 = empty.  A lookup matches when the
// tags agree AND the symbols are equal: a phrase is its first symbol, non-triggers, one trigger, so a representative that agrees with
// the len symbols of this phrase ends where it does -- the representative's own length is never looked up.  Entries are write-once:
// a stale "empty" is corrected by the value the device-scope CAS returns.  A probe sequence longer than REC_MAX_PROBES means the table
// is too full (more distinct phrases than a repetitive collection has): the caller gives this route up.
// (First version, S-32G, 81 M phrases: 37 ms -- every insertion bumped ONE global counter and every lookup read the representative's
//  bounds, a third random sector per phrase.)
constexpr uint32_t REC_MAX_PROBES = 128;
__device__ __forceinline__ uint64_t rs_phrase_hash(const uint32_t *S, uint32_t a, uint32_t b)
{
    uint64_t h = 0x243F6A8885A308D3ULL;
    for (uint32_t i = a; i <= b; ++i) { h = (h ^ S[i]) * 0x9E3779B97F4A7C15ULL; h ^= h >> 29; }
    h *= 0xBF58476D1CE4E5B9ULL; h ^= h >> 32;
    return h;
}
__global__ __launch_bounds__(BLOCK) void k_rs_dedup(const uint32_t *S, const uint32_t *ps, uint64_t k, unsigned long long *table, uint32_t tmask, uint32_t *eid, uint32_t *isrep, uint32_t *overflow)
{
    const uint64_t j = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (j >= k) return;
    const uint32_t a = ps[j], b = ps[j + 1];
    const uint64_t h = rs_phrase_hash(S, a, b);
    const uint32_t tag = (uint32_t)(h >> 32);
    const unsigned long long mine = ((unsigned long long)tag << 32) | (unsigned long long)(a + 1u);
    uint32_t slot = (uint32_t)h & tmask, rep = 0, probe = 0;
    for (; probe < REC_MAX_PROBES; ++probe) {
        unsigned long long e = table[slot];
        if (e == 0ULL) {
            e = atomicCAS(&table[slot], 0ULL, mine);
            if (e == 0ULL) { rep = 1; break; }
        }
        if ((uint32_t)(e >> 32) == tag) {
            const uint32_t ra = (uint32_t)e - 1u;
            bool eq = true;
            for (uint32_t d = 0; d <= b - a; ++d) if (S[ra + d] != S[a + d]) { eq = false; break; }
            if (eq) break;
        }
        slot = (slot + 1u) & tmask;
    }
    if (probe == REC_MAX_PROBES) atomicExch(overflow, 1u);
    eid[j] = slot; isrep[j] = rep;
}
// deterministic word ids: the distinct phrases in the order of their 64-bit content hashes; the phrase that holds the final 0 last
__global__ __launch_bounds__(BLOCK) void k_rs_rep_keys(const uint32_t *S, const uint32_t *ps, const uint32_t *replist, uint64_t nw, uint64_t k, uint64_t *keys, uint32_t *vals)
{
    const uint64_t t = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= nw) return;
    const uint32_t j = replist[t];
    keys[t] = (uint64_t)j + 1 == k ? ~0ULL : (rs_phrase_hash(S, ps[j], ps[j + 1]) >> 1);
    vals[t] = (uint32_t)t;
}
__global__ __launch_bounds__(BLOCK) void k_rs_assign_ids(const uint32_t *sorted_t, const uint32_t *replist, const uint32_t *eid, const uint32_t *ps, uint64_t nw, uint32_t *slot2id, uint32_t *wrep, uint32_t *wlen1)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i > nw) return;
    if (i == nw) { wlen1[i] = 0u; return; }
    const uint32_t j = replist[sorted_t[i]];
    slot2id[eid[j]] = (uint32_t)i; wrep[i] = j;
    wlen1[i] = ps[j + 1] - ps[j] + 2u;                      // symbols + separator
}
__global__ __launch_bounds__(BLOCK) void k_rs_wid(const uint32_t *eid, const uint32_t *slot2id, uint64_t k, uint32_t *wid, uint32_t *wid_keep, uint32_t *idx)
{
    const uint64_t j = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (j < k) { const uint32_t id = slot2id[eid[j]]; wid[j] = id; wid_keep[j] = id; idx[j] = (uint32_t)j; }      // (wid is the sort's input; wid_keep stays in phrase order for the names)
}
// D2 = the words in id order, symbols + 2, each followed by the separator 1 (the last one, which ends in S's own 0, by the final 0):
// strings compare before any separator is reached (prefix-free), a separator is below every symbol, so the suffixes with ONE string
// are neighbours in SA(D2) and the positions that stand for no string (separators, last symbols) never lie between them
__global__ __launch_bounds__(BLOCK) void k_rs_dict_build(const uint32_t *S, const uint32_t *ps, const uint32_t *wrep, const uint32_t *wstart, uint64_t nw, uint32_t *D, uint32_t *wd)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= nw) return;
    const uint32_t j = wrep[i], a = ps[j], L = ps[j + 1] - a + 1u, o = wstart[i];
    for (uint32_t d = 0; d < L; ++d) { D[o + d] = S[a + d] + 2u; wd[o + d] = (uint32_t)i; }
    D[o + L] = i + 1 == nw ? 0u : 1u; wd[o + L] = (uint32_t)i;
}
// woff[id] = first entry of word id in the list of phrases sorted by word (every word occurs)
__global__ __launch_bounds__(BLOCK) void k_rs_list_bounds(const uint32_t *skeys, uint64_t k, uint64_t nw, uint32_t *woff)
{
    const uint64_t e = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (e >= k) return;
    const uint32_t id = skeys[e];
    if (e == 0 || skeys[e - 1] != id) woff[id] = (uint32_t)e;
    if (e + 1 == k) woff[nw] = (uint32_t)k;
}
// per suffix-array slot of D2: rows it stands for (0: a separator or the last symbol of a word), whole-word flag
__global__ __launch_bounds__(BLOCK) void k_rs_slots(const uint32_t *SAD, const uint32_t *wd, const uint32_t *wstart, const uint32_t *woff, uint64_t ND, uint32_t *rows, uint32_t *whole, uint32_t *valid)
{
    const uint64_t s = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (s >= ND) return;
    const uint32_t x = SAD[s], i = wd[x], o = x - wstart[i], L = wstart[i + 1] - wstart[i] - 1u;
    rows[s] = o + 1u < L ? woff[i + 1] - woff[i] : 0u;
    valid[s] = o + 1u < L ? 1u : 0u;
    whole[s] = o == 0u ? 1u : 0u;
}
__global__ __launch_bounds__(BLOCK) void k_rs_word_ranks(const uint32_t *SAD, const uint32_t *wd, const uint32_t *whole, const uint32_t *wpos, uint64_t ND, uint32_t *wrank)
{
    const uint64_t s = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (s < ND && whole[s]) wrank[wd[SAD[s]]] = wpos[s] + 1u;
}
__global__ __launch_bounds__(BLOCK) void k_rs_names(const uint32_t *wid, const uint32_t *wrank, uint64_t k, uint32_t *P2)
{
    const uint64_t j = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (j < k) P2[j] = wrank[wid[j]];
    else if (j == k) P2[j] = 0u;
}
// per list entry (a phrase, grouped by word): rank of the sampled suffix behind it, text position of its first symbol
__global__ __launch_bounds__(BLOCK) void k_rs_list_payload(const uint32_t *inv, const uint32_t *R2, const uint32_t *ps, uint64_t k, uint32_t *ikey, uint32_t *ipos)
{
    const uint64_t e = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (e >= k) return;
    const uint32_t j = inv[e];
    ikey[e] = R2[j + 1]; ipos[e] = ps[j];
}
// valid slots (v = index in the compacted list): head of a class iff the string differs from the previous valid slot's
__global__ __launch_bounds__(BLOCK) void k_rs_heads(const uint32_t *D, const uint32_t *SAD, const uint32_t *vlist, uint64_t nv, uint32_t *head)
{
    const uint64_t v = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (v >= nv) return;
    // (everything that depends on s is loaded before the loop: hipcc 7.2 -O3 lost s across it for the lanes that skip the loop --
    //  the copy it keeps around the loop body is only written inside -- and a row offset was read at a wild address behind it: the one
    //  GPU fault of this file's first run on the card, invisible to the CPU interpreter)
    const uint32_t s = vlist[v];
    const uint32_t sp = v ? vlist[v - 1] : 0u;
    uint32_t hd = (v == 0 || sp + 1u != s) ? 1u : 0u;
    uint32_t x = SAD[s], y = SAD[s ? s - 1 : 0];
    if (!hd) {
        for (uint32_t d = 0; d <= REC_MAX_PHRASE + 1u; ++d) {
            const uint32_t a = D[x + d], b = D[y + d];
            if (a != b) { hd = (a | b) > 1u ? 1u : 0u; break; }
            if (a <= 1u) break;
        }
    }
    head[v] = hd;
}
// srec[v] = { first row, first list entry of its word, offset of the string in the word, index of its class }
__global__ __launch_bounds__(BLOCK) void k_rs_slot_records(const uint32_t *SAD, const uint32_t *vlist, const uint32_t *rowoff, const uint32_t *head, const uint32_t *hpos /*heads in front of v*/, const uint32_t *wd, const uint32_t *wstart,
                                                            const uint32_t *woff, uint64_t nv, uint4 *srec)
{
    const uint64_t v = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (v >= nv) return;
    const uint32_t s = vlist[v], x = SAD[s], i = wd[x];
    srec[v] = make_uint4(rowoff[s], woff[i], x - wstart[i], hpos[v] + head[v] - 1u);
}
__global__ __launch_bounds__(BLOCK) void k_rs_class_rows(const uint32_t *chead, const uint4 *srec, uint64_t nc, uint64_t nv, uint64_t R, uint32_t *crow, uint32_t *chead_end)
{
    const uint64_t ci = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (ci < nc) crow[ci] = srec[chead[ci]].x;
    else if (ci == nc) { crow[ci] = (uint32_t)R; *chead_end = (uint32_t)nv; }
}

// first index in [lo, hi) with a[idx] >= key
__device__ __forceinline__ uint32_t rs_lower_bound(const uint32_t *a, uint32_t lo, uint32_t hi, uint64_t key)
{
    while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if ((uint64_t)a[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}

// lanes of the wave that hold the same NB-bit digit as this one (prims.h: same_digit_lanes is the 8-bit form)
template <int NB> __device__ __forceinline__ void rs_same_digit_lanes(uint32_t d, uint32_t &plo, uint32_t &phi)
{
#pragma unroll
    for (int bb = 0; bb < NB; ++bb) {
        const uint32_t neg = (uint32_t)(((int32_t)(d << (31 - bb))) >> 31);
        const unsigned long long m = __builtin_amdgcn_ballot_w64(neg != 0u);
        plo &= ~((uint32_t)m ^ neg); phi &= ~((uint32_t)(m >> 32) ^ neg);
    }
}

// Assembly.  Workgroup b owns the classes whose first row lies in [b*STEP, (b+1)*STEP); it takes them in batches of whole classes
// of at most tile_rows rows: the rows (list entries of the member words) are gathered into LDS, sorted there by
// (index of the class in the batch, rank of the sampled suffix behind the occurrence) -- stable LSD passes over 9-bit digits with
// wave-ballot ranking, the pass of the class sort of sufsort.h with a wider digit: S-32G has 27 key bits and ~125 classes per batch,
// 35 bits = 4 passes (8-bit digits and the class's first row as high part: 6 passes, 13.1 ms) -- and stored: SA[1 + row] = text
// position (row 0 is the final 0).  A class with more rows than a tile is appended to `bigc` (global sort route).
constexpr int RA_DB = 9, RA_RADIX = 1 << RA_DB;
template <bool RANK, bool KEYOUT> __global__ __launch_bounds__(BLOCK) void k_rs_assemble(const uint4 *srec, const uint32_t *chead /*nc + 1*/, const uint32_t *crow /*nc + 1*/, uint32_t nc,
                                                                            const uint32_t *ikey, const uint32_t *ipos, int keybits, uint32_t tile_rows, uint32_t *SA, uint32_t *rank,
                                                                            uint32_t *bigc, uint32_t *nbig, uint32_t *okey /*KEYOUT (dictrec.h): the key of every row, in row order*/,
                                                                            uint8_t *oflag /*KEYOUT: bit 31 of a list position marks the start of a dictionary word; oflag[row] = the row is that position itself (sflag)*/)
{
    constexpr int ITEMS = RS_ITEMS, TILE = RS_TILE;
    constexpr uint32_t STEP = TILE / 2;
    __shared__ uint32_t wh[BLOCK / WAVE][RA_RADIX];
    __shared__ uint64_t skeys[TILE];
    __shared__ uint16_t sidx[TILE];
    __shared__ uint16_t srow[TILE + 1];
    __shared__ uint32_t red[4];
    __shared__ uint32_t ctl[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t w0 = (uint64_t)blockIdx.x * STEP;
    if (threadIdx.x == 0) { ctl[0] = rs_lower_bound(crow, 0u, nc, w0); ctl[1] = rs_lower_bound(crow, 0u, nc, w0 + STEP); }
    __syncthreads();
    uint32_t ci = ctl[0];
    const uint32_t ci1 = ctl[1];
    while (ci < ci1) {
        __syncthreads();                                       // ctl / LDS of the previous batch are free
        if (threadIdx.x == 0) {
            // largest cj in (ci, ci1] with crow[cj] - crow[ci] <= tile_rows (crow ascends strictly: every class has rows)
            const uint64_t lim = (uint64_t)crow[ci] + tile_rows;
            uint32_t lo = ci, hi = ci1;                        // invariant: crow[lo] <= lim
            while (lo < hi) { const uint32_t mid = lo + ((hi - lo + 1) >> 1); if ((uint64_t)crow[mid] <= lim) lo = mid; else hi = mid - 1; }
            ctl[2] = lo;
            if (lo == ci) bigc[atomicAdd(nbig, 1u)] = ci;
        }
        __syncthreads();
        const uint32_t cj = ctl[2];
        if (cj == ci) { ++ci; continue; }
        const uint32_t r0 = crow[ci], n = crow[cj] - r0, v0 = chead[ci], ns = chead[cj] - v0;
        for (uint32_t t = threadIdx.x; t < ns; t += BLOCK) srow[t] = (uint16_t)(srec[v0 + t].x - r0);
        if (threadIdx.x == 0) srow[ns] = (uint16_t)n;
        __syncthreads();
        const uint32_t nit = (n + BLOCK - 1) / BLOCK;
        uint32_t xi[ITEMS];
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const uint32_t i = threadIdx.x + (uint32_t)it * BLOCK;
            xi[it] = 0u;
            if (i < n) {
                uint32_t lo = 0, hi = ns - 1;                  // largest t with srow[t] <= i
                while (lo < hi) { const uint32_t mid = lo + ((hi - lo + 1) >> 1); if (srow[mid] <= i) lo = mid; else hi = mid - 1; }
                const uint4 rec = srec[v0 + lo];
                const uint32_t e = rec.y + (i - (rec.x - r0));
                xi[it] = KEYOUT ? (rec.z ? (ipos[e] & 0x7FFFFFFFu) + rec.z : ipos[e]) : ipos[e] + rec.z;
                skeys[i] = ((uint64_t)(rec.w - ci) << keybits) | ikey[e];
                sidx[i] = (uint16_t)i;
            }
        }
        for (uint32_t i = n + threadIdx.x; i < nit * BLOCK; i += BLOCK) { skeys[i] = ~0ULL; sidx[i] = (uint16_t)i; }
        __syncthreads();
        int tbits = keybits;                                   // key bits + bits of the largest class index of the batch
        for (uint32_t hs = cj - 1 - ci; hs; hs >>= 1) ++tbits;
        const int npass = (tbits + RA_DB - 1) / RA_DB;
        const uint32_t base = (uint32_t)wave * (nit * WAVE) + lane;
        for (int p = 0; p < npass; ++p) {
            const int sh = RA_DB * p;
            uint64_t kk[ITEMS]; uint16_t vv[ITEMS]; unsigned dg[ITEMS];
#pragma unroll
            for (int w = 0; w < BLOCK / WAVE; ++w) { wh[w][threadIdx.x] = 0; wh[w][threadIdx.x + BLOCK] = 0; }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                if ((uint32_t)it < nit) {
                    const uint32_t i = base + (uint32_t)it * WAVE;
                    kk[it] = skeys[i]; vv[it] = sidx[i];
                    const uint32_t d = (uint32_t)(kk[it] >> sh) & (RA_RADIX - 1);
                    uint32_t plo = 0xFFFFFFFFu, phi = 0xFFFFFFFFu;
                    rs_same_digit_lanes<RA_DB>(d, plo, phi);
                    const int leader = plo ? __builtin_ctz(plo) : 32 + __builtin_ctz(phi);
                    uint32_t old = 0;
                    if (lane == leader) { old = wh[wave][d]; wh[wave][d] = old + (uint32_t)__builtin_popcount(plo) + (uint32_t)__builtin_popcount(phi); }
                    old = __shfl(old, leader);
                    dg[it] = d | ((old + __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u))) << RA_DB);
                }
            }
            __syncthreads();
            {   // thread t owns the digits 2t and 2t + 1
                const unsigned d0 = 2u * threadIdx.x;
                uint32_t c0[BLOCK / WAVE], c1[BLOCK / WAVE]; uint32_t total = 0;
#pragma unroll
                for (int w = 0; w < BLOCK / WAVE; ++w) { const uint2 q = *reinterpret_cast<const uint2 *>(&wh[w][d0]); c0[w] = q.x; c1[w] = q.y; total += q.x + q.y; }
                uint32_t tt;
                uint32_t run = block_excl_sum(total, red, &tt);
                uint32_t o0[BLOCK / WAVE];
#pragma unroll
                for (int w = 0; w < BLOCK / WAVE; ++w) { o0[w] = run; run += c0[w]; }
#pragma unroll
                for (int w = 0; w < BLOCK / WAVE; ++w) { *reinterpret_cast<uint2 *>(&wh[w][d0]) = make_uint2(o0[w], run); run += c1[w]; }
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                if ((uint32_t)it < nit) {
                    const uint32_t li = wh[wave][dg[it] & (RA_RADIX - 1)] + (dg[it] >> RA_DB);
                    skeys[li] = kk[it]; sidx[li] = vv[it];
                }
            }
            __syncthreads();
        }
        if (KEYOUT) {
            const uint64_t km = keybits >= 32 ? 0xFFFFFFFFULL : ((1ULL << keybits) - 1ULL);
            for (uint32_t i = threadIdx.x; i < n; i += BLOCK) okey[1u + r0 + i] = (uint32_t)(skeys[i] & km);
            __syncthreads();
        }
        // the keys are not needed any more: their LDS holds the text positions, indexed by row before the sort
        uint32_t *sx = reinterpret_cast<uint32_t *>(skeys);
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) { const uint32_t i = threadIdx.x + (uint32_t)it * BLOCK; if (i < n) sx[i] = xi[it]; }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += BLOCK) {
            uint32_t x = sx[sidx[i]];
            if (KEYOUT) { oflag[1u + r0 + i] = (uint8_t)(x >> 31); x &= 0x7FFFFFFFu; }
            SA[1u + r0 + i] = x;
            if (RANK) rank[x] = 1u + r0 + i;
        }
        ci = cj;
    }
}

// ---- classes with more rows than a tile: their rows through the global radix sort ------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_rs_big_sizes(const uint32_t *bigc, uint64_t nb, const uint32_t *crow, uint32_t *sizes)
{
    const uint64_t b = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (b < nb) sizes[b] = crow[bigc[b] + 1] - crow[bigc[b]];
    else if (b == nb) sizes[b] = 0u;
}
__global__ __launch_bounds__(BLOCK) void k_rs_big_rows(const uint32_t *bigc, const uint32_t *bigoff /*nb + 1*/, uint32_t nb, uint64_t nbr, const uint32_t *crow, const uint32_t *chead, const uint4 *srec,
                                                       const uint32_t *ikey, const uint32_t *ipos, uint64_t *keys, uint32_t *vals, int flagged /*dictrec.h: bit 31 of a list position is a flag (see k_rs_assemble)*/)
{
    const uint64_t q = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (q >= nbr) return;
    uint32_t lo = 0, hi = nb - 1;                              // largest b with bigoff[b] <= q
    while (lo < hi) { const uint32_t mid = lo + ((hi - lo + 1) >> 1); if ((uint64_t)bigoff[mid] <= q) lo = mid; else hi = mid - 1; }
    const uint32_t b = lo, c = bigc[b], row = crow[c] + (uint32_t)(q - bigoff[b]);
    uint32_t vl = chead[c], vh = chead[c + 1] - 1;             // largest v with srec[v].x <= row
    while (vl < vh) { const uint32_t mid = vl + ((vh - vl + 1) >> 1); if (srec[mid].x <= row) vl = mid; else vh = mid - 1; }
    const uint4 rec = srec[vl];
    const uint32_t e = rec.y + (row - rec.x);
    keys[q] = ((uint64_t)b << 32) | ikey[e];
    vals[q] = flagged ? (rec.z ? (ipos[e] & 0x7FFFFFFFu) + rec.z : ipos[e]) : ipos[e] + rec.z;
}
template <bool RANK, bool KEYOUT> __global__ __launch_bounds__(BLOCK) void k_rs_big_store(const uint64_t *keys, const uint32_t *vals, uint64_t nbr, const uint32_t *bigc, const uint32_t *bigoff, const uint32_t *crow, uint32_t *SA, uint32_t *rank, uint32_t *okey, uint8_t *oflag)
{
    const uint64_t q = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (q >= nbr) return;
    const uint32_t b = (uint32_t)(keys[q] >> 32);
    const uint32_t row = crow[bigc[b]] + (uint32_t)(q - bigoff[b]);
    uint32_t x = vals[q];
    if (KEYOUT) { oflag[1u + row] = (uint8_t)(x >> 31); x &= 0x7FFFFFFFu; }
    SA[1u + row] = x;
    if (RANK) rank[x] = 1u + row;
    if (KEYOUT) okey[1u + row] = (uint32_t)keys[q];
}
__global__ __launch_bounds__(BLOCK) void k_rs_first_row(uint32_t *SA, uint32_t *rank, uint64_t N)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) { SA[0] = (uint32_t)(N - 1); if (rank) rank[N - 1] = 0u; }
}

inline int sort_int_suffixes(pfp_ctx *c, const uint32_t *dS, uint64_t N, uint64_t maxsym, uint32_t *SA, uint32_t *rank, int *rounds, int depth = 0, bool want_rank = true);

// prefix doubling (sufsort.h): the bottom of the recursion and the route of inputs that do not shrink
inline int sort_int_suffixes_doubling(pfp_ctx *c, const uint32_t *dS, uint64_t N, uint64_t maxsym, uint32_t *SA, uint32_t *rank, int *rounds)
{
    const size_t mk = c->arena.mark_hi();
    uint64_t *k0, *k1; uint32_t *v0, *v1;
    PFP_ALLOC_HI(c, k0, uint64_t, N); PFP_ALLOC_HI(c, k1, uint64_t, N);
    PFP_ALLOC_HI(c, v0, uint32_t, N); PFP_ALLOC_HI(c, v1, uint32_t, N);
    const int sb = bits_for(maxsym);
    const int nsym = (c->tun.int_key_symbols == 3 && 3 * sb <= 64) ? 3 : 2;
    PFP_LAUNCH(c, K_SS_INIT_KEYS, N * 16, k_int_init_keys, nblocks(N, BLOCK), dS, N, sb, nsym, k0, v0);
    BitRange rr = {0, nsym * sb};
    PFP_TRY(suffix_sort_doubling<false>(c, N, k0, v0, k1, v1, &rr, 1, (uint32_t)nsym, (const uint8_t *)nullptr, SA, rank, (uint2 *)nullptr, rounds));
    c->arena.release_hi(mk);
    return PFP_OK;
}

// *taken = 0: nothing was written, the caller sorts by doubling
inline int suffix_sort_pfp(pfp_ctx *c, const uint32_t *dS, uint64_t N, uint64_t maxsym, uint32_t *SA, uint32_t *rank /*nullable*/, int depth, int *taken)
{
    *taken = 0;
    const bool forced = c->tun.parse_rec > 0, verbose = c->tun.verbose != 0;
    const uint32_t p2 = c->tun.parse_rec_p2 >= 2 ? (uint32_t)c->tun.parse_rec_p2 : 4u;
    if (N < 8 || N + 64 >= 0xFFFFFFFFULL || maxsym + 2 >= 0xFFFFFFFFULL) return PFP_OK;
    const size_t mk = c->arena.mark_hi();
    HostTimer tm;
    // ---- level-2 phrases
    const unsigned gt = nblocks(N, RS_TRIG_TILE);
    uint32_t *bcnt, *d_cnt;
    PFP_ALLOC_HI(c, bcnt, uint32_t, gt); PFP_ALLOC_HI(c, d_cnt, uint32_t, 8);
    PFP_HIP(c, hipMemsetAsync(d_cnt, 0, 32, c->stream));
    PFP_LAUNCH(c, K_REC_PARSE, N * 4, k_rs_trig_count, gt, dS, N, p2, bcnt);
    PFP_TRY((device_scan<uint32_t, 0>(c, bcnt, bcnt, (uint64_t)gt, d_cnt)));
    uint32_t k32 = 0; PFP_TRY(d2h_u32(c, d_cnt, &k32));
    const uint64_t k = k32;                                    // phrases; the last one ends in the final 0
    if (k < 2 || (!forced && k * 2 > N)) { c->arena.release_hi(mk); return PFP_OK; }
    uint32_t *ps; PFP_ALLOC_HI(c, ps, uint32_t, k + 1);
    PFP_LAUNCH(c, K_REC_PARSE, N * 4 + k * 4, k_rs_trig_write, gt, dS, N, p2, (const uint32_t *)bcnt, ps);
    PFP_LAUNCH(c, K_REC_PARSE, k * 4, k_rs_max_phrase, nblocks(k, BLOCK), (const uint32_t *)ps, k, d_cnt + 1);
    uint32_t maxlen = 0; PFP_TRY(d2h_u32(c, d_cnt + 1, &maxlen));
    if (maxlen > REC_MAX_PHRASE) {
        if (verbose) fprintf(stderr, "[pfbwt_hip] recursive parse sort given up: a level-2 phrase of %u symbols\n", maxlen);
        c->arena.release_hi(mk); return PFP_OK;
    }
    // ---- distinct phrases
    int tl = c->tun.parse_rec_table_log2 > 0 ? c->tun.parse_rec_table_log2 : bits_for(k / 8 + 1023);      // S-32G: 16 M slots for 2.6 M distinct among 81 M phrases
    if (tl > 31) tl = 31;
    const uint64_t tsize = 1ULL << tl;
    unsigned long long *table; uint32_t *eid, *isrep, *pos;
    PFP_ALLOC_HI(c, table, unsigned long long, tsize); PFP_ALLOC_HI(c, eid, uint32_t, k); PFP_ALLOC_HI(c, isrep, uint32_t, k); PFP_ALLOC_HI(c, pos, uint32_t, k);
    PFP_HIP(c, hipMemsetAsync(table, 0, tsize * 8, c->stream));
    // (the per-XCD column order of the text de-duplication, parse.h, was tried here too: the parse of a collection is laid out like its text.  No gain: 72.4 against 72.1 ms of parse BWT, r04po)
    PFP_LAUNCH(c, K_REC_DEDUP, N * 8 + k * 24, k_rs_dedup, nblocks(k, BLOCK), dS, (const uint32_t *)ps, k, table, (uint32_t)(tsize - 1), eid, isrep, d_cnt + 3);
    uint32_t ovf = 0; PFP_TRY(d2h_u32(c, d_cnt + 3, &ovf));
    if (ovf) {
        if (verbose) fprintf(stderr, "[pfbwt_hip] recursive parse sort given up: the table of %llu slots is too full for the distinct level-2 phrases among %llu\n", (unsigned long long)tsize, (unsigned long long)k);
        c->arena.release_hi(mk); return PFP_OK;
    }
    uint32_t *replist; PFP_ALLOC_HI(c, replist, uint32_t, k < tsize ? k : tsize);
    PFP_TRY(device_compact(c, nullptr, isrep, k, replist, pos, d_cnt + 4));
    uint32_t nw32 = 0; PFP_TRY(d2h_u32(c, d_cnt + 4, &nw32));
    const uint64_t nw = nw32;
    uint32_t *slot2id, *wrep, *wstart, *wid = isrep /*reused*/, *inv0, *wid1, *inv1, *woff;
    PFP_ALLOC_HI(c, slot2id, uint32_t, tsize); PFP_ALLOC_HI(c, wrep, uint32_t, nw); PFP_ALLOC_HI(c, wstart, uint32_t, nw + 1);
    PFP_ALLOC_HI(c, inv0, uint32_t, k); PFP_ALLOC_HI(c, wid1, uint32_t, k); PFP_ALLOC_HI(c, inv1, uint32_t, k); PFP_ALLOC_HI(c, woff, uint32_t, nw + 1);
    {
        const size_t mk2 = c->arena.mark_hi();
        uint64_t *hk0, *hk1; uint32_t *hv0, *hv1;
        PFP_ALLOC_HI(c, hk0, uint64_t, nw); PFP_ALLOC_HI(c, hk1, uint64_t, nw); PFP_ALLOC_HI(c, hv0, uint32_t, nw); PFP_ALLOC_HI(c, hv1, uint32_t, nw);
        PFP_LAUNCH(c, K_REC_DEDUP, nw * 40, k_rs_rep_keys, nblocks(nw, BLOCK), dS, (const uint32_t *)ps, (const uint32_t *)replist, nw, k, hk0, hv0);
        BitRange hr = {0, 64};
        uint64_t *sk; uint32_t *sv;
        PFP_TRY(radix_sort_pairs<uint64_t>(c, hk0, hv0, hk1, hv1, nw, &hr, 1, &sk, &sv));
        PFP_LAUNCH(c, K_REC_DEDUP, nw * 24, k_rs_assign_ids, nblocks(nw + 1, BLOCK), (const uint32_t *)sv, (const uint32_t *)replist, (const uint32_t *)eid, (const uint32_t *)ps, nw, slot2id, wrep, wstart);
        c->arena.release_hi(mk2);
    }
    PFP_TRY((device_scan<uint32_t, 0>(c, wstart, wstart, nw + 1, d_cnt + 5)));      // wstart[nw] = ND
    uint32_t nd32 = 0; PFP_TRY(d2h_u32(c, d_cnt + 5, &nd32));
    const uint64_t ND = nd32;
    if (verbose) fprintf(stderr, "[pfbwt_hip] recursive parse sort (depth %d): N=%llu -> %llu phrases (longest %u), %llu distinct, D2=%llu symbols (%.1f ms so far)\n", depth, (unsigned long long)N,
                         (unsigned long long)k, maxlen, (unsigned long long)nw, (unsigned long long)ND, tm.ms());
    if (!forced && (ND + k) * 10 > N * 6) { c->arena.release_hi(mk); return PFP_OK; }      // does not shrink enough to pay for the assembly
    PFP_LAUNCH(c, K_REC_DEDUP, k * 16, k_rs_wid, nblocks(k, BLOCK), (const uint32_t *)eid, (const uint32_t *)slot2id, k, wid, eid /*in place: the entry index is not needed any more*/, inv0);
    // inverted lists: phrases grouped by word (stable: ascending inside a word)
    uint32_t *swid, *inv;
    {
        BitRange wr = {0, bits_for(nw ? nw - 1 : 0)};
        PFP_TRY(radix_sort_pairs<uint32_t>(c, wid, inv0, wid1, inv1, k, &wr, 1, &swid, &inv));
        PFP_LAUNCH(c, K_REC_PARSE, k * 4, k_rs_list_bounds, nblocks(k, BLOCK), (const uint32_t *)swid, k, nw, woff);
    }
    const uint32_t *widp = eid;                                // word ids in phrase order (the sort consumed its own copy)
    // ---- D2 and its suffix array
    uint32_t *D, *wd, *SAD;
    PFP_ALLOC_HI(c, D, uint32_t, ND + 4); PFP_ALLOC_HI(c, wd, uint32_t, ND); PFP_ALLOC_HI(c, SAD, uint32_t, ND);
    PFP_LAUNCH(c, K_REC_PARSE, ND * 12, k_rs_dict_build, nblocks(nw, BLOCK), dS, (const uint32_t *)ps, (const uint32_t *)wrep, (const uint32_t *)wstart, nw, D, wd);
    {
        const size_t mk2 = c->arena.mark_hi();
        uint32_t *rkD; PFP_ALLOC_HI(c, rkD, uint32_t, ND);
        int r2 = 0;
        PFP_TRY(sort_int_suffixes(c, D, ND, maxsym + 2, SAD, rkD, &r2, depth + 1, false));
        c->arena.release_hi(mk2);
    }
    if (verbose) fprintf(stderr, "[pfbwt_hip]   D2 sorted (%.1f ms so far)\n", tm.ms());
    // ---- slots: rows, word ranks, names
    uint32_t *rows, *whole, *rowoff, *wposs, *wrank, *P2, *SA2, *R2, *valid;
    PFP_ALLOC_HI(c, valid, uint32_t, ND); PFP_ALLOC_HI(c, rows, uint32_t, ND); PFP_ALLOC_HI(c, whole, uint32_t, ND); PFP_ALLOC_HI(c, rowoff, uint32_t, ND); PFP_ALLOC_HI(c, wposs, uint32_t, ND);
    PFP_ALLOC_HI(c, wrank, uint32_t, nw); PFP_ALLOC_HI(c, P2, uint32_t, k + 1); PFP_ALLOC_HI(c, SA2, uint32_t, k + 1); PFP_ALLOC_HI(c, R2, uint32_t, k + 1);
    PFP_LAUNCH(c, K_REC_PARSE, ND * 28, k_rs_slots, nblocks(ND, BLOCK), (const uint32_t *)SAD, (const uint32_t *)wd, (const uint32_t *)wstart, (const uint32_t *)woff, ND, rows, whole, valid);
    PFP_TRY((device_scan<uint32_t, 0>(c, whole, wposs, ND, nullptr)));
    PFP_LAUNCH(c, K_REC_PARSE, ND * 16, k_rs_word_ranks, nblocks(ND, BLOCK), (const uint32_t *)SAD, (const uint32_t *)wd, (const uint32_t *)whole, (const uint32_t *)wposs, ND, wrank);
    PFP_LAUNCH(c, K_REC_PARSE, k * 12, k_rs_names, nblocks(k + 1, BLOCK), widp, (const uint32_t *)wrank, k, P2);
    {
        int r2 = 0;
        PFP_TRY(sort_int_suffixes(c, P2, k + 1, nw, SA2, R2, &r2, depth + 1, true));
    }
    if (verbose) fprintf(stderr, "[pfbwt_hip]   P2 sorted (%.1f ms so far)\n", tm.ms());
    // ---- assembly
    uint32_t *ikey = SA2 /*not needed any more*/, *ipos = P2;
    PFP_LAUNCH(c, K_REC_PARSE, k * 20, k_rs_list_payload, nblocks(k, BLOCK), (const uint32_t *)inv, (const uint32_t *)R2, (const uint32_t *)ps, k, ikey, ipos);
    PFP_TRY((device_scan<uint32_t, 0>(c, rows, rowoff, ND, d_cnt + 6)));
    uint32_t R32 = 0; PFP_TRY(d2h_u32(c, d_cnt + 6, &R32));
    if ((uint64_t)R32 != N - 1) { c->arena.release_hi(mk); return PFP_E_CORRUPT; }
    uint32_t *vlist, *head, *chead, *crow;
    PFP_ALLOC_HI(c, vlist, uint32_t, ND); PFP_ALLOC_HI(c, head, uint32_t, ND); PFP_ALLOC_HI(c, chead, uint32_t, ND + 1); PFP_ALLOC_HI(c, crow, uint32_t, ND + 1);
    PFP_TRY(device_compact(c, nullptr, valid, ND, vlist, wposs, d_cnt + 7));               // the slots that stand for a string
    uint32_t nv32 = 0; PFP_TRY(d2h_u32(c, d_cnt + 7, &nv32));
    const uint64_t nv = nv32;
    uint4 *srec; PFP_ALLOC_HI(c, srec, uint4, nv);
    PFP_LAUNCH(c, K_REC_PARSE, nv * 40, k_rs_heads, nblocks(nv, BLOCK), (const uint32_t *)D, (const uint32_t *)SAD, (const uint32_t *)vlist, nv, head);
    PFP_HIP(c, hipMemsetAsync(d_cnt, 0, 32, c->stream));
    PFP_TRY(device_compact(c, nullptr, head, nv, chead, wposs, d_cnt));                    // wposs[v] = heads in front of v
    PFP_LAUNCH(c, K_REC_PARSE, nv * 44, k_rs_slot_records, nblocks(nv, BLOCK), (const uint32_t *)SAD, (const uint32_t *)vlist, (const uint32_t *)rowoff, (const uint32_t *)head, (const uint32_t *)wposs, (const uint32_t *)wd,
               (const uint32_t *)wstart, (const uint32_t *)woff, nv, srec);
    uint32_t nc32 = 0; PFP_TRY(d2h_u32(c, d_cnt, &nc32));
    const uint64_t nc = nc32;
    PFP_LAUNCH(c, K_REC_PARSE, nc * 24, k_rs_class_rows, nblocks(nc + 1, BLOCK), (const uint32_t *)chead, (const uint4 *)srec, nc, nv, N - 1, crow, chead + nc);
    uint32_t tile_rows = c->tun.parse_rec_tile_rows ? c->tun.parse_rec_tile_rows : (uint32_t)RS_TILE;
    if (tile_rows > (uint32_t)RS_TILE) tile_rows = RS_TILE;
    if (tile_rows < 2) tile_rows = 2;
    const int keybits = bits_for(k);
    uint32_t *bigc; PFP_ALLOC_HI(c, bigc, uint32_t, nc + 1);
    const unsigned ga = nblocks(N - 1, RS_TILE / 2);
    // algorithmic bytes per row: list entry 8 in, text position 4 out (+ 4 rank); per slot 16
    if (rank) PFP_LAUNCH(c, K_REC_ASSEMBLE, (N - 1) * 16 + nv * 16, (k_rs_assemble<true, false>), ga, (const uint4 *)srec, (const uint32_t *)chead, (const uint32_t *)crow, (uint32_t)nc, (const uint32_t *)ikey, (const uint32_t *)ipos,
                         keybits, tile_rows, SA, rank, bigc, d_cnt + 1, (uint32_t *)nullptr, (uint8_t *)nullptr);
    else PFP_LAUNCH(c, K_REC_ASSEMBLE, (N - 1) * 12 + nv * 16, (k_rs_assemble<false, false>), ga, (const uint4 *)srec, (const uint32_t *)chead, (const uint32_t *)crow, (uint32_t)nc, (const uint32_t *)ikey, (const uint32_t *)ipos,
                    keybits, tile_rows, SA, rank, bigc, d_cnt + 1, (uint32_t *)nullptr, (uint8_t *)nullptr);
    PFP_LAUNCH(c, K_MISC, 8, k_rs_first_row, 1, SA, rank, N);
    uint32_t nb32 = 0; PFP_TRY(d2h_u32(c, d_cnt + 1, &nb32));
    if (nb32) {      // classes with more rows than a tile
        const uint64_t nb = nb32;
        uint32_t *bigoff; PFP_ALLOC_HI(c, bigoff, uint32_t, nb + 1);
        PFP_LAUNCH(c, K_REC_PARSE, nb * 12, k_rs_big_sizes, nblocks(nb + 1, BLOCK), (const uint32_t *)bigc, nb, (const uint32_t *)crow, bigoff);
        PFP_TRY((device_scan<uint32_t, 0>(c, bigoff, bigoff, nb + 1, d_cnt + 2)));
        uint32_t nbr32 = 0; PFP_TRY(d2h_u32(c, d_cnt + 2, &nbr32));
        const uint64_t nbr = nbr32;
        if (verbose) fprintf(stderr, "[pfbwt_hip]   assembly: %llu classes with more than %u rows (%llu rows) through the global sort\n", (unsigned long long)nb, tile_rows, (unsigned long long)nbr);
        uint64_t *bk0, *bk1; uint32_t *bv0, *bv1;
        PFP_ALLOC_HI(c, bk0, uint64_t, nbr); PFP_ALLOC_HI(c, bk1, uint64_t, nbr); PFP_ALLOC_HI(c, bv0, uint32_t, nbr); PFP_ALLOC_HI(c, bv1, uint32_t, nbr);
        PFP_LAUNCH(c, K_REC_PARSE, nbr * 40, k_rs_big_rows, nblocks(nbr, BLOCK), (const uint32_t *)bigc, (const uint32_t *)bigoff, (uint32_t)nb, nbr, (const uint32_t *)crow, (const uint32_t *)chead, (const uint4 *)srec,
                   (const uint32_t *)ikey, (const uint32_t *)ipos, bk0, bv0, 0);
        BitRange br[2] = {{0, keybits}, {32, 32 + bits_for(nb - 1)}};
        uint64_t *sk; uint32_t *sv;
        PFP_TRY(radix_sort_pairs<uint64_t>(c, bk0, bv0, bk1, bv1, nbr, br, 2, &sk, &sv));
        if (rank) PFP_LAUNCH(c, K_REC_PARSE, nbr * 24, (k_rs_big_store<true, false>), nblocks(nbr, BLOCK), (const uint64_t *)sk, (const uint32_t *)sv, nbr, (const uint32_t *)bigc, (const uint32_t *)bigoff, (const uint32_t *)crow, SA, rank, (uint32_t *)nullptr, (uint8_t *)nullptr);
        else PFP_LAUNCH(c, K_REC_PARSE, nbr * 20, (k_rs_big_store<false, false>), nblocks(nbr, BLOCK), (const uint64_t *)sk, (const uint32_t *)sv, nbr, (const uint32_t *)bigc, (const uint32_t *)bigoff, (const uint32_t *)crow, SA, rank, (uint32_t *)nullptr, (uint8_t *)nullptr);
    }
    PFP_HIP(c, hipStreamSynchronize(c->stream));
    if (verbose) fprintf(stderr, "[pfbwt_hip]   assembled: %llu slots, %llu classes (%.1f ms)\n", (unsigned long long)nv, (unsigned long long)nc, tm.ms());
    c->arena.release_hi(mk);
    *taken = 1;
    return PFP_OK;
}

// suffix array of S[0..N) (S[N-1] == 0 unique smallest), integer alphabet with values <= maxsym; rank[x] = slot of suffix x
// (written when want_rank; always by the doubling route, which works in it)
inline int sort_int_suffixes(pfp_ctx *c, const uint32_t *dS, uint64_t N, uint64_t maxsym, uint32_t *SA, uint32_t *rank, int *rounds, int depth, bool want_rank)
{
    const int mode = c->tun.parse_rec;                         // -1: by the input's shape, 0: never, 1: wherever the route can run at all
    const int max_depth = c->tun.parse_rec_depth > 0 ? c->tun.parse_rec_depth : 1;
    if (mode != 0 && depth < max_depth) {
        // a repetitive collection: few distinct symbols for its length (S-32G: 1.5 M words, 325 M phrases); a single genome's parse
        // (nearly every phrase its own word) has nothing to gain
        const bool shape = N >= c->tun.parse_rec_min && maxsym * 8 <= N;
        if (mode > 0 || shape) {
            int taken = 0;
            PFP_TRY(suffix_sort_pfp(c, dS, N, maxsym, SA, want_rank ? rank : (uint32_t *)nullptr, depth, &taken));
            if (taken) { if (rounds) *rounds = 1; return PFP_OK; }
        }
    }
    return sort_int_suffixes_doubling(c, dS, N, maxsym, SA, rank, rounds);
}

} // namespace pfp
