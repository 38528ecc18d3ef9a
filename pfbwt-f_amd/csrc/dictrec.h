// pfbwt-f_amd/csrc/dictrec.h -- the suffix sort of the DICTIONARY through one level of prefix-free parsing of the dictionary itself: what
// recsort.h does for the parse, for a string of bytes whose words end in EndOfWord and whose byte-identical suffixes must stay ONE class.
//
// Reference: sort_dict_suffixes, include/pfbwt.hpp:206-223 -> gsacak (+ LCP), gsa/gsacak.c:2504-2524 -> gSACA_K_LCP :1649-1929 -- again an
// induced sort with a recursion on the names of sampled substrings (:1769), and only "glcp[j] >= suff_len" is consumed (pfbwt.hpp:137).
// The dictionary of a pangenome is itself repetitive: 870 K words for the 325 K loci of S-32G, the variants of a locus identical up to a SNP
// (its level-2 dictionary is 0.3 of it, measured before this file was written).  So:
//   * level-2 phrases inside every word: a phrase ends where a hash of the last DR_W bytes is 0 mod p2 (windows never cross a word end) and
//     at the word's EndOfWord; consecutive phrases overlap by DR_W bytes (prefix-free parsing proper, pfparser.hpp:335-352, per word);
//   * every dictionary offset x is owned by the phrase in which its suffix is longer than DR_W (the last phrase of a word owns all that
//     is left, the EndOfWord position included); suffix x sorts by (its string up to the phrase end, the sampled suffix at the next phrase);
//   * D2 = the distinct phrases as a small dictionary of their own (EndOfWord-separated, the last phrases of words without their
//     EndOfWord -- the separator stands for it): sorted by the EXISTING dictionary sorter (sufsort.h, k_round<true>), whose classes of
//     identical suffixes are exactly the classes of phrase-suffix strings;
//   * P2 = per word the ranks of its phrases, words separated by 1: sorted by the integer sorter (recsort.h / sufsort.h); two sampled
//     suffixes that agree up to their separator are the SAME suffix of the dictionary: tie classes by comparing neighbours;
//   * assembly (k_rs_assemble of recsort.h, with the keys kept): the rows of a D2 class are the occurrences of its member phrases, sorted by
//     the tie class of the sampled suffix behind them; equal (class, key) = byte-identical dictionary suffixes = one class (srank).
// Outputs: gsa (slot -> offset), srank (slot -> first slot of its class), sflag (slot -> the suffix starts a word) -- what the text-round
// sort of sufsort.h leaves for the emission and the word ranks.  The order INSIDE a class is not the offset order of gsacak (nothing in
// the engine asks for it: emit.h finds a group's first member by word rank); the gsacak drop-in keeps the old sorter.
#pragma once
#include "recsort.h"

namespace pfp {

constexpr int DR_W = 4;                        // bytes of a level-2 trigger window
constexpr uint32_t DR_MAX_PHRASE = 1024;       // longest level-2 phrase this route accepts (a run of N without a trigger is one phrase: old route)

__device__ __forceinline__ uint32_t dr_trigger(const uint8_t *D, uint64_t x, uint32_t p2)
{
    const uint8_t c = D[x];
    if (c == EndOfWord) return 1u;
    if (c == EndOfDict || x < (uint64_t)(DR_W - 1)) return 0u;
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < DR_W; ++k) { const uint32_t b = D[x - (DR_W - 1) + k]; if (b <= 1u) return 0u; v = (v << 8) | b; }
    uint32_t h = v * 0x9E3779B1u; h ^= h >> 15; h *= 0x85EBCA77u; h ^= h >> 13;
    return (h % p2 == 0u) ? 1u : 0u;
}
__global__ __launch_bounds__(BLOCK) void k_dr_trig_count(const uint8_t *D, uint64_t N, uint32_t p2, uint32_t *cnt)
{
    __shared__ uint32_t red[4];
    const uint64_t x0 = ((uint64_t)blockIdx.x * BLOCK + threadIdx.x) * 16;
    uint32_t n = 0;
    for (int k = 0; k < 16; ++k) if (x0 + k < N) n += dr_trigger(D, x0 + k, p2);
    uint32_t tot;
    (void)block_excl_sum(n, red, &tot);
    if (threadIdx.x == 0) cnt[blockIdx.x] = tot;
}
// pe[j] = position of the j-th phrase end (ascending)
__global__ __launch_bounds__(BLOCK) void k_dr_trig_write(const uint8_t *D, uint64_t N, uint32_t p2, const uint32_t *blockoff, uint32_t *pe)
{
    __shared__ uint32_t red[4];
    const uint64_t x0 = ((uint64_t)blockIdx.x * BLOCK + threadIdx.x) * 16;
    uint32_t m = 0;
    for (int k = 0; k < 16; ++k) if (x0 + k < N) m |= dr_trigger(D, x0 + k, p2) << k;
    uint32_t tot;
    uint32_t o = blockoff[blockIdx.x] + block_excl_sum((uint32_t)__popc(m), red, &tot);
    while (m) { const int b = __ffs((int)m) - 1; pe[o++] = (uint32_t)(x0 + (uint32_t)b); m &= m - 1u; }
}
// phrase j = D[ps[j] .. pe[j]]: it starts at its word's first byte, or DR_W - 1 bytes in front of the previous phrase's end
__global__ __launch_bounds__(BLOCK) void k_dr_starts(const uint8_t *D, const uint32_t *pe, uint64_t k, uint32_t *ps, uint32_t *maxlen)
{
    __shared__ uint32_t red[4];
    const uint64_t j = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    uint32_t len = 0;
    if (j < k) {
        uint32_t s = 0;
        if (j) { const uint32_t pv = pe[j - 1]; s = D[pv] == EndOfWord ? pv + 1u : pv - (uint32_t)(DR_W - 1); }
        ps[j] = s; len = pe[j] - s + 1u;
    }
    uint32_t tot;
    (void)block_incl_max(len, red, &tot);
    if (threadIdx.x == 0 && tot > *(volatile uint32_t *)maxlen) atomicMax(maxlen, tot);
}
__device__ __forceinline__ uint64_t dr_phrase_hash(const uint8_t *D, uint32_t a, uint32_t b)
{
    uint64_t h = 0x243F6A8885A308D3ULL;
    for (uint32_t i = a; i <= b; ++i) { h = (h ^ D[i]) * 0x9E3779B97F4A7C15ULL; h ^= h >> 29; }
    h *= 0xBF58476D1CE4E5B9ULL; h ^= h >> 32;
    return h;
}
// exact de-duplication of the phrases (bytes, the EndOfWord of a word's last phrase included): entry = tag << 32 | (phrase index + 1)
__global__ __launch_bounds__(BLOCK) void k_dr_dedup(const uint8_t *D, const uint32_t *ps, const uint32_t *pe, uint64_t k, unsigned long long *table, uint32_t tmask, uint32_t *eid, uint32_t *isrep, uint32_t *overflow)
{
    const uint64_t j = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (j >= k) return;
    const uint32_t a = ps[j], b = pe[j];
    const uint64_t h = dr_phrase_hash(D, a, b);
    const uint32_t tag = (uint32_t)(h >> 32);
    const unsigned long long mine = ((unsigned long long)tag << 32) | (unsigned long long)(uint32_t)(j + 1);
    uint32_t slot = (uint32_t)h & tmask, rep = 0, probe = 0;
    for (; probe < REC_MAX_PROBES; ++probe) {
        unsigned long long e = table[slot];
        if (e == 0ULL) {
            e = atomicCAS(&table[slot], 0ULL, mine);
            if (e == 0ULL) { rep = 1; break; }
        }
        if ((uint32_t)(e >> 32) == tag) {
            const uint32_t r = (uint32_t)e - 1u, ra = ps[r];
            if (pe[r] - ra == b - a) {
                bool eq = true;
                for (uint32_t d = 0; d <= b - a; ++d) if (D[ra + d] != D[a + d]) { eq = false; break; }
                if (eq) break;
            }
        }
        slot = (slot + 1u) & tmask;
    }
    if (probe == REC_MAX_PROBES) atomicExch(overflow, 1u);
    eid[j] = slot; isrep[j] = rep;
}
__global__ __launch_bounds__(BLOCK) void k_dr_rep_keys(const uint8_t *D, const uint32_t *ps, const uint32_t *pe, const uint32_t *replist, uint64_t nw2, uint64_t *keys, uint32_t *vals)
{
    const uint64_t t = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= nw2) return;
    const uint32_t j = replist[t];
    keys[t] = dr_phrase_hash(D, ps[j], pe[j]); vals[t] = (uint32_t)t;
}
// per distinct phrase i (in the order of the sorted hashes): its representative, whether it closes a word, its D2 length + separator
__global__ __launch_bounds__(BLOCK) void k_dr_assign_ids(const uint8_t *D, const uint32_t *sorted_t, const uint32_t *replist, const uint32_t *eid, const uint32_t *ps, const uint32_t *pe, uint64_t nw2,
                                                         uint32_t *slot2id, uint32_t *wrep, uint32_t *wlen1, uint8_t *lastw)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i > nw2) return;
    if (i == nw2) { wlen1[i] = 0u; return; }
    const uint32_t j = replist[sorted_t[i]];
    const uint32_t last = D[pe[j]] == EndOfWord ? 1u : 0u;
    slot2id[eid[j]] = (uint32_t)i; wrep[i] = j; lastw[i] = (uint8_t)last;
    wlen1[i] = pe[j] - ps[j] + 1u - last + 1u;             // bytes without the word's EndOfWord, + the separator
}
__global__ __launch_bounds__(BLOCK) void k_dr_dict_build(const uint8_t *D, const uint32_t *ps, const uint32_t *wrep, const uint32_t *wstart, uint64_t nw2, uint64_t ND, uint8_t *D2, uint32_t *wd2)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i > nw2) return;
    if (i == nw2) { D2[ND - 1] = EndOfDict; wd2[ND - 1] = (uint32_t)(nw2 ? nw2 - 1 : 0); return; }
    const uint32_t j = wrep[i], a = ps[j], o = wstart[i], L = wstart[i + 1] - o - 1u;
    for (uint32_t d = 0; d < L; ++d) { D2[o + d] = D[a + d]; wd2[o + d] = (uint32_t)i; }
    D2[o + L] = EndOfWord; wd2[o + L] = (uint32_t)i;
}
// P2: per dictionary word the ranks (+ 2) of its phrases, then the separator 1; the final 0 behind the last word.  Phrase j of word wj
// sits at P2[j + wj]
__global__ __launch_bounds__(BLOCK) void k_dr_names(const uint8_t *D, const uint32_t *pe, const uint32_t *wordid, const uint32_t *wid2, const uint32_t *wrank2, uint64_t k, uint64_t nwords, uint32_t *P2)
{
    const uint64_t j = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (j < k) {
        const uint32_t e = pe[j], wj = wordid[e];
        P2[j + wj] = wrank2[wid2[j]] + 2u;
        if (D[e] == EndOfWord) P2[j + wj + 1] = 1u;
    } else if (j == k) P2[k + nwords] = 0u;
}
// neighbours in SA(P2) that agree up to their separator are the same suffix of the dictionary
__global__ __launch_bounds__(BLOCK) void k_dr_p2_heads(const uint32_t *P2, const uint32_t *SA2, uint64_t n2, uint32_t *headslot)
{
    const uint64_t s = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (s >= n2) return;
    uint32_t hd = 1u;
    if (s > 0) {
        const uint32_t x = SA2[s], y = SA2[s - 1];
        for (uint32_t d = 0; d <= DR_MAX_PHRASE * 64u; ++d) {
            const uint32_t a = x + d < n2 ? P2[x + d] : 0u, b = y + d < n2 ? P2[y + d] : 0u;
            if (a != b) { hd = (a | b) > 1u ? 1u : 0u; break; }
            if (a <= 1u) { hd = 0u; break; }
        }
    }
    headslot[s] = hd ? (uint32_t)s : 0u;
}
__global__ __launch_bounds__(BLOCK) void k_dr_p2_class(const uint32_t *SA2, const uint32_t *headslot /*max-scanned*/, uint64_t n2, uint32_t *rc)
{
    const uint64_t s = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (s < n2) rc[SA2[s]] = headslot[s];
}
// list entry (a phrase occurrence, grouped by distinct phrase): key = tie class of the sampled suffix at the next phrase of its word (0 behind
// a word's last phrase: nothing follows), position = first byte of the occurrence (| bit 31 when that byte starts a word)
__global__ __launch_bounds__(BLOCK) void k_dr_list_payload(const uint8_t *D, const uint32_t *inv, const uint32_t *ps, const uint32_t *pe, const uint32_t *wordid, const uint32_t *rc, uint64_t k, uint32_t *ikey, uint32_t *ipos)
{
    const uint64_t e = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (e >= k) return;
    const uint32_t j = inv[e], en = pe[j];
    ikey[e] = D[en] == EndOfWord ? 0u : rc[j + 1u + wordid[en]];
    const uint32_t x = ps[j];      // bit 31: the occurrence starts a dictionary word (row `x` itself then gets sflag = 1 from the assembly; positions stay below 2^31)
    ipos[e] = x | ((D[x] > EndOfWord && (x == 0 || D[x - 1] == EndOfWord)) ? 0x80000000u : 0u);
}
// per slot of SA(D2): rows it stands for, whole-phrase flag.  Offset o of phrase W (L bytes): a closing phrase owns every offset up to its
// separator (the word's EndOfWord), any other phrase the offsets whose suffix is longer than the window
__global__ __launch_bounds__(BLOCK) void k_dr_slots(const uint32_t *SA2d, const uint32_t *wd2, const uint32_t *wstart, const uint8_t *lastw, const uint32_t *woff, uint64_t ND, uint32_t *rows, uint32_t *whole, uint32_t *valid)
{
    const uint64_t s = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (s >= ND) return;
    const uint32_t y = SA2d[s];
    uint32_t v = 0u, r = 0u, wh = 0u;
    if (y + 1 < ND) {                                         // (the final EndOfDict stands for nothing)
        const uint32_t W = wd2[y], o = y - wstart[W], L = wstart[W + 1] - wstart[W] - 1u;
        v = lastw[W] ? 1u : (o + (uint32_t)DR_W + 1u <= L ? 1u : 0u);
        r = v ? woff[W + 1] - woff[W] : 0u;
        wh = (o == 0u && L > 0u) ? 1u : 0u;
    }
    rows[s] = r; whole[s] = wh; valid[s] = v;
}
__global__ __launch_bounds__(BLOCK) void k_dr_word_ranks(const uint32_t *SA2d, const uint32_t *wd2, const uint32_t *whole, const uint32_t *wpos, uint64_t ND, uint32_t *wrank2)
{
    const uint64_t s = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (s < ND && whole[s]) wrank2[wd2[SA2d[s]]] = wpos[s];
}
// valid slots: head of a D2 class iff the class head slot differs from the previous valid slot's
__global__ __launch_bounds__(BLOCK) void k_dr_heads(const uint32_t *vlist, const uint32_t *srank2, uint64_t nv, uint32_t *head)
{
    const uint64_t v = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (v >= nv) return;
    head[v] = (v == 0 || srank2[vlist[v]] != srank2[vlist[v - 1]]) ? 1u : 0u;
}
__global__ __launch_bounds__(BLOCK) void k_dr_slot_records(const uint32_t *SA2d, const uint32_t *vlist, const uint32_t *rowoff, const uint32_t *head, const uint32_t *hpos, const uint32_t *wd2, const uint32_t *wstart,
                                                           const uint32_t *woff, uint64_t nv, uint4 *srec)
{
    const uint64_t v = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (v >= nv) return;
    const uint32_t s = vlist[v], y = SA2d[s], W = wd2[y];
    srec[v] = make_uint4(rowoff[s], woff[W], y - wstart[W], hpos[v] + head[v] - 1u);
}
// rows of the assembled array: a class of identical dictionary suffixes starts where a D2 class starts or the key changes
__global__ __launch_bounds__(BLOCK) void k_dr_mark_class_rows(const uint32_t *crow, uint64_t nc, uint32_t *cstart)
{
    const uint64_t ci = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (ci < nc) cstart[1u + crow[ci]] = 1u;
}
__global__ __launch_bounds__(BLOCK) void k_dr_final_heads(const uint32_t *okey, const uint32_t *cstart, uint64_t N, uint32_t *headslot)
{
    const uint64_t r = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r >= N) return;
    const bool hd = r <= 1 || cstart[r] || okey[r] != okey[r - 1];
    headslot[r] = hd ? (uint32_t)r : 0u;
}
// *taken = 0: nothing was written (the dictionary does not shrink, a phrase is too long, ...): the caller sorts with sufsort.h
inline int dict_sort_pfp(pfp_ctx *c, uint32_t *gsa, uint32_t *srank, uint8_t *sflag, int *taken)
{
    *taken = 0;
    const uint64_t N = c->dsize, nwords = c->dwords;
    const uint8_t *D = c->d_dict;
    const bool forced = c->tun.dict_rec > 0, verbose = c->tun.verbose != 0;
    const uint32_t p2 = c->tun.dict_rec_p2 >= 2 ? (uint32_t)c->tun.dict_rec_p2 : 16u;
    if (N < 64 || N + 64 >= 0x7FFFFFFFULL || !c->d_wordid) return PFP_OK;
    const size_t mk = c->arena.mark_hi();
    HostTimer tm;
    const unsigned gt = nblocks(N, 16 * BLOCK);
    uint32_t *bcnt, *d_cnt;
    PFP_ALLOC_HI(c, bcnt, uint32_t, gt); PFP_ALLOC_HI(c, d_cnt, uint32_t, 16);
    PFP_HIP(c, hipMemsetAsync(d_cnt, 0, 64, c->stream));
    PFP_LAUNCH(c, K_REC_PARSE, N, k_dr_trig_count, gt, D, N, p2, bcnt);
    PFP_TRY((device_scan<uint32_t, 0>(c, bcnt, bcnt, (uint64_t)gt, d_cnt)));
    uint32_t k32 = 0; PFP_TRY(d2h_u32(c, d_cnt, &k32));
    const uint64_t k = k32;
    if (k < nwords || (!forced && k * 6 > N)) { c->arena.release_hi(mk); return PFP_OK; }
    uint32_t *pe, *ps;
    PFP_ALLOC_HI(c, pe, uint32_t, k + 1); PFP_ALLOC_HI(c, ps, uint32_t, k + 1);
    PFP_LAUNCH(c, K_REC_PARSE, N + k * 4, k_dr_trig_write, gt, D, N, p2, (const uint32_t *)bcnt, pe);
    PFP_LAUNCH(c, K_REC_PARSE, k * 9, k_dr_starts, nblocks(k, BLOCK), D, (const uint32_t *)pe, k, ps, d_cnt + 1);
    uint32_t maxlen = 0; PFP_TRY(d2h_u32(c, d_cnt + 1, &maxlen));
    if (maxlen > DR_MAX_PHRASE) {
        if (verbose) fprintf(stderr, "[pfbwt_hip] recursive dictionary sort given up: a level-2 phrase of %u bytes\n", maxlen);
        c->arena.release_hi(mk); return PFP_OK;
    }
    // ---- distinct phrases
    int tl = c->tun.parse_rec_table_log2 > 0 ? c->tun.parse_rec_table_log2 : bits_for(k / 2 + 1023);
    if (tl > 31) tl = 31;
    const uint64_t tsize = 1ULL << tl;
    unsigned long long *table; uint32_t *eid, *isrep, *pos;
    PFP_ALLOC_HI(c, table, unsigned long long, tsize); PFP_ALLOC_HI(c, eid, uint32_t, k); PFP_ALLOC_HI(c, isrep, uint32_t, k); PFP_ALLOC_HI(c, pos, uint32_t, k);
    PFP_HIP(c, hipMemsetAsync(table, 0, tsize * 8, c->stream));
    PFP_LAUNCH(c, K_REC_DEDUP, N * 2 + k * 24, k_dr_dedup, nblocks(k, BLOCK), D, (const uint32_t *)ps, (const uint32_t *)pe, k, table, (uint32_t)(tsize - 1), eid, isrep, d_cnt + 3);
    uint32_t ovf = 0; PFP_TRY(d2h_u32(c, d_cnt + 3, &ovf));
    if (ovf) { if (verbose) fprintf(stderr, "[pfbwt_hip] recursive dictionary sort given up: phrase table too full\n"); c->arena.release_hi(mk); return PFP_OK; }
    uint32_t *replist; PFP_ALLOC_HI(c, replist, uint32_t, k < tsize ? k : tsize);
    PFP_TRY(device_compact(c, nullptr, isrep, k, replist, pos, d_cnt + 4));
    uint32_t nw32 = 0; PFP_TRY(d2h_u32(c, d_cnt + 4, &nw32));
    const uint64_t nw2 = nw32;
    uint32_t *slot2id, *wrep, *wstart, *wid2 = isrep, *inv0, *wid1, *inv1, *woff; uint8_t *lastw;
    PFP_ALLOC_HI(c, slot2id, uint32_t, tsize); PFP_ALLOC_HI(c, wrep, uint32_t, nw2); PFP_ALLOC_HI(c, wstart, uint32_t, nw2 + 1); PFP_ALLOC_HI(c, lastw, uint8_t, nw2 + 1);
    PFP_ALLOC_HI(c, inv0, uint32_t, k); PFP_ALLOC_HI(c, wid1, uint32_t, k); PFP_ALLOC_HI(c, inv1, uint32_t, k); PFP_ALLOC_HI(c, woff, uint32_t, nw2 + 1);
    {
        const size_t mk2 = c->arena.mark_hi();
        uint64_t *hk0, *hk1; uint32_t *hv0, *hv1;
        PFP_ALLOC_HI(c, hk0, uint64_t, nw2); PFP_ALLOC_HI(c, hk1, uint64_t, nw2); PFP_ALLOC_HI(c, hv0, uint32_t, nw2); PFP_ALLOC_HI(c, hv1, uint32_t, nw2);
        PFP_LAUNCH(c, K_REC_DEDUP, nw2 * 40, k_dr_rep_keys, nblocks(nw2, BLOCK), D, (const uint32_t *)ps, (const uint32_t *)pe, (const uint32_t *)replist, nw2, hk0, hv0);
        BitRange hr = {0, 64};
        uint64_t *sk; uint32_t *sv;
        PFP_TRY(radix_sort_pairs<uint64_t>(c, hk0, hv0, hk1, hv1, nw2, &hr, 1, &sk, &sv));
        PFP_LAUNCH(c, K_REC_DEDUP, nw2 * 24, k_dr_assign_ids, nblocks(nw2 + 1, BLOCK), D, (const uint32_t *)sv, (const uint32_t *)replist, (const uint32_t *)eid, (const uint32_t *)ps, (const uint32_t *)pe, nw2, slot2id, wrep, wstart, lastw);
        c->arena.release_hi(mk2);
    }
    PFP_TRY((device_scan<uint32_t, 0>(c, wstart, wstart, nw2 + 1, d_cnt + 5)));
    uint32_t nd32 = 0; PFP_TRY(d2h_u32(c, d_cnt + 5, &nd32));
    const uint64_t ND = (uint64_t)nd32 + 1;                    // + the final EndOfDict
    const uint64_t n2 = k + nwords + 1;                        // length of P2
    if (verbose) fprintf(stderr, "[pfbwt_hip] recursive dictionary sort: %llu bytes, %llu words -> %llu level-2 phrases (longest %u), %llu distinct, D2 = %llu bytes (%.1f ms so far)\n",
                         (unsigned long long)N, (unsigned long long)nwords, (unsigned long long)k, maxlen, (unsigned long long)nw2, (unsigned long long)ND, tm.ms());
    if (!forced && (ND * 2 > N)) { c->arena.release_hi(mk); return PFP_OK; }      // the dictionary is not repetitive enough to pay for the assembly
    PFP_LAUNCH(c, K_REC_DEDUP, k * 16, k_rs_wid, nblocks(k, BLOCK), (const uint32_t *)eid, (const uint32_t *)slot2id, k, wid2, eid, inv0);
    uint32_t *swid, *inv;
    {
        BitRange wr = {0, bits_for(nw2 ? nw2 - 1 : 0)};
        PFP_TRY(radix_sort_pairs<uint32_t>(c, wid2, inv0, wid1, inv1, k, &wr, 1, &swid, &inv));
        PFP_LAUNCH(c, K_REC_PARSE, k * 4, k_rs_list_bounds, nblocks(k, BLOCK), (const uint32_t *)swid, k, nw2, woff);
    }
    const uint32_t *widp = eid;
    // ---- D2 and its suffix array with classes (the dictionary sorter of sufsort.h)
    uint8_t *D2; uint32_t *wd2, *gsa2, *srank2;
    PFP_ALLOC_HI(c, D2, uint8_t, ND + 64); PFP_ALLOC_HI(c, wd2, uint32_t, ND); PFP_ALLOC_HI(c, gsa2, uint32_t, ND); PFP_ALLOC_HI(c, srank2, uint32_t, ND);
    PFP_HIP(c, hipMemsetAsync(D2 + ND, 0, 64, c->stream));
    PFP_LAUNCH(c, K_REC_PARSE, ND * 6, k_dr_dict_build, nblocks(nw2 + 1, BLOCK), D, (const uint32_t *)ps, (const uint32_t *)wrep, (const uint32_t *)wstart, nw2, ND, D2, wd2);
    {
        const size_t mk2 = c->arena.mark_hi();
        uint64_t *k0, *k1; uint32_t *v0, *v1; uint2 *rj2;
        PFP_ALLOC_HI(c, k0, uint64_t, ND); PFP_ALLOC_HI(c, k1, uint64_t, ND); PFP_ALLOC_HI(c, v0, uint32_t, ND); PFP_ALLOC_HI(c, v1, uint32_t, ND); PFP_ALLOC_HI(c, rj2, uint2, ND);
        PFP_LAUNCH(c, K_SS_INIT_KEYS, ND * 13, k_dict_init_keys, nblocks(ND, DK_TILE), (const uint8_t *)D2, ND, k0, v0);
        BitRange full = {0, DK_KEY_BITS};
        int rounds = 0;
        PFP_TRY(suffix_sort_doubling<true>(c, ND, k0, v0, k1, v1, &full, 1, DK_CHARS, D2, gsa2, (uint32_t *)nullptr, rj2, &rounds, 9, srank2));
        c->arena.release_hi(mk2);
    }
    if (verbose) fprintf(stderr, "[pfbwt_hip]   D2 sorted (%.1f ms so far)\n", tm.ms());
    // ---- slots, phrase ranks, names
    uint32_t *rows, *whole, *rowoff, *wposs, *wrank2, *P2, *SA2, *R2, *valid, *rc;
    PFP_ALLOC_HI(c, valid, uint32_t, ND); PFP_ALLOC_HI(c, rows, uint32_t, ND); PFP_ALLOC_HI(c, whole, uint32_t, ND); PFP_ALLOC_HI(c, rowoff, uint32_t, ND); PFP_ALLOC_HI(c, wposs, uint32_t, ND);
    PFP_ALLOC_HI(c, wrank2, uint32_t, nw2); PFP_ALLOC_HI(c, P2, uint32_t, n2); PFP_ALLOC_HI(c, SA2, uint32_t, n2); PFP_ALLOC_HI(c, R2, uint32_t, n2); PFP_ALLOC_HI(c, rc, uint32_t, n2);
    PFP_LAUNCH(c, K_REC_PARSE, ND * 28, k_dr_slots, nblocks(ND, BLOCK), (const uint32_t *)gsa2, (const uint32_t *)wd2, (const uint32_t *)wstart, (const uint8_t *)lastw, (const uint32_t *)woff, ND, rows, whole, valid);
    PFP_TRY((device_scan<uint32_t, 0>(c, whole, wposs, ND, nullptr)));
    PFP_LAUNCH(c, K_REC_PARSE, ND * 16, k_dr_word_ranks, nblocks(ND, BLOCK), (const uint32_t *)gsa2, (const uint32_t *)wd2, (const uint32_t *)whole, (const uint32_t *)wposs, ND, wrank2);
    PFP_LAUNCH(c, K_REC_PARSE, k * 16, k_dr_names, nblocks(k + 1, BLOCK), D, (const uint32_t *)pe, (const uint32_t *)c->d_wordid, widp, (const uint32_t *)wrank2, k, nwords, P2);
    {
        int r2 = 0;
        PFP_TRY(sort_int_suffixes(c, P2, n2, nw2 + 1, SA2, R2, &r2, 1, true));      // (depth 1: prefix doubling)
    }
    PFP_LAUNCH(c, K_REC_PARSE, n2 * 16, k_dr_p2_heads, nblocks(n2, BLOCK), (const uint32_t *)P2, (const uint32_t *)SA2, n2, R2 /*reused: head slots*/);
    PFP_TRY((device_scan<uint32_t, 1>(c, R2, R2, n2, nullptr)));
    PFP_LAUNCH(c, K_REC_PARSE, n2 * 12, k_dr_p2_class, nblocks(n2, BLOCK), (const uint32_t *)SA2, (const uint32_t *)R2, n2, rc);
    if (verbose) fprintf(stderr, "[pfbwt_hip]   P2 sorted (%.1f ms so far)\n", tm.ms());
    // ---- assembly
    uint32_t *ikey = SA2, *ipos = P2;
    PFP_LAUNCH(c, K_REC_PARSE, k * 24, k_dr_list_payload, nblocks(k, BLOCK), D, (const uint32_t *)inv, (const uint32_t *)ps, (const uint32_t *)pe, (const uint32_t *)c->d_wordid, (const uint32_t *)rc, k, ikey, ipos);
    PFP_TRY((device_scan<uint32_t, 0>(c, rows, rowoff, ND, d_cnt + 6)));
    uint32_t R32 = 0; PFP_TRY(d2h_u32(c, d_cnt + 6, &R32));
    if ((uint64_t)R32 != N - 1) { c->arena.release_hi(mk); return PFP_E_CORRUPT; }
    uint32_t *vlist, *head, *chead, *crow;
    PFP_ALLOC_HI(c, vlist, uint32_t, ND); PFP_ALLOC_HI(c, head, uint32_t, ND); PFP_ALLOC_HI(c, chead, uint32_t, ND + 1); PFP_ALLOC_HI(c, crow, uint32_t, ND + 1);
    PFP_TRY(device_compact(c, nullptr, valid, ND, vlist, wposs, d_cnt + 7));
    uint32_t nv32 = 0; PFP_TRY(d2h_u32(c, d_cnt + 7, &nv32));
    const uint64_t nv = nv32;
    uint4 *srec; PFP_ALLOC_HI(c, srec, uint4, nv);
    PFP_LAUNCH(c, K_REC_PARSE, nv * 12, k_dr_heads, nblocks(nv, BLOCK), (const uint32_t *)vlist, (const uint32_t *)srank2, nv, head);
    PFP_HIP(c, hipMemsetAsync(d_cnt, 0, 32, c->stream));
    PFP_TRY(device_compact(c, nullptr, head, nv, chead, wposs, d_cnt));
    PFP_LAUNCH(c, K_REC_PARSE, nv * 44, k_dr_slot_records, nblocks(nv, BLOCK), (const uint32_t *)gsa2, (const uint32_t *)vlist, (const uint32_t *)rowoff, (const uint32_t *)head, (const uint32_t *)wposs, (const uint32_t *)wd2,
               (const uint32_t *)wstart, (const uint32_t *)woff, nv, srec);
    uint32_t nc32 = 0; PFP_TRY(d2h_u32(c, d_cnt, &nc32));
    const uint64_t nc = nc32;
    PFP_LAUNCH(c, K_REC_PARSE, nc * 24, k_rs_class_rows, nblocks(nc + 1, BLOCK), (const uint32_t *)chead, (const uint4 *)srec, nc, nv, N - 1, crow, chead + nc);
    uint32_t tile_rows = c->tun.parse_rec_tile_rows ? c->tun.parse_rec_tile_rows : (uint32_t)RS_TILE;
    if (tile_rows > (uint32_t)RS_TILE) tile_rows = RS_TILE;
    if (tile_rows < 2) tile_rows = 2;
    const int keybits = bits_for(n2);
    uint32_t *bigc, *okey, *cstart;
    PFP_ALLOC_HI(c, bigc, uint32_t, nc + 1); PFP_ALLOC_HI(c, okey, uint32_t, N); PFP_ALLOC_HI(c, cstart, uint32_t, N + 1);
    PFP_HIP(c, hipMemsetAsync(cstart, 0, (N + 1) * 4, c->stream));
    const unsigned ga = nblocks(N - 1, RS_TILE / 2);
    PFP_LAUNCH(c, K_REC_ASSEMBLE, (N - 1) * 16 + nv * 16, (k_rs_assemble<false, true>), ga, (const uint4 *)srec, (const uint32_t *)chead, (const uint32_t *)crow, (uint32_t)nc, (const uint32_t *)ikey, (const uint32_t *)ipos,
               keybits, tile_rows, gsa, (uint32_t *)nullptr, bigc, d_cnt + 1, okey, sflag);
    PFP_LAUNCH(c, K_MISC, 8, k_rs_first_row, 1, gsa, (uint32_t *)nullptr, N);
    uint32_t nb32 = 0; PFP_TRY(d2h_u32(c, d_cnt + 1, &nb32));
    if (nb32) {
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
                   (const uint32_t *)ikey, (const uint32_t *)ipos, bk0, bv0, 1);
        BitRange br[2] = {{0, keybits}, {32, 32 + bits_for(nb - 1)}};
        uint64_t *sk; uint32_t *sv;
        PFP_TRY(radix_sort_pairs<uint64_t>(c, bk0, bv0, bk1, bv1, nbr, br, 2, &sk, &sv));
        PFP_LAUNCH(c, K_REC_PARSE, nbr * 24, (k_rs_big_store<false, true>), nblocks(nbr, BLOCK), (const uint64_t *)sk, (const uint32_t *)sv, nbr, (const uint32_t *)bigc, (const uint32_t *)bigoff, (const uint32_t *)crow, gsa,
                   (uint32_t *)nullptr, okey, sflag);
    }
    // ---- classes of identical dictionary suffixes, word-start flags
    PFP_LAUNCH(c, K_REC_PARSE, nc * 8, k_dr_mark_class_rows, nblocks(nc, BLOCK), (const uint32_t *)crow, nc, cstart);
    PFP_LAUNCH(c, K_REC_PARSE, N * 12, k_dr_final_heads, nblocks(N, BLOCK), (const uint32_t *)okey, (const uint32_t *)cstart, N, srank);
    PFP_TRY((device_scan<uint32_t, 1>(c, srank, srank, N, nullptr)));
    PFP_HIP(c, hipMemsetAsync(sflag, 0, 1, c->stream));      // row 0 is the final EndOfDict; every other row's flag came out of the assembly (k_rs_assemble / k_rs_big_store)
    PFP_HIP(c, hipStreamSynchronize(c->stream));
    if (verbose) fprintf(stderr, "[pfbwt_hip]   dictionary assembled: %llu slots, %llu D2 classes (%.1f ms)\n", (unsigned long long)nv, (unsigned long long)nc, tm.ms());
    c->arena.release_hi(mk);
    *taken = 1;
    return PFP_OK;
}

} // namespace pfp
