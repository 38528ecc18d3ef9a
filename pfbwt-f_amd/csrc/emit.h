// pfbwt-f_amd/csrc/emit.h -- parse-BWT rows and the BWT/SA emission.
//
// Replaces (reference file:line):
//   * rows of the parse BWT, include/pfparser.hpp:430-462 (bwlast, bwsai, word of each row, ilist);
//   * PrefixFreeBWT::generate_bwt_lcp include/pfbwt.hpp:96-194 with get_word_suflen :83-85 and
//     get_word_ilist :259-268 (rank/select bit vectors become the prefix sums ws[] and F[]);
//   * the CLI's out_fn src/pfbwt-f.cpp:298-328 (row 0 := n, run counting, .ssa/.esa samples).
// The reference walks the dictionary suffix array sequentially and pushes one record at a time
// through a callback.  Here every output row is computed independently ("output-stationary"):
// row o finds its suffix-array slot by binary search over the scanned per-slot counts, its position
// inside a multi-word group by ranking its parse-BWT row q in the other members' ilist ranges.
#pragma once
#include "prims.h"
#include "parse.h"

namespace pfp {

// pfparser.hpp:430-451.  SAP = suffix array of ranks+[0] (m+1 entries), P = 1-based ranks, sai = ye.
// Row i with s = SAP[i] needs last[s-2], sai[s-1] and P[s-1]: three random reads of three arrays.  They are packed first
// (streaming) into ONE 16-byte record per phrase, rec[j] = { sai[j], P[j], last[j-1] (last[m-1] for j = 0) }, so that a
// row costs one random 16-byte gather (r01: 69-132 GB of HBM traffic per launch for 7.8 GB of algorithmic bytes).
__global__ __launch_bounds__(BLOCK) void k_pbwt_pack(const uint32_t *P, const uint8_t *last, const tpos_t *sai, uint64_t m, uint4 *rec)
{
    const uint64_t j = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (j >= m) return;
    const uint64_t v = sai ? (uint64_t)sai[j] : 0ULL;
    rec[j] = make_uint4((uint32_t)v, (uint32_t)(v >> 32), P[j], (uint32_t)last[j ? j - 1 : m - 1]);
}
// The emission of a whole-word slot writes bwlast[ilist[k]] for the word's occurrences k (pfbwt.hpp:116-128): 325 M random byte
// reads on S-32G (12 ms).  With fewer than 2^29 parse rows the byte travels through the sort that makes ilist instead, as a 3-bit
// code in the top bits of the row id: 0 (the row of the end-of-string phrase), Dollar, '-', A, C, G, N, T -- the bytes a normalised
// text holds.  Any other byte sets *bad and the caller gathers.
constexpr int BWL_SHIFT = 29;
__device__ __forceinline__ uint32_t bwl_code(uint32_t c) { return c == 0 ? 0u : c == Dollar ? 1u : c == '-' ? 2u : c == 'A' ? 3u : c == 'C' ? 4u : c == 'G' ? 5u : c == 'N' ? 6u : c == 'T' ? 7u : 8u; }
__device__ __forceinline__ uint8_t bwl_byte(uint32_t code) { return code == 0 ? (uint8_t)0 : code == 1 ? Dollar : code == 2 ? (uint8_t)'-' : code == 3 ? (uint8_t)'A' : code == 4 ? (uint8_t)'C' : code == 5 ? (uint8_t)'G' : code == 6 ? (uint8_t)'N' : (uint8_t)'T'; }
__global__ __launch_bounds__(BLOCK) void k_pbwt_rows(const uint32_t *SAP, const uint4 *rec, uint64_t m, uint8_t *bwlast, tpos_t *bwsai, uint32_t *W, uint32_t *rowid, int pack, uint32_t *bad)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i > m) return;
    const uint32_t s = SAP[i];
    if (s == 0) { bwlast[i] = 0; if (bwsai) bwsai[i] = 0; W[i] = 0; rowid[i] = (uint32_t)i; return; }
    const uint4 R = rec[s - 1];
    bwlast[i] = (uint8_t)R.w;                                   // last[s - 2], last[m - 1] when s == 1 (:443-449)
    if (bwsai) bwsai[i] = (tpos_t)(((uint64_t)R.y << 32) | R.x);
    W[i] = R.z;
    uint32_t id = (uint32_t)i;
    if (pack) { const uint32_t code = bwl_code(R.w & 0xFFu); if (code > 7u) *bad = 1u; else id |= code << BWL_SHIFT; }
    rowid[i] = id;
}
// the sorted (packed) row ids -> ilist and the bytes in ilist order
__global__ __launch_bounds__(BLOCK) void k_ilist_split(const uint32_t *packed, uint64_t n, uint32_t *ilist, uint8_t *bwl_il)
{
    const uint64_t k = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (k >= n) return;
    const uint32_t v = packed[k];
    ilist[k] = v & ((1u << BWL_SHIFT) - 1u); bwl_il[k] = bwl_byte(v >> BWL_SHIFT);
}
__global__ __launch_bounds__(BLOCK) void k_bwsai_by_ilist(const uint32_t *ilist, const tpos_t *bwsai, uint64_t n, tpos_t *out)
{
    const uint64_t k = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (k < n) out[k] = bwsai[ilist[k]];
}
__global__ __launch_bounds__(BLOCK) void k_ilist_gather(const uint32_t *packed, uint64_t n, uint32_t mask, const uint8_t *bwlast, uint32_t *ilist, uint8_t *bwl_il)
{
    const uint64_t k = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (k >= n) return;
    const uint32_t q = packed[k] & mask;
    ilist[k] = q; bwl_il[k] = bwlast[q];
}

__global__ __launch_bounds__(BLOCK) void k_u32_add_store(const uint32_t *in, uint64_t n, uint32_t add, uint32_t *out)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) out[i] = in[i] + add;
}

// Word ranks (sort_dict + generate_ranks, pfparser.hpp:494-517) from the dictionary suffix sort: the
// class-head slot of a word's first byte orders the words (two distinct words are never byte-identical,
// so their whole-word suffixes sit in different classes).  keys = grank[ws[id]], sorted -> rank.
__global__ __launch_bounds__(BLOCK) void k_wordstart_keys(const uint32_t *ws, const uint2 *grank, uint64_t dwords, uint32_t *keys, uint32_t *vals)
{
    const uint64_t id = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (id >= dwords) return;
    keys[id] = grank[ws[id]].x; vals[id] = (uint32_t)id;
}
__global__ __launch_bounds__(BLOCK) void k_word_rank(const uint32_t *sorted_ids, uint64_t dwords, const uint32_t *occw, uint32_t *wrank, uint32_t *idofrank, uint32_t *occ)
{
    const uint64_t r = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r >= dwords) return;
    const uint32_t id = sorted_ids[r];
    wrank[id] = (uint32_t)r; idofrank[r] = id; occ[r] = occw[id];
}
// word ranks from the word-start flags of the slots (text-round dictionary sort): the r-th flagged slot holds the word of rank r.
// Flags are counted per tile of WF_TILE slots (16 per thread), the tile counts scanned, the ranks inside a tile by a block scan --
// the flags are bytes and one in a hundred is set: widening 3.4 G of them to words for a device-wide scan moved 45 GB for nothing.
constexpr int WF_PER_THREAD = 16, WF_TILE = BLOCK * WF_PER_THREAD;
__device__ __forceinline__ uint32_t wf_mask16(const uint8_t *sflag, uint64_t i0, uint64_t n)
{
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < WF_PER_THREAD; ++k) if (i0 + k < n && sflag[i0 + k]) m |= 1u << k;
    return m;
}
__global__ __launch_bounds__(BLOCK) void k_flag_tile_count(const uint8_t *sflag, uint64_t n, uint32_t *tilecnt)
{
    __shared__ uint32_t red[4];
    const uint64_t i0 = ((uint64_t)blockIdx.x * BLOCK + threadIdx.x) * WF_PER_THREAD;
    uint32_t tot;
    (void)block_excl_sum((uint32_t)__popc(wf_mask16(sflag, i0, n)), red, &tot);
    if (threadIdx.x == 0) tilecnt[blockIdx.x] = tot;
}
__global__ __launch_bounds__(BLOCK) void k_word_rank_flags(const uint8_t *sflag, const uint32_t *tilebase, const uint32_t *SA, const uint32_t *wordid, uint64_t dsize, uint32_t dwords,
                                                         const uint32_t *occw, uint32_t *wrank, uint32_t *idofrank, uint32_t *occ)
{
    __shared__ uint32_t red[4];
    const uint64_t i0 = ((uint64_t)blockIdx.x * BLOCK + threadIdx.x) * WF_PER_THREAD;
    uint32_t m = wf_mask16(sflag, i0, dsize), tot;
    uint32_t r = tilebase[blockIdx.x] + block_excl_sum((uint32_t)__popc(m), red, &tot);
    while (m) {
        const int k = __ffs((int)m) - 1; m &= m - 1;
        const uint32_t id = wordid[SA[i0 + k]];
        if (r < dwords && id < dwords) { wrank[id] = r; idofrank[r] = id; occ[r] = occw[id]; }
        ++r;
    }
}
__global__ __launch_bounds__(BLOCK) void k_parse_ranks(const uint32_t *pid, const uint32_t *wrank, uint64_t m, uint32_t *parse)
{
    const uint64_t j = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (j < m) parse[j] = wrank[pid[j]] + 1u;    // generate_ranks, pfparser.hpp:504-517 (1-based)
}
__global__ __launch_bounds__(BLOCK) void k_sorted_lengths(const uint32_t *ws, const uint32_t *idofrank, uint64_t dwords, uint32_t *len1, tpos_t *srcstart)
{
    const uint64_t r = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r >= dwords) return;
    const uint32_t id = idofrank[r];
    len1[r] = ws[id + 1] - ws[id]; srcstart[r] = ws[id];
}

// --pfbwt-only: index a loaded .dict image: flag EndOfWord bytes
__global__ __launch_bounds__(BLOCK) void k_eow_flags(const uint8_t *D, uint64_t dsize, uint32_t *flag)
{
    const uint64_t x = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (x < dsize) flag[x] = D[x] == EndOfWord ? 1u : 0u;
}
__global__ __launch_bounds__(BLOCK) void k_ws_from_flags(const uint8_t *D, uint64_t dsize, const uint32_t *wordid, uint32_t *ws)
{
    const uint64_t x = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (x >= dsize) return;
    if (x == 0) ws[0] = 0;
    if (D[x] == EndOfWord) ws[wordid[x] + 1] = (uint32_t)x + 1u;
}

// ---- emission -----------------------------------------------------------------------------------
struct EmitArgs {
    const uint8_t *D; uint64_t dsize; uint32_t dwords; int w;
    const uint32_t *SA, *ws, *wrank /*nullable*/, *occ, *F, *ilist;
    const uint32_t *srank;  // per slot: first slot of its class of equal suffixes (kept by the dictionary suffix sort)
    const tpos_t *bwsai;    // nullable
    const uint2 *posinfo;   // nullable (only without prec)
    const uint4 *prec;      // nullable: per dictionary offset { unused, first ilist index of its word, occ of its word, suffix length | code of the preceding byte << 26 | whole word << 30 | inside a word << 31 }
    const uint32_t *wordid; // per dictionary offset: word id
    int use_prec;
    int use_e0;             // full SA wanted, positions fit 32 bits: prec.x = bwsai of a word's occurrence where the word occurs once (carried to the rows through s_g0 / sinfo.z)
    const uint4 *winfo;     // per word id: { first byte, offset of its EndOfWord, first ilist index F[rank], occ[rank] }
    const uint8_t *bwlast;
    const uint8_t *bwl_il;  // nullable: bwlast in ilist order
    const tpos_t *bwsai_il; // nullable: bwsai in ilist order (full SA wanted): a row of a one-member slot then needs no parse row at all
    const void *EB;         // exclusive scan of the per-slot row counts: uint32_t, or uint64_t when n+1 >= 2^32 (template EBT)
    const uint32_t *s_sl;   // per slot: suffix length
    const uint32_t *s_fb;   // per slot: first ilist index of the slot's word (F[rank])
    const uint8_t *s_fl;    // per slot: SF_* flags
    const uint8_t *s_pc;    // per slot: BWT byte of a proper-suffix slot (preceding dictionary byte, 0 after the first Dollar)
    const uint32_t *s_g0;   // per slot: first slot of its group of equal suffixes
    const uint32_t *gk;     // per group head slot: number of members
    const void *cnt;        // per slot: rows it produces (EBT)
    const uint4 *sinfo;     // per slot, packed for the row kernel: { s_fb, rows, s_g0, members of the group (24 bits) | SF_* flags << 24 }
    const uint32_t *tile_slot;  // slot that holds output row b * EMIT_TILE, b = 0 .. ceil(nout / EMIT_TILE) (last entry: dsize - 1)
    // rows of groups with many members are not ranked one by one: they are collected here and sorted by (group, q)
    uint64_t *big_keys; uint32_t *big_vals; unsigned long long *big_count;  // big_count[1] != 0: list overflow
    uint64_t big_cap, big_total;
    uint64_t nout, n;
    uint64_t e0, e1;        // rows (in enumeration order) this launch walks
    uint64_t w0, w1;        // output positions this launch may write: [w0, w1) -> buffer index pos - w0 (multi-GPU slices)
    // Run-aware emission (no full SA wanted): rows of a group whose members all have the same preceding byte are one run
    // of that byte -- they are written by k_fill without looking at the occurrence lists.  Only the rows of the other
    // ("special") slots -- whole words, groups with a whole-word member, groups with two or more distinct preceding
    // bytes -- are enumerated row by row, through a compacted list of them: elist[j] = slot, cpos[slot] = number of
    // special slots in front of it (= j for a special slot), ENB[j] = exclusive scan of their counts (ecount + 1 entries),
    // etile_slot = tile table over j, qspec = their parse rows (index: ENB[cpos[group head]] - q0 + position inside
    // the group).  special == 0: ENB == EB, elist == nullptr (identity), every row is enumerated.
    const void *ENB; const uint32_t *etile_slot, *elist, *cpos; uint32_t ecount; int special; uint64_t q0;
    const uint32_t *qspec;  // samples-only mode: parse rows of the special rows of this window
    const uint32_t *gqf, *gql;   // per head slot of a uniform multi-member group: parse row of its first / last output row
    // group-stationary route of the special rows (k_emit_groups): what it leaves to k_emit -- gleft[j] != 0: the group whose head is
    // the j-th special slot; tile_left[t] != 0: enumeration tile t of this launch holds rows of such a group.  nullptr: k_emit walks everything
    uint8_t *gleft, *tile_left; uint32_t group_rows_cap; uint32_t rank_members_max;   // groups of more members are not ranked by bisection inside a batch: they go to the LDS sort
    const uint4 *cinfo;          // per special slot j (k_special_pack): { first ilist index, members of its group, its index inside the group, preceding byte | SF_* flags << 8 }
    const unsigned long long *cgb;   // per special slot: output row of the first row of its group
    const uint32_t *town;        // per enumeration tile t: head (index of special slots) of the first group that starts at or behind row t * EMIT_TILE
    uint32_t *lglist; unsigned long long *lgcount; uint64_t lgcap; int qpasses /*8-bit digits that hold a parse row*/;   // groups of more rows than a batch holds, taken one per workgroup by k_emit_groups_large (heads as indices of special slots)
    unsigned long long *gstat;   // PFP_VERBOSE: rows left to k_emit by reason [0] whole-word member, [1] sort route, [2] too many rows, [3] too many slots; [4..11] rows of left groups by log4 of the group's rows
};
constexpr uint8_t SF_MULTI = 1, SF_FULL = 2, SF_BIG = 4, SF_GFULL = 8, SF_NONUNI = 16, SF_E0 = 32;   // E0: a one-member slot of a word that occurs once -- s_g0 / sinfo.z hold bwsai of that occurrence (texts < 2^32), not a head slot   // GFULL: some member of the group is a whole word; NONUNI: members with different preceding bytes
__device__ __forceinline__ bool slot_is_special(uint32_t fl) { return (fl & (SF_FULL | SF_GFULL | SF_NONUNI)) != 0; }
constexpr int EG_SLOTS = 2 * BLOCK;          // slots (members) a batch / an LDS-sorted group may have (k_emit_groups)
constexpr uint32_t BIG_GROUP_MEMBERS = 64;  // groups with more members than this take the sort route (measured: below ~64 ranking is faster)
// posinfo[x] = { word id of dictionary offset x | 4-bit code of D[x-1] << 28 , class-head slot of x }: one 8-byte
// gather per slot instead of three separate random reads (wordid, grank, D[x-1])
constexpr uint32_t WID_MASK = 0x0FFFFFFFu;
__device__ __forceinline__ uint32_t dict_code4(uint32_t c) { return c <= 2 ? c : (c == '-') ? 3u : (c == 'A') ? 4u : (c == 'C') ? 5u : (c == 'G') ? 6u : (c == 'N') ? 7u : 8u; }
__device__ __forceinline__ uint8_t dict_byte4(uint32_t code) { return code <= 2 ? (uint8_t)code : code == 3 ? (uint8_t)'-' : code == 4 ? (uint8_t)'A' : code == 5 ? (uint8_t)'C' : code == 6 ? (uint8_t)'G' : code == 7 ? (uint8_t)'N' : (uint8_t)'T'; }
__global__ __launch_bounds__(BLOCK) void k_pack_winfo(const uint32_t *ws, const uint32_t *wrank /*nullable*/, const uint32_t *occ, const uint32_t *F, uint64_t dwords, uint4 *winfo)
{
    const uint64_t id = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (id >= dwords) return;
    const uint32_t rk = wrank ? wrank[id] : (uint32_t)id;
    winfo[id] = make_uint4(ws[id], ws[id + 1] - 1u, F[rk], occ[rk]);
}
__global__ __launch_bounds__(BLOCK) void k_pack_posinfo(const uint8_t *D, const uint32_t *wordid, uint64_t dsize, uint2 *posinfo)
{
    const uint64_t x = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (x >= dsize) return;
    posinfo[x] = make_uint2(wordid[x] | (dict_code4(x ? D[x - 1] : 0u) << 28), 0u);
}

// prec[x]: everything k_emit_slots needs of dictionary offset x in ONE 16-byte record, written in text order (the word record
// is read once per word there).  A slot then costs one random gather instead of two dependent ones (posinfo[x], then
// winfo[word of x]) -- on a non-repetitive text (S-3G: 3.4 G slots) that chain was 166 ms.  Suffix lengths must fit 26 bits
// (words of 64 Mbase and more: the caller keeps the two-gather route).
constexpr uint32_t PREC_SL_BITS = 26, PREC_FULL = 1u << 30, PREC_VALID = 1u << 31;
__global__ __launch_bounds__(BLOCK) void k_pack_prec(const uint8_t *D, const uint32_t *wordid, const uint4 *winfo, uint64_t dsize, uint32_t dwords, uint4 *prec, const tpos_t *bwsai_il /*nullable*/)
{
    const uint64_t x = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (x >= dsize) return;
    const uint32_t id = wordid[x];
    uint4 R = make_uint4(0u, 0u, 0u, 0u);
    if (id < dwords) {
        const uint4 W = winfo[id];
        uint32_t code = 0, full = 0;
        if ((uint32_t)x == W.x) full = PREC_FULL;
        else { code = dict_code4(D[x - 1]); if (code == Dollar && (uint32_t)x - 1u == W.x) code = 0; }      // pfbwt.hpp:132 "gsa[i]-1 ? dict[..] : 0"
        R.y = W.z; R.z = W.w; R.w = (W.y - (uint32_t)x) | (code << PREC_SL_BITS) | full | PREC_VALID;
        if (bwsai_il && W.w == 1u) R.x = (uint32_t)bwsai_il[W.z];      // the word occurs once: the text position its rows' SA values count from (one read per word: consecutive offsets share it)
    }
    prec[x] = R;
}
__global__ __launch_bounds__(BLOCK) void k_max_word_length(const uint32_t *ws, uint64_t dwords, uint32_t *out)
{
    __shared__ uint32_t red[4];
    const uint64_t id = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    uint32_t tot;
    (void)block_incl_max(id < dwords ? ws[id + 1] - ws[id] : 0u, red, &tot);
    if (threadIdx.x == 0) atomicMax(out, tot);
}

__device__ __forceinline__ uint32_t word_rank_of(const EmitArgs &a, uint32_t id) { return a.wrank ? a.wrank[id] : id; }

template <typename T> __device__ __forceinline__ uint32_t upper_bound_t(const T *a, uint32_t n, T x)
{   // first index with a[idx] > x
    uint32_t lo = 0, hi = n;
    while (lo < hi) { uint32_t mid = lo + ((hi - lo) >> 1); if (a[mid] <= x) lo = mid + 1; else hi = mid; }
    return lo;
}

// Per suffix-array slot (the random gathers happen here, once per slot, not once per output row):
// cnt = rows produced (occ of the word if suff_len > w, pfbwt.hpp:114), suffix length, ilist base,
// preceding byte, whole-word flag (pfbwt.hpp:116), multi-word-group flag (pfbwt.hpp:137).
template <typename EBT> __global__ __launch_bounds__(BLOCK) void k_emit_slots(EmitArgs a, EBT *cnt, unsigned long long *hard_rows, uint32_t *s_sl, uint32_t *s_fb, uint8_t *s_fl, uint8_t *s_pc,
                                                                             uint32_t *s_g0, uint32_t *gk /*per head slot: members*/, uint8_t *gfl /*per head slot: has a whole-word member*/,
                                                                             uint8_t *gnu /*per head slot: members with different preceding bytes*/)
{
    __shared__ uint8_t hd[BLOCK + 1];   // is slot (block base + t) the head of its class of equal suffixes
    __shared__ uint8_t pcl[BLOCK + 1];  // preceding byte of slot (block base + t); 0xFF: whole word (its group is special anyway)
    __shared__ uint32_t red[4];
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    const bool valid = i < a.dsize;
    uint32_t x = 0; uint2 P = make_uint2(0, 0); uint4 R = make_uint4(0, 0, 0, 0);
    if (valid) { x = a.SA[i]; if (a.prec) R = a.prec[x]; else P = a.posinfo[x]; P.y = a.srank[i]; }
    hd[threadIdx.x] = (valid && P.y == (uint32_t)i) ? 1 : 0;
    if (threadIdx.x == 0) {
        const uint64_t nx = (uint64_t)(blockIdx.x + 1) * BLOCK;
        uint8_t h = 1, hp = 0xFF;
        if (nx < a.dsize) {
            const uint32_t xn = a.SA[nx];
            if (a.prec) {
                const uint4 Rn = a.prec[xn];
                h = a.srank[nx] == (uint32_t)nx ? 1 : 0;
                if (!h && (Rn.w & PREC_VALID) && !(Rn.w & PREC_FULL)) hp = dict_byte4((Rn.w >> PREC_SL_BITS) & 15u);
            } else {
            const uint2 Pn = a.posinfo[xn];
            h = a.srank[nx] == (uint32_t)nx ? 1 : 0;
            if (!h) {   // the first slot of the next block continues a group of this block: its preceding byte
                const uint32_t idn = Pn.x & WID_MASK;
                if (idn < a.dwords) {
                    const uint32_t wsn = a.winfo[idn].x;
                    if (xn != wsn) { hp = dict_byte4(Pn.x >> 28); if (hp == Dollar && xn - 1 == wsn) hp = 0; }
                }
            }
            }
        }
        hd[BLOCK] = h; pcl[BLOCK] = hp;
    }
    __syncthreads();
    uint32_t c = 0, sl = 0, fb = 0; uint8_t fl = 0, pc = 0;
    if (a.prec) {
        if (valid && (R.w & PREC_VALID)) {
            sl = R.w & ((1u << PREC_SL_BITS) - 1u);
            if (sl > (uint32_t)a.w) {
                c = R.z; fb = R.y;
                if (!hd[threadIdx.x] || (i + 1 < a.dsize && !hd[threadIdx.x + 1])) fl |= SF_MULTI;   // group of >= 2 equal suffixes (pfbwt.hpp:137)
                if (R.w & PREC_FULL) fl |= SF_FULL; else pc = dict_byte4((R.w >> PREC_SL_BITS) & 15u);
                if (a.use_e0 && c == 1u && !(fl & SF_MULTI)) { fl |= SF_E0; P.y = R.x; }      // s_g0 of such a slot is never asked for a head
            }
        }
    } else {
    const uint32_t id = P.x & WID_MASK;
    if (valid && id < a.dwords) {
        const uint4 W = a.winfo[id];
        const uint32_t wsid = W.x;
        sl = W.y - x;
        if (sl > (uint32_t)a.w) {
            c = W.w; fb = W.z;
            if (!hd[threadIdx.x] || (i + 1 < a.dsize && !hd[threadIdx.x + 1])) fl |= SF_MULTI;   // group of >= 2 equal suffixes (pfbwt.hpp:137)
            if (x == wsid) fl |= SF_FULL;
            else { pc = dict_byte4(P.x >> 28); if (pc == Dollar && x - 1 == wsid) pc = 0; }   // pfbwt.hpp:132 "gsa[i]-1 ? dict[..] : 0"
        }
    }
    }
    pcl[threadIdx.x] = (fl & SF_FULL) ? (uint8_t)0xFF : pc;
    __syncthreads();
    if (valid) {
        cnt[i] = (EBT)c; s_sl[i] = sl; s_fb[i] = fb; s_fl[i] = fl; s_pc[i] = pc; s_g0[i] = P.y;
        if (fl & SF_MULTI) {
            const bool lastm = i + 1 >= a.dsize || hd[threadIdx.x + 1];
            if (lastm) gk[P.y] = (uint32_t)i - P.y + 1u;     // last member: group size
            if (fl & SF_FULL) gfl[P.y] = 1;
            // two neighbouring members with different preceding bytes: the group's rows are not one run (pfbwt.hpp:146-159)
            else if (!lastm && pcl[threadIdx.x + 1] != 0xFF && pcl[threadIdx.x + 1] != pc) gnu[P.y] = 1;
        }
    }
    uint32_t tot;   // rows that sit in multi-word groups (the reference's "hard" bookkeeping, pfbwt.hpp:188)
    (void)block_excl_sum((fl & SF_MULTI) ? c : 0u, red, &tot);
    if (threadIdx.x == 0 && tot) atomicAdd(hard_rows, (unsigned long long)tot);
}

__device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t *a, uint32_t n, uint32_t x)
{   // number of entries < x
    uint32_t lo = 0, hi = n;
    while (lo < hi) { uint32_t mid = lo + ((hi - lo) >> 1); if (a[mid] < x) lo = mid + 1; else hi = mid; }
    return lo;
}

// Groups with more than BIG_GROUP_MEMBERS members (on a 1000-haplotype panel the suffixes of length w+1, w+2 have
// tens of members and thousands of rows): ranking every row in every other member's ilist costs O(members) bisections
// per row.  Their rows are instead collected as (group head slot, q) keys, sorted, and placed by their index inside
// the group.  Groups with a whole-word member keep the ranking route (reference quirk handling lives there).
template <typename EBT> __global__ __launch_bounds__(BLOCK) void k_big_mark(const EBT *cnt, const uint32_t *s_g0, const uint32_t *gk, const uint8_t *gfl, const uint8_t *gnu, const uint32_t *s_fb, const uint32_t *ilist, uint64_t dsize, uint32_t min_members,
                                                                            int runaware, uint8_t *s_fl, uint4 *sinfo, EBT *cnt2 /*runaware: rows of the special slots*/, uint32_t *gqf, uint32_t *gql, unsigned long long *big_rows,
                                                                            const EBT *EB /*exclusive scan of cnt*/, const EBT *total, uint32_t max_rows /*0: no limit; else groups of more rows take the sort route*/, int many_in_lds)
{
    __shared__ unsigned long long red[4];
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    unsigned long long mine = 0;
    if (i < dsize) {
        uint8_t fl = s_fl[i];
        const uint32_t g0 = s_g0[i];
        const uint32_t c = (uint32_t)cnt[i], fb = s_fb[i];
        uint32_t k = 1;
        if (fl & SF_MULTI) {
            k = gk[g0];
            if (gfl[g0]) fl |= SF_GFULL;
            else {
                if (gnu[g0]) fl |= SF_NONUNI;
                // without a full SA only the groups that are not one run of a byte are merged at all
                bool big = k > min_members;
                if (max_rows && (fl & SF_NONUNI)) {
                    const uint64_t grows = (uint64_t)(g0 + k < dsize ? EB[g0 + k] : *total) - (uint64_t)EB[g0];
                    if (!big) big = grows > (uint64_t)max_rows;
                    else if (many_in_lds && k <= (uint32_t)EG_SLOTS && grows <= (uint64_t)max_rows) big = false;      // many members, but the group fits the LDS sort of k_emit_groups_large, which does not care how many lists it merges
                }
                if (big && (!runaware || (fl & SF_NONUNI))) { fl |= SF_BIG; mine = (unsigned long long)cnt[i]; }
            }
            s_fl[i] = fl;
        }
        sinfo[i] = make_uint4(fb, c, g0, (k < 0xFFFFFFu ? k : 0xFFFFFFu) | ((uint32_t)fl << 24));
        if (runaware) {
            const bool sp = slot_is_special(fl);
            cnt2[i] = sp ? cnt[i] : (EBT)0;
            if (!sp && (fl & SF_MULTI) && c) {   // one run: only its first and last row can be sampled (smallest / largest parse row of the members)
                atomicMin(&gqf[g0], ilist[fb]); atomicMax(&gql[g0], ilist[fb + c - 1u]);
            }
        }
    }
    unsigned long long tot;
    (void)block_excl_sum(mine, red, &tot);
    if (threadIdx.x == 0 && tot) atomicAdd(big_rows, tot);
}
// PFP_VERBOSE only: rows by (members of their group, rows of their group) in power-of-two buckets
template <typename EBT> __global__ __launch_bounds__(BLOCK) void k_group_stats(const EBT *cnt, const EBT *EB, const uint32_t *s_g0, const uint32_t *gk, const uint8_t *s_fl, uint64_t dsize, uint64_t nout, unsigned long long *hist /*[8][8]*/)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= dsize || !cnt[i]) return;
    uint32_t k = 1; uint64_t tg = (uint64_t)cnt[i];
    if (s_fl[i] & SF_MULTI) { const uint32_t g0 = s_g0[i]; k = gk[g0]; tg = (g0 + k < dsize ? (uint64_t)EB[g0 + k] : nout) - (uint64_t)EB[g0]; }
    int kb = 0; while (kb < 7 && (2u << kb) <= k) ++kb;          // k: 1, 2-3, 4-7, 8-15, 16-31, 32-63, 64-127, 128+
    int tb = 0; while (tb < 7 && (1024ull << (2 * tb)) <= tg) ++tb;   // rows: <1K, <4K, <16K, <64K, <256K, <1M, <4M, more
    atomicAdd(&hist[kb * 8 + tb], (unsigned long long)cnt[i]);
}
__global__ __launch_bounds__(BLOCK) void k_big_heads(const uint64_t *keys, uint64_t nb, uint32_t *headidx)
{
    const uint64_t t = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t < nb) headidx[t] = (t == 0 || (keys[t] >> 32) != (keys[t - 1] >> 32)) ? (uint32_t)t : 0u;
}
template <typename SAT, typename EBT> __global__ __launch_bounds__(BLOCK) void k_big_place(EmitArgs a, const uint64_t *keys, const uint32_t *vals, const uint32_t *tg, uint64_t nb, uint8_t *bwt, SAT *sa, uint32_t *qrow)
{
    const uint64_t t = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= nb) return;
    const uint32_t g0 = (uint32_t)(keys[t] >> 32), q = (uint32_t)keys[t], i = vals[t];
    const uint64_t pos = (uint64_t)reinterpret_cast<const EBT *>(a.EB)[g0] + (t - tg[t]);
    if (a.special && qrow) qrow[(uint64_t)reinterpret_cast<const EBT *>(a.ENB)[a.cpos[g0]] - a.q0 + (t - tg[t])] = q;   // every enumerated row, also outside the window
    if (pos < a.w0 || pos >= a.w1) return;
    bwt[pos - a.w0] = a.s_pc[i];                                // no whole-word member in these groups
    if (sa) {
        SAT v = (SAT)((SAT)(a.bwsai[q]) - (SAT)a.s_sl[i]);
        if (pos == 0) v = (SAT)a.n;
        sa[pos - a.w0] = v;
    }
    if (qrow && !a.special) qrow[pos - a.w0] = q;
}

// Position of a row inside a group without whole-word members: gb + r + the number of occurrences of the OTHER members
// that precede parse row q (pfbwt.hpp:137-181: the members' ilists are merged by value).  Everything comes from the
// packed per-slot records (no SA / posinfo / winfo gathers); the bisections of up to four members run interleaved so
// that their loads are in flight together.
template <int W> __device__ __forceinline__ uint32_t rank_in_members(const EmitArgs &a, const uint32_t *mem /*W slots, ~0u = none*/, uint32_t q)
{
    const uint32_t *base[W]; uint32_t lo[W], len[W];
#pragma unroll
    for (int j = 0; j < W; ++j) {
        const uint4 S = mem[j] != ~0u ? a.sinfo[mem[j]] : make_uint4(0, 0, 0, 0);      // one 16-byte load per member: list start and length
        base[j] = a.ilist + S.x; lo[j] = 0; len[j] = S.y;
    }
    bool any = false;
#pragma unroll
    for (int j = 0; j < W; ++j) any |= len[j] != 0;
    while (any) {      // number of list entries < q, all W bisections in step
        uint32_t v[W], half[W];
#pragma unroll
        for (int j = 0; j < W; ++j) { half[j] = len[j] >> 1; v[j] = len[j] ? base[j][lo[j] + half[j]] : 0u; }
        any = false;
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const bool less = len[j] && v[j] < q;
            lo[j] = less ? lo[j] + half[j] + 1 : lo[j];
            len[j] = less ? len[j] - half[j] - 1 : half[j];
            any |= len[j] != 0;
        }
    }
    uint32_t before = 0;
#pragma unroll
    for (int j = 0; j < W; ++j) before += lo[j];
    return before;
}
// the same for W lists with their own q each (the rows a thread has in flight are ranked together)
template <int W> __device__ __forceinline__ void rank_lists(const EmitArgs &a, const uint32_t *mem /*W slots, ~0u = none*/, const uint32_t *q, uint32_t *out)
{
    const uint32_t *base[W]; uint32_t lo[W], len[W];
    bool any = false;
#pragma unroll
    for (int j = 0; j < W; ++j) {
        const uint4 S = mem[j] != ~0u ? a.sinfo[mem[j]] : make_uint4(0, 0, 0, 0);
        base[j] = a.ilist + S.x; lo[j] = 0; len[j] = S.y; any |= S.y != 0;
    }
    while (any) {
        uint32_t v[W], half[W];
#pragma unroll
        for (int j = 0; j < W; ++j) { half[j] = len[j] >> 1; v[j] = len[j] ? base[j][lo[j] + half[j]] : 0u; }
        any = false;
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const bool less = len[j] && v[j] < q[j];
            lo[j] = less ? lo[j] + half[j] + 1 : lo[j];
            len[j] = less ? len[j] - half[j] - 1 : half[j];
            any |= len[j] != 0;
        }
    }
#pragma unroll
    for (int j = 0; j < W; ++j) out[j] = lo[j];
}
template <typename EBT> __device__ __forceinline__ uint64_t plain_group_pos(const EmitArgs &a, uint32_t i, uint32_t r, uint32_t q, uint32_t g0, uint32_t k)
{
    uint64_t before = 0;
    if (k == 2) {                     // the common case: one other member, one plain bisection
        const uint32_t other = (i == g0) ? g0 + 1 : g0;
        before = rank_in_members<1>(a, &other, q);
    } else if (k == 3) {
        uint32_t mem[2]; int c = 0;
#pragma unroll
        for (uint32_t j = 0; j < 3; ++j) if (g0 + j != i) mem[c++ & 1] = g0 + j;
        before = rank_in_members<2>(a, mem, q);
    } else {
        for (uint32_t s0 = 0; s0 < k; s0 += 4) {
            uint32_t mem[4];
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) { const uint32_t s = g0 + s0 + j; mem[j] = (s0 + j < k && s != i) ? s : ~0u; }
            before += rank_in_members<4>(a, mem, q);
        }
    }
    return (uint64_t)reinterpret_cast<const EBT *>(a.EB)[g0] + before + r;
}

// position of a row inside a multi-word group (pfbwt.hpp:137-181) and whether a whole-word member
// emits EndOfWord.  gsacak puts byte-identical suffixes in dictionary-position order, i.e. by word rank.
// The reference's loop starts at the FIRST member: if that one is a whole word it is emitted alone
// (:116-128) and the rest forms its own group; otherwise all members are merged by ilist position and a
// whole-word member contributes dict[gsa-1] == EndOfWord as its BWT byte (:140).
template <typename EBT> __device__ __forceinline__ uint64_t multi_group_pos(const EmitArgs &a, uint32_t i, uint32_t r, uint32_t q, bool self_full, bool *full_emits_eow)
{
    const uint32_t g0 = a.s_g0[i];                          // head slot and word of a slot: per-slot array of k_emit_slots, per-offset word ids
    const uint32_t idi = a.wordid[a.SA[i]];
    const uint32_t rk = word_rank_of(a, idi);
    uint64_t before = 0;
    uint32_t first_rk = rk, first_before = 0, first_occ = a.winfo[idi].w; bool first_full = self_full;
    for (uint32_t s = g0; s < a.dsize; ++s) {
        if (!(a.s_fl[s] & SF_MULTI) || a.s_g0[s] != g0) break;      // (s_g0 of a one-member slot may hold a text position, SF_E0)
        if (s == i) continue;
        const uint32_t xs = a.SA[s];
        const uint32_t ids = a.wordid[xs];
        const uint32_t rs = word_rank_of(a, ids);
        const uint4 Ws = a.winfo[ids];
        const uint32_t oc = Ws.w;
        const uint32_t lb = lower_bound_u32(a.ilist + Ws.z, oc, q);
        before += lb;
        if (rs < first_rk) { first_rk = rs; first_before = lb; first_occ = oc; first_full = (xs == Ws.x); }
    }
    const uint64_t gb = reinterpret_cast<const EBT *>(a.EB)[g0];
    *full_emits_eow = false;
    if (first_full) return (first_rk == rk) ? gb + r : gb + first_occ + (before - first_before) + r;
    *full_emits_eow = self_full;
    return gb + before + r;
}

// Output-stationary emission.  A workgroup owns EMIT_TILE consecutive rows; the slots they come from
// are a contiguous range found by two binary searches per workgroup; that slice of EB goes to LDS and
// every row finds its slot there.
#ifndef PFP_EMIT_PER_THREAD
#define PFP_EMIT_PER_THREAD 8
#endif
constexpr int EMIT_PER_THREAD = PFP_EMIT_PER_THREAD;
constexpr int EMIT_TILE = BLOCK * EMIT_PER_THREAD;
#ifndef PFP_EMIT_LDS_SLOTS
#define PFP_EMIT_LDS_SLOTS 4096
#endif
constexpr int EMIT_LDS_SLOTS = PFP_EMIT_LDS_SLOTS;
#ifndef PFP_EMIT_ROWS_IN_FLIGHT
#define PFP_EMIT_ROWS_IN_FLIGHT 2
#endif
constexpr int EMIT_ROWS_IN_FLIGHT = PFP_EMIT_ROWS_IN_FLIGHT;
#ifndef PFP_EMIT_RANK_W
#define PFP_EMIT_RANK_W 2
#endif
constexpr int EMIT_RANK_W = PFP_EMIT_RANK_W;     // member lists per row ranked at a time
static_assert(EMIT_PER_THREAD % EMIT_ROWS_IN_FLIGHT == 0, "rows per thread");

// tile_slot[b] = the slot whose rows include output row b * EMIT_TILE (every slot marks the tile starts it covers)
template <typename EBT> __global__ __launch_bounds__(BLOCK) void k_tile_slots(const EBT *cnt, const EBT *EB, uint64_t dsize, uint64_t ntiles, uint32_t *tile_slot)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i == 0) tile_slot[ntiles] = (uint32_t)(dsize - 1);
    if (i >= dsize) return;
    const uint64_t c = (uint64_t)cnt[i];
    if (!c) return;
    const uint64_t b0 = (uint64_t)EB[i], b1 = b0 + c;
    for (uint64_t b = (b0 + EMIT_TILE - 1) / EMIT_TILE; b * EMIT_TILE < b1; ++b) tile_slot[b] = (uint32_t)i;
}
// last slot with EB <= o, bisecting only between the table entries around o
template <typename EBT> __device__ __forceinline__ uint32_t slot_of_row(const EmitArgs &a, uint64_t o)
{
    const EBT *EB = reinterpret_cast<const EBT *>(a.EB);
    const uint64_t tb = o / EMIT_TILE;
    const uint32_t lo = a.tile_slot[tb], hi = a.tile_slot[tb + 1];
    return lo + upper_bound_t<EBT>(EB + lo, hi - lo + 1u, (EBT)o) - 1u;
}

template <typename SAT, typename EBT> __global__ __launch_bounds__(BLOCK) void k_emit(EmitArgs a, uint8_t *bwt, SAT *sa, uint32_t *qrow /*or: parse-BWT row of every output row (samples-only mode)*/)
{
    __shared__ uint32_t eb[EMIT_LDS_SLOTS];                  // EB[i0 + k] - EB[i0]
    __shared__ uint32_t rs[EMIT_TILE];                       // slot (relative to i0) of every row of the tile
    __shared__ uint32_t red[4];
    const EBT *EB = reinterpret_cast<const EBT *>(a.EB);
    const EBT *ENB = reinterpret_cast<const EBT *>(a.ENB);      // enumeration order: == EB unless only the special slots are walked
    // tiles are aligned to multiples of EMIT_TILE in the global row numbering, so that the slots under a tile come from
    // the precomputed tile_slot table (k_tile_slots) instead of two 27-step bisections of EB per workgroup
    if (a.tile_left && !a.tile_left[blockIdx.x]) return;     // k_emit_groups has written every row of this tile
    const uint64_t tb = a.e0 / EMIT_TILE + blockIdx.x;
    const uint64_t o0 = tb * EMIT_TILE > a.e0 ? tb * EMIT_TILE : a.e0;
    const uint64_t o1 = ((tb + 1) * EMIT_TILE < a.e1) ? (tb + 1) * EMIT_TILE : a.e1;   // exclusive
    const uint32_t i0 = a.etile_slot[tb], i1 = a.etile_slot[tb + 1];   // a superset of the slots of rows [o0, o1)
    const uint32_t ns = i1 - i0 + 1u;
    const uint64_t ebase = (uint64_t)ENB[i0];
    const bool in_lds = ns <= (uint32_t)EMIT_LDS_SLOTS && o1 - ebase < 0xFFFFFFFFULL;
    if (in_lds) for (uint32_t k = threadIdx.x; k < ns; k += BLOCK) eb[k] = (uint32_t)((uint64_t)ENB[i0 + k] - ebase);
    for (int k = threadIdx.x; k < EMIT_TILE; k += BLOCK) rs[k] = 0;
    __syncthreads();
    if (in_lds) {
        // row -> slot without a search per row: every slot marks the row it starts at (the last of the slots that start
        // at the same row wins: slots without rows keep EB unchanged), an inclusive max-scan spreads the marks
        const uint32_t rel0 = (uint32_t)(o0 - ebase), nrow = (uint32_t)(o1 - o0);
        for (uint32_t t = threadIdx.x; t < ns; t += BLOCK) {
            const uint32_t e = eb[t];
            if (e <= rel0) atomicMax(&rs[0], t);
            else if (e - rel0 < nrow) atomicMax(&rs[e - rel0], t);
        }
        __syncthreads();
        uint32_t loc[EMIT_PER_THREAD], run = 0;
#pragma unroll
        for (int j = 0; j < EMIT_PER_THREAD; ++j) { const uint32_t v = rs[threadIdx.x * EMIT_PER_THREAD + j]; run = v > run ? v : run; loc[j] = run; }
        uint32_t tot;
        const uint32_t inc = block_incl_max(run, red, &tot);
        uint32_t prev = __shfl_up(inc, 1);                       // exclusive: the maximum over the threads in front
        if ((threadIdx.x & 63) == 0) prev = 0;
        __shared__ uint32_t wmax[BLOCK / WAVE];
        if ((threadIdx.x & 63) == 63) wmax[threadIdx.x >> 6] = inc;
        __syncthreads();
        if ((threadIdx.x & 63) == 0 && threadIdx.x) prev = wmax[(threadIdx.x >> 6) - 1];
#pragma unroll
        for (int j = 0; j < EMIT_PER_THREAD; ++j) rs[threadIdx.x * EMIT_PER_THREAD + j] = loc[j] > prev ? loc[j] : prev;
        __syncthreads();
    }
    // EMIT_ROWS_IN_FLIGHT rows per thread are taken through the load stages together (slot search, per-slot fields,
    // ilist, bwsai): the kernel is bound by the latency of these dependent loads, not by bandwidth.
#pragma unroll 1
    for (int k = 0; k < EMIT_PER_THREAD; k += EMIT_ROWS_IN_FLIGHT) {
        uint64_t o[EMIT_ROWS_IN_FLIGHT]; uint32_t i[EMIT_ROWS_IN_FLIGHT], r[EMIT_ROWS_IN_FLIGHT], q[EMIT_ROWS_IN_FLIGHT];
        uint4 S[EMIT_ROWS_IN_FLIGHT]; uint8_t fl[EMIT_ROWS_IN_FLIGHT]; bool on[EMIT_ROWS_IN_FLIGHT]; uint64_t sv[EMIT_ROWS_IN_FLIGHT];
#pragma unroll
        for (int u = 0; u < EMIT_ROWS_IN_FLIGHT; ++u) {
            o[u] = o0 + (uint64_t)(k + u) * BLOCK + threadIdx.x;
            on[u] = o[u] < o1;
            const uint64_t oo = on[u] ? o[u] : o0;
            if (in_lds) { const uint32_t j = rs[(uint32_t)(oo - o0)]; i[u] = i0 + j; r[u] = (uint32_t)(oo - ebase) - eb[j]; }
            else { i[u] = upper_bound_t<EBT>(ENB, a.ecount, (EBT)oo) - 1u; r[u] = (uint32_t)(oo - (uint64_t)ENB[i[u]]); }
        }
        uint32_t jc[EMIT_ROWS_IN_FLIGHT];
#pragma unroll
        for (int u = 0; u < EMIT_ROWS_IN_FLIGHT; ++u) jc[u] = i[u];
        if (a.elist) {
#pragma unroll
            for (int u = 0; u < EMIT_ROWS_IN_FLIGHT; ++u) i[u] = a.elist[i[u]];       // index in the list of special slots -> slot
        }
#pragma unroll
        for (int u = 0; u < EMIT_ROWS_IN_FLIGHT; ++u) { S[u] = a.sinfo[i[u]]; fl[u] = (uint8_t)(S[u].w >> 24); }
        if (a.gleft) {      // only the rows of the groups k_emit_groups left alone
#pragma unroll
            for (int u = 0; u < EMIT_ROWS_IN_FLIGHT; ++u) {
                const uint32_t jh = (fl[u] & SF_MULTI) ? jc[u] - (i[u] - S[u].z) : jc[u];
                if (on[u] && !a.gleft[jh]) on[u] = false;
            }
        }
        // parse-BWT row of this occurrence -- a random 4-byte read, skipped where nothing asks for it: with bwsai / bwlast in ilist
        // order (bwsai_il, bwl_il) the rows of one-member slots (nearly all rows of a non-repetitive text) take their SA value and
        // their whole-word byte from the list position itself
#pragma unroll
        for (int u = 0; u < EMIT_ROWS_IN_FLIGHT; ++u) {
            const bool needq = (fl[u] & SF_MULTI) || qrow || ((fl[u] & SF_FULL) && !a.bwl_il) || (sa && !a.bwsai_il);
            q[u] = needq ? a.ilist[S[u].x + r[u]] : 0u;
        }
#pragma unroll
        for (int u = 0; u < EMIT_ROWS_IN_FLIGHT; ++u) sv[u] = (sa && on[u] && !(fl[u] & SF_BIG)) ? ((fl[u] & SF_E0) ? (uint64_t)S[u].z : a.bwsai_il ? (uint64_t)a.bwsai_il[S[u].x + r[u]] : (uint64_t)a.bwsai[q[u]]) : 0ULL;
        // rows in ordinary multi-member groups: the bisections of all rows in flight run in one loop (EMIT_RANK_W lists per
        // row at a time), so that a thread has EMIT_ROWS_IN_FLIGHT * EMIT_RANK_W dependent-load chains going instead of one
        uint32_t before[EMIT_ROWS_IN_FLIGHT], gk_[EMIT_ROWS_IN_FLIGHT], maxk = 0;
#pragma unroll
        for (int u = 0; u < EMIT_ROWS_IN_FLIGHT; ++u) {
            const bool plain = on[u] && (fl[u] & SF_MULTI) && !(fl[u] & (SF_GFULL | SF_BIG));
            gk_[u] = plain ? (S[u].w & 0xFFFFFFu) : 0u; before[u] = 0;
            maxk = gk_[u] > maxk ? gk_[u] : maxk;
        }
        for (uint32_t s0 = 0; s0 < maxk; s0 += EMIT_RANK_W) {
            uint32_t mem[EMIT_ROWS_IN_FLIGHT * EMIT_RANK_W], qq[EMIT_ROWS_IN_FLIGHT * EMIT_RANK_W], res[EMIT_ROWS_IN_FLIGHT * EMIT_RANK_W];
#pragma unroll
            for (int u = 0; u < EMIT_ROWS_IN_FLIGHT; ++u)
#pragma unroll
                for (int j = 0; j < EMIT_RANK_W; ++j) {
                    const uint32_t s = S[u].z + s0 + j;
                    mem[u * EMIT_RANK_W + j] = (s0 + j < gk_[u] && s != i[u]) ? s : ~0u; qq[u * EMIT_RANK_W + j] = q[u];
                }
            rank_lists<EMIT_ROWS_IN_FLIGHT * EMIT_RANK_W>(a, mem, qq, res);
#pragma unroll
            for (int u = 0; u < EMIT_ROWS_IN_FLIGHT; ++u)
#pragma unroll
                for (int j = 0; j < EMIT_RANK_W; ++j) before[u] += res[u * EMIT_RANK_W + j];
        }
#pragma unroll
        for (int u = 0; u < EMIT_ROWS_IN_FLIGHT; ++u) {
            const bool self_full = (fl[u] & SF_FULL) != 0;
            uint64_t pos = o[u];
            bool full_emits_eow = false;
            const bool big = on[u] && (fl[u] & SF_BIG) != 0;
            {   // rows of many-member groups go to the sort list: one atomic per wave, lanes take consecutive entries
                const unsigned long long bm = __ballot(big);
                if (bm) {
                    const int lane = threadIdx.x & 63, leader = __ffsll((long long)bm) - 1;
                    unsigned long long basei = 0;
                    if (lane == leader) basei = atomicAdd(a.big_count, (unsigned long long)__popcll(bm));
                    basei = __shfl(basei, leader);
                    if (big) {
                        const unsigned long long idx = basei + (unsigned long long)__popcll(bm & (lane ? (~0ULL >> (64 - lane)) : 0ULL));
                        if (idx < a.big_cap) { a.big_keys[idx] = ((uint64_t)S[u].z << 32) | q[u]; a.big_vals[idx] = i[u]; }
                        else a.big_count[1] = 1;
                    }
                }
            }
            if (!on[u] || big) continue;
            uint32_t g = i[u];                                          // slot whose EB / ENB entry the row's position counts from
            if (fl[u] & SF_GFULL) { pos = multi_group_pos<EBT>(a, i[u], r[u], q[u], self_full, &full_emits_eow); g = S[u].z; }
            else if (fl[u] & SF_MULTI) { pos = (uint64_t)EB[S[u].z] + before[u] + r[u]; g = S[u].z; }
            else if (a.special) pos = (uint64_t)EB[i[u]] + r[u];
            if (a.special && qrow) qrow[(uint64_t)ENB[a.cpos[g]] - a.q0 + (pos - (uint64_t)EB[g])] = q[u];   // every enumerated row, also outside the window
            const uint8_t c = self_full ? (full_emits_eow ? (uint8_t)EndOfWord : a.bwl_il ? a.bwl_il[S[u].x + r[u]] : a.bwlast[q[u]]) : a.s_pc[i[u]];   // pfbwt.hpp:116-128 / :132
            if (pos < a.w0 || pos >= a.w1) continue;                    // row of a boundary group that lands in another slice
            bwt[pos - a.w0] = c;
            if (sa) {
                SAT v = (SAT)((SAT)sv[u] - (SAT)a.s_sl[i[u]]);          // UPDATE_SA, pfbwt.hpp:87-89 (suff_len :83-85)
                if (pos == 0) v = (SAT)a.n;                             // src/pfbwt-f.cpp:301
                sa[pos - a.w0] = v;
            }
            if (qrow && !a.special) qrow[pos - a.w0] = q[u];
        }
    }
}

// Group-stationary emission of the special rows (run-aware mode; pfbwt.hpp:116-181 for the rows k_fill cannot write as runs).
// k_emit ranks every row of a multi-member group in the other members' occurrence lists by bisections in memory -- ~10 dependent
// loads per row, 52 ms for the 1.3 G special rows of S-32G (0.03 of the HBM roofline, latency-bound).  Here the tile follows the
// GROUPS, not the rows: a workgroup owns the groups whose first enumeration row lies in its stripe of EMIT_TILE rows, takes them in
// batches of whole groups (at most EG_BUF rows, EG_SLOTS slots), reads the members' occurrence lists ONCE, coalesced, into LDS,
// ranks there, and stores bytes and parse rows at their places in the group's stretch of the output.  A whole-word slot is a group of one list (its bytes
// are bwlast[q]: the one random gather left).  Groups that do not fit a batch, groups with a whole-word member (reference quirk,
// multi_group_pos) and sort-route groups are left to k_emit: their heads are marked in gleft[], the enumeration tiles their rows
// touch in tile_left[].
constexpr int EG_BUF = 4096, EG_PER_THREAD = EG_BUF / BLOCK;
constexpr int EG1_BUF = 8192, EG2_BUF = 16384;              // rows of a group k_emit_groups_large holds in LDS (S-32G: 349 M special rows sit in groups of 4-16 K rows, 8-63 members)
__device__ __forceinline__ uint32_t group_members(const EmitArgs &a, const uint4 &S) { const uint32_t k = S.w & 0xFFFFFFu; return k == 0xFFFFFFu ? a.gk[S.z] : k; }
// what a batch needs of a special slot, gathered once per build (the chain elist -> sinfo -> s_pc / EB costs a workgroup three
// dependent memory round trips per batch otherwise: k_emit_groups is bound by such chains, not by bytes)
template <typename EBT> __global__ __launch_bounds__(BLOCK) void k_special_pack(EmitArgs a, uint32_t nsp, uint4 *cinfo, unsigned long long *cgb)
{
    const uint64_t j = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (j >= nsp) return;
    const uint32_t i = a.elist[j]; const uint4 S = a.sinfo[i];
    const uint32_t fl = S.w >> 24; const bool multi = (fl & SF_MULTI) != 0;
    cinfo[j] = make_uint4(S.x, multi ? group_members(a, S) : 1u, multi ? i - S.z : 0u, (uint32_t)a.s_pc[i] | (fl << 8));
    cgb[j] = (unsigned long long)reinterpret_cast<const EBT *>(a.EB)[multi ? S.z : i];
}
// head of the first group that starts at or behind enumeration row v, searched from slot `lo` on (a slot at or in front of it)
template <typename EBT> __device__ __forceinline__ uint64_t group_start_at_or_after(const EBT *ENB, const uint4 *cinfo, uint64_t nsp, uint64_t lo, uint64_t v)
{
    while (lo < nsp && (uint64_t)ENB[lo] < v) ++lo;
    if (lo >= nsp) return nsp;
    const uint4 ci = cinfo[lo];
    if (!((ci.w >> 8) & SF_MULTI)) return lo;
    const uint64_t head = lo - ci.z;
    return (uint64_t)ENB[head] < v ? head + ci.y : head;
}
template <typename EBT> __global__ __launch_bounds__(BLOCK) void k_tile_own(const EBT *ENB, const uint4 *cinfo, const uint32_t *etile_slot, uint64_t nsp, uint64_t ntiles, uint32_t *town)
{
    const uint64_t t = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t > ntiles) return;
    town[t] = (uint32_t)group_start_at_or_after<EBT>(ENB, cinfo, nsp, etile_slot[t], t * EMIT_TILE);      // every special slot has rows: the walk is one or two steps
}
template <typename EBT> __global__ __launch_bounds__(BLOCK) void k_emit_groups(EmitArgs a, uint8_t *bwt, uint32_t *qrow)
{
    __shared__ uint32_t LQ[EG_BUF];             // parse rows of the batch, in enumeration order (= list by list)
    __shared__ uint32_t s_eb[EG_SLOTS + 1], s_fb[EG_SLOTS], s_kf[EG_SLOTS], s_dl[EG_SLOTS];
    __shared__ uint64_t s_ob[EG_SLOTS];         // output row of the first row of the slot's group, minus that row's place in the batch
    __shared__ uint8_t s_pc[EG_SLOTS];
    __shared__ uint32_t red[4];
    __shared__ uint64_t jb[2];
    const EBT *ENB = reinterpret_cast<const EBT *>(a.ENB);
    const uint64_t tile0 = a.e0 / EMIT_TILE, tb = tile0 + blockIdx.x;
    const uint64_t o0 = tb * EMIT_TILE > a.e0 ? tb * EMIT_TILE : a.e0;
    const uint64_t o1 = ((tb + 1) * EMIT_TILE < a.e1) ? (tb + 1) * EMIT_TILE : a.e1;
    if (threadIdx.x < 2) {      // head of the first group that starts at or behind row v (v = o0: first owned group, v = o1: end of the owned groups)
        const uint64_t v = threadIdx.x ? o1 : o0;
        const uint64_t tv = threadIdx.x ? tb + 1 : tb;
        uint64_t j;
        if (v == tv * EMIT_TILE) j = a.town[tv];
        else {                          // first / last tile of a window that does not start / end on a tile boundary
            uint64_t lo = a.etile_slot[tb], hi = (uint64_t)a.etile_slot[tb + 1] + 1u;
            if (hi > a.ecount) hi = a.ecount;
            if (lo > hi) lo = hi;
            while (lo < hi) { const uint64_t mid = lo + ((hi - lo) >> 1); if ((uint64_t)ENB[mid] < v) lo = mid + 1; else hi = mid; }
            j = group_start_at_or_after<EBT>(ENB, a.cinfo, a.ecount, lo, v);
        }
        jb[threadIdx.x] = j;
    }
    __syncthreads();
    uint64_t j = jb[0]; const uint64_t jE = jb[1];
    const uint32_t cap = a.group_rows_cap < (uint32_t)EG_BUF ? a.group_rows_cap : (uint32_t)EG_BUF;
    while (j < jE) {                                            // uniform: one batch of whole groups per turn
        const uint32_t nload = jE - j < (uint64_t)EG_SLOTS ? (uint32_t)(jE - j) : (uint32_t)EG_SLOTS;
        const uint64_t B0 = (uint64_t)ENB[j];
        uint32_t cand[2] = {0u, 0u}, bad = 0xFFFFFFFFu;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const uint32_t s = threadIdx.x + (uint32_t)t * BLOCK;
            if (s < nload) {
                const uint4 ci = a.cinfo[j + s];
                const uint32_t fl = ci.w >> 8, dl = ci.z, k = ci.y;
                const uint64_t e0 = (uint64_t)ENB[j + s] - B0, e1 = (uint64_t)ENB[j + s + 1] - B0;
                s_eb[s] = e0 < 0xFFFFFFFFull ? (uint32_t)e0 : 0xFFFFFFFFu;
                if (s + 1 == nload) s_eb[nload] = e1 < 0xFFFFFFFFull ? (uint32_t)e1 : 0xFFFFFFFFu;
                s_fb[s] = ci.x; s_kf[s] = (k < 0xFFFFFFu ? k : 0xFFFFFFu) | (fl << 24); s_dl[s] = dl; s_pc[s] = (uint8_t)ci.w;
                s_ob[s] = (uint64_t)a.cgb[j + s];
                if (((fl & (SF_GFULL | SF_BIG)) || k > a.rank_members_max) && s < bad) bad = s;
                if (dl + 1u == k && e1 <= (uint64_t)cap) cand[t] = s + 1u;       // a batch may end behind this slot
            }
        }
        uint32_t tot;
        (void)block_incl_max(~bad, red, &tot);
        const uint32_t firstbad = ~tot;                        // first slot of a group k_emit keeps (all members of such a group carry the flag)
        uint32_t c = 0;
#pragma unroll
        for (int t = 0; t < 2; ++t) if (cand[t] && cand[t] <= firstbad && cand[t] > c) c = cand[t];
        uint32_t cut;
        (void)block_incl_max(c, red, &cut);
        if (cut == 0) {      // the group at j is left to k_emit
            const uint32_t k0 = a.cinfo[j].y;
            const uint64_t r0 = B0, r1 = (uint64_t)ENB[j + k0];
            {   // a group of ordinary members that is only too long for a batch: one workgroup of k_emit_groups_large takes it
                const uint32_t f0 = s_kf[0] >> 24;
                if (threadIdx.x == 0) {
                    unsigned long long idx = ~0ULL;
                    if (a.lglist && (f0 & SF_MULTI) && !(f0 & (SF_GFULL | SF_BIG)) && k0 <= (uint32_t)EG_SLOTS && r1 - r0 <= (uint64_t)EG2_BUF) {
                        const int big = r1 - r0 > (uint64_t)EG1_BUF ? 1 : 0;      // two lists: a workgroup that holds 8 K rows leaves room for a second one on its CU
                        idx = atomicAdd(a.lgcount + big, 1ULL);
                        if (idx < a.lgcap) a.lglist[(uint64_t)big * a.lgcap + idx] = (uint32_t)j;
                    }
                    jb[0] = idx < a.lgcap ? 1 : 0;
                }
                __syncthreads();
                const bool taken = jb[0] != 0;
                __syncthreads();
                if (taken) { j += k0; continue; }
            }
            if (threadIdx.x == 0) a.gleft[j] = 1;
            if (a.gstat && threadIdx.x == 0) {
                const uint32_t f0 = s_kf[0] >> 24;
                atomicAdd(&a.gstat[(f0 & SF_GFULL) ? 0 : (f0 & SF_BIG) ? 1 : (r1 - r0 > (uint64_t)cap) ? 2 : 3], (unsigned long long)(r1 - r0));
                int b = 0; while (b < 7 && (1024ull << (2 * b)) <= r1 - r0) ++b;
                atomicAdd(&a.gstat[4 + b], (unsigned long long)(r1 - r0));
            }
            if (r1 > r0) for (uint64_t t = r0 / EMIT_TILE + threadIdx.x; t <= (r1 - 1) / EMIT_TILE; t += BLOCK) a.tile_left[t - tile0] = 1;
            j += k0;
            __syncthreads();
            continue;
        }
        __syncthreads();
        const uint32_t nrows = s_eb[cut];
#pragma unroll
        for (int t = 0; t < 2; ++t) { const uint32_t s = threadIdx.x + (uint32_t)t * BLOCK; if (s < cut) s_ob[s] -= (uint64_t)s_eb[s - s_dl[s]]; }
        // the lists, coalesced (consecutive rows of a slot are consecutive list entries), then the bytes of the whole-word rows
        uint32_t qv[EG_PER_THREAD], sv[EG_PER_THREAD]; uint8_t cv[EG_PER_THREAD];
#pragma unroll
        for (int it = 0; it < EG_PER_THREAD; ++it) {
            const uint32_t r = threadIdx.x + (uint32_t)it * BLOCK;
            sv[it] = 0; qv[it] = 0; cv[it] = 0;
            if (r < nrows) {
                const uint32_t s = upper_bound_t<uint32_t>(s_eb, cut, r) - 1u, at = s_fb[s] + (r - s_eb[s]);
                sv[it] = s; qv[it] = a.ilist[at];
                if (!((s_kf[s] >> 24) & SF_MULTI) && a.bwl_il) cv[it] = a.bwl_il[at];
            }
        }
#pragma unroll
        for (int it = 0; it < EG_PER_THREAD; ++it) {
            const uint32_t r = threadIdx.x + (uint32_t)it * BLOCK;
            if (r < nrows) { LQ[r] = qv[it]; const uint32_t fl = s_kf[sv[it]] >> 24; if (fl & SF_MULTI) cv[it] = s_pc[sv[it]]; else if (!a.bwl_il) cv[it] = a.bwlast[qv[it]]; }
        }
        __syncthreads();
        // Place inside the group: own index + entries of the other members' lists in front of q (pfbwt.hpp:137-181), by bisection
        // in LDS.  Measured on S-32G (tools/emit_bench.py; 12 of the kernel's 18 ms are this ranking): 2, 4 or 8 rows of a thread
        // ranked in lock step are slower (20 / 23 / 33 ms: the rows of a thread sit in different groups, every row then waits for
        // the largest group), consecutive rows per thread with the other lists walked or galloped instead of bisected are much
        // slower (64-110 ms: the slowest lane of a wave sets the pace of every step).
#pragma unroll
        for (int it = 0; it < EG_PER_THREAD; ++it) {
            const uint32_t r = threadIdx.x + (uint32_t)it * BLOCK;
            if (r < nrows) {
                const uint32_t s = sv[it], kf = s_kf[s];
                uint32_t pos = r;
                if ((kf >> 24) & SF_MULTI) {
                    const uint32_t sh = s - s_dl[s], k = kf & 0xFFFFFFu, q = qv[it];
                    uint32_t before = 0;
                    for (uint32_t mm = 0; mm < k; ++mm) {
                        const uint32_t sm = sh + mm;
                        if (sm != s) before += lower_bound_u32(LQ + s_eb[sm], s_eb[sm + 1] - s_eb[sm], q);
                    }
                    pos = s_eb[sh] + before + (r - s_eb[s]);
                }
                // stored straight to their places: a whole-word slot's rows are consecutive, a group's rows stay inside its own stretch
                if (qrow) qrow[B0 - a.q0 + pos] = qv[it];
                const uint64_t o = s_ob[s] + pos;
                if (o >= a.w0 && o < a.w1) bwt[o - a.w0] = cv[it];
            }
        }
        __syncthreads();
        j += cut;
    }
}

// The groups k_emit_groups found too long for a batch (lglist): one group per workgroup turn.  Ranking every row in every other
// member's list costs members x log(list) LDS reads per row (measured on S-32G, groups of 4-16 K rows with 8-63 members: 66 ms,
// twice what k_emit needs in memory); merging the members' lists is SORTING the group's parse rows, so the (parse row, member)
// pairs are radix-sorted in LDS -- the wave-ballot LSD passes of the class sort (sufsort.h), 8-bit digits, as many passes as the
// parse has row bits -- and the sorted order IS the output order: bytes and parse rows leave coalesced.
template <int ITEMS> __device__ __forceinline__ void lds_sort_pairs_u32(uint32_t *keys, uint16_t *vals, uint32_t n, int npass, uint32_t (*wh)[RS_RADIX], uint32_t *red)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t nit = (n + BLOCK - 1) / BLOCK;                  // <= ITEMS; wave w owns the contiguous pairs [w * nit * 64, (w + 1) * nit * 64)
    for (uint32_t j = n + threadIdx.x; j < nit * BLOCK; j += BLOCK) { keys[j] = 0xFFFFFFFFu; vals[j] = 0; }      // padding stays behind the real pairs (stable)
    const uint32_t base = (uint32_t)wave * (nit * WAVE) + lane;
    for (int p = 0; p < npass; ++p) {
        const int sh = 8 * p;
        uint32_t k[ITEMS], dg[ITEMS]; uint16_t v[ITEMS];
#pragma unroll
        for (int w = 0; w < BLOCK / WAVE; ++w) wh[w][threadIdx.x] = 0;
        __syncthreads();
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            if ((uint32_t)it < nit) {
                const uint32_t i = base + (uint32_t)it * WAVE;
                k[it] = keys[i]; v[it] = vals[i];
                const uint32_t d = (k[it] >> sh) & 255u;
                uint32_t plo = 0xFFFFFFFFu, phi = 0xFFFFFFFFu;
                same_digit_lanes(d, plo, phi);
                const int leader = plo ? __builtin_ctz(plo) : 32 + __builtin_ctz(phi);
                uint32_t old = 0;
                if (lane == leader) { old = wh[wave][d]; wh[wave][d] = old + (uint32_t)__builtin_popcount(plo) + (uint32_t)__builtin_popcount(phi); }
                old = __shfl(old, leader);
                dg[it] = d | ((old + __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u))) << 8);
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
            if ((uint32_t)it < nit) { const uint32_t li = wh[wave][dg[it] & 255u] + (dg[it] >> 8); keys[li] = k[it]; vals[li] = v[it]; }
        }
        __syncthreads();
    }
}
template <typename EBT, int BUF> __global__ __launch_bounds__(BLOCK) void k_emit_groups_large(EmitArgs a, uint8_t *bwt, uint32_t *qrow)
{
    __shared__ uint32_t LQ[BUF];
    __shared__ uint16_t LS[BUF];
    __shared__ uint32_t wh[BLOCK / WAVE][RS_RADIX];
    __shared__ uint32_t s_eb[EG_SLOTS + 1], s_fb[EG_SLOTS];
    __shared__ uint8_t s_pc[EG_SLOTS];
    __shared__ uint32_t red[4];
    const EBT *ENB = reinterpret_cast<const EBT *>(a.ENB);
    constexpr int WHICH = BUF > EG1_BUF ? 1 : 0;
    const unsigned long long have = a.lgcount[WHICH];
    const uint64_t cnt = have < a.lgcap ? have : a.lgcap;
    for (uint64_t g = blockIdx.x; g < cnt; g += gridDim.x) {      // uniform
        const uint64_t j = a.lglist[(uint64_t)WHICH * a.lgcap + g];
        const uint32_t k = a.cinfo[j].y;                           // <= EG_SLOTS, rows <= BUF (k_emit_groups checked)
        const uint64_t B0 = (uint64_t)ENB[j], outbase = (uint64_t)a.cgb[j];
        for (uint32_t s = threadIdx.x; s < k; s += BLOCK) {
            const uint4 ci = a.cinfo[j + s];
            s_eb[s] = (uint32_t)((uint64_t)ENB[j + s] - B0); s_fb[s] = ci.x; s_pc[s] = (uint8_t)ci.w;
            if (s + 1 == k) s_eb[k] = (uint32_t)((uint64_t)ENB[j + k] - B0);
        }
        __syncthreads();
        const uint32_t nrows = s_eb[k];
        for (uint32_t r = threadIdx.x; r < nrows; r += BLOCK) { const uint32_t s = upper_bound_t<uint32_t>(s_eb, k, r) - 1u; LQ[r] = a.ilist[s_fb[s] + (r - s_eb[s])]; LS[r] = (uint16_t)s; }
        __syncthreads();
        lds_sort_pairs_u32<BUF / BLOCK>(LQ, LS, nrows, a.qpasses, wh, red);      // distinct words' occurrence lists share no parse row: no ties
        for (uint32_t r = threadIdx.x; r < nrows; r += BLOCK) {
            if (qrow) qrow[B0 - a.q0 + r] = LQ[r];
            const uint64_t o = outbase + r;
            if (o >= a.w0 && o < a.w1) bwt[o - a.w0] = s_pc[LS[r]];
        }
        __syncthreads();
    }
}

// Windows of output rows (multi-GPU slices, chunks of a huge text): a window owns output positions [lo, hi); rows of
// a group of equal suffixes that straddles a window boundary are enumerated for both neighbours, each keeps what lands
// in its window.  out[0] = first enumeration row (start of the group containing row lo), out[1] = end of the group
// containing row hi-1.
template <typename EBT> __global__ __launch_bounds__(BLOCK) void k_slice_bounds(EmitArgs a, uint64_t lo, uint64_t hi, unsigned long long *out)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const EBT *EB = reinterpret_cast<const EBT *>(a.EB);
    const EBT *ENB = reinterpret_cast<const EBT *>(a.ENB);
    const uint32_t il = slot_of_row<EBT>(a, lo);
    const uint32_t gh = (a.s_fl[il] & SF_MULTI) ? a.s_g0[il] : il;
    out[0] = EB[gh];
    out[2] = a.special ? (unsigned long long)ENB[a.cpos[gh]] : 0ULL;      // the same range in the enumeration of the special rows
    if (hi >= a.nout) { out[1] = a.nout; out[3] = a.special ? (unsigned long long)ENB[a.ecount] : 0ULL; return; }
    uint32_t s = slot_of_row<EBT>(a, hi - 1);
    if (a.s_fl[s] & SF_MULTI) { const uint32_t g = a.s_g0[s]; while (s < a.dsize && (a.s_fl[s] & SF_MULTI) && a.s_g0[s] == g) ++s; }
    else ++s;
    {   // first slot at or behind s whose rows start at or behind hi: EB never decreases (slots that produce no rows -- dictionary
        // suffixes of length <= w come in clusters of millions -- repeat the next one's value), so this is a bisection, not a walk
        uint64_t lo_ = s, hi_ = a.dsize;
        while (lo_ < hi_) { const uint64_t mid = lo_ + ((hi_ - lo_) >> 1); if ((uint64_t)EB[mid] < hi) lo_ = mid + 1; else hi_ = mid; }
        s = (uint32_t)lo_;
    }
    out[1] = s < a.dsize ? (uint64_t)EB[s] : a.nout;
    out[3] = a.special ? (unsigned long long)ENB[s < a.dsize ? a.cpos[s] : a.ecount] : 0ULL;
}

// Run-aware emission, the bulk of the rows: a slot whose group is one run writes cnt copies of its preceding byte.  Output-
// stationary: a thread owns 16 consecutive rows (aligned in the global row numbering -> one 16-byte store).  The slots
// under a tile of 4096 rows come from tile_slot; those that produce rows (at most 4097: dictionary suffixes of length
// <= w have none and come in clusters of thousands) are compacted into LDS with their first row relative to the tile;
// a thread bisects that list for its first row and walks on from there.  The rows of special slots get the placeholder
// s_pc too and are overwritten by k_emit afterwards (same stream).  bwt points at row a.w0 and (bwt - a.w0) is 16-byte
// aligned (host).
// The 16-byte row stores of k_fill (32 GB per build on S-32G; nothing in the kernel reads them back).  PFP_FILL_NT=1 gives them the
// streaming (non-temporal) policy; measured against the plain store on one box: profiles/r04nt_*.
#ifndef PFP_FILL_NT
#define PFP_FILL_NT 0
#endif
__device__ __forceinline__ void fill_store16(uint8_t *dst, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3)
{
#if PFP_FILL_NT && !defined(PFBWT_EMU_HIP_RUNTIME_H)
    typedef uint32_t fill_v4u __attribute__((ext_vector_type(4)));
    fill_v4u v = {x0, x1, x2, x3};
    __builtin_nontemporal_store(v, reinterpret_cast<fill_v4u *>(dst));
#else
    *reinterpret_cast<uint4 *>(dst) = make_uint4(x0, x1, x2, x3);
#endif
}
constexpr int FILL_PER_THREAD = 16, FILL_SUB = BLOCK * FILL_PER_THREAD;      // 4096 rows = 2 emission tiles
constexpr int FILL_GROUPS = 4;                                                // groups of FILL_SUB rows that share one slot list (a super-tile)
constexpr uint32_t FILL_MAX_SUBS = 8;                                         // super-tiles per workgroup, at most
static_assert(FILL_SUB % EMIT_TILE == 0, "fill tiles are whole emission tiles");
template <typename EBT> __global__ __launch_bounds__(BLOCK) void k_fill(EmitArgs a, uint8_t *bwt, uint32_t subs_per_wg)
{
    // A workgroup walks `subs_per_wg` super-tiles of FILL_GROUPS x FILL_SUB rows.  One super-tile is one chain of dependent
    // accesses (first slot under it -> the slots' row counts, first rows and bytes -> compaction in LDS -> stores), and that
    // latency, not HBM, bounded the kernel when a chain ended in 4 KB of output (1.7 TB/s): the list of slots with rows is
    // built once for the whole super-tile (a pangenome has ~25 such slots under 4096 rows) and every thread then writes one
    // 16-row piece in each of the FILL_GROUPS groups (each store instruction of a wave is 1 KB contiguous).  A super-tile
    // with more than FILL_SUB slots with rows (non-repetitive text: about one slot per row) is done group by group.
    __shared__ uint32_t eb[FILL_SUB + 2];       // first row (relative to the base row, clamped to 0) of the k-th slot with rows
    __shared__ uint8_t pcs[FILL_SUB + 2];
    __shared__ uint32_t wcnt[BLOCK / WAVE];
    __shared__ uint32_t sts[FILL_MAX_SUBS * FILL_GROUPS + 1];
    const EBT *EB = reinterpret_cast<const EBT *>(a.EB);
    const EBT *cnt = reinterpret_cast<const EBT *>(a.cnt);
    const uint64_t ntiles = (a.nout + EMIT_TILE - 1) / EMIT_TILE;
    constexpr uint64_t TPS = FILL_SUB / EMIT_TILE;
    constexpr uint64_t SUPER = (uint64_t)FILL_GROUPS * FILL_SUB;
    const uint64_t st0 = a.w0 / SUPER + (uint64_t)blockIdx.x * subs_per_wg;       // first super-tile of the workgroup
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lt = lane ? (~0ULL >> (64 - lane)) : 0ULL;
    for (uint32_t k = threadIdx.x; k <= subs_per_wg * FILL_GROUPS; k += BLOCK) {   // slot under the first row of every group
        const uint64_t t = (st0 * FILL_GROUPS + k) * TPS;
        sts[k] = a.tile_slot[t < ntiles ? t : ntiles];
    }
    __syncthreads();
    // slots with rows among [i0, i1] -> eb / pcs (rows relative to `base`); returns how many, or ~0u when the list would not fit
    auto compact = [&](uint32_t i0, uint32_t i1, uint64_t base) -> uint32_t {
        const uint32_t ns = i1 - i0 + 1u;
        uint32_t nz = 0;                                      // slots with rows so far (the same in every thread)
        for (uint32_t k0 = 0; k0 < ns; k0 += BLOCK) {
            const uint32_t k = k0 + threadIdx.x;
            const EBT cn = k < ns ? cnt[i0 + k] : (EBT)0;
            const uint64_t e = k < ns ? (uint64_t)EB[i0 + k] : 0ULL;
            const uint8_t pc = k < ns ? a.s_pc[i0 + k] : (uint8_t)0;
            const bool has = cn != 0;
            const unsigned long long bal = __ballot(has);
            __syncthreads();                                  // wcnt (and, first round, eb / pcs of the previous list) are free
            if (lane == 0) wcnt[wave] = (uint32_t)__popcll(bal);
            __syncthreads();
            uint32_t pos = nz, tot = 0;
#pragma unroll
            for (int v = 0; v < BLOCK / WAVE; ++v) { const uint32_t cw = wcnt[v]; if (v < wave) pos += cw; tot += cw; }
            if (nz + tot > (uint32_t)FILL_SUB + 1u) return ~0u;        // uniform; FILL_SUB rows have at most FILL_SUB + 1 slots with rows around them
            if (has) { pos += (uint32_t)__popcll(bal & lt); eb[pos] = e > base ? (uint32_t)(e - base) : 0u; pcs[pos] = pc; }
            nz += tot;
        }
        if (threadIdx.x == 0) eb[nz] = 0xFFFFFFFFu;
        __syncthreads();
        return nz;
    };
    // the thread's 16 rows of the group that starts at row g0 (list in LDS: nz slots, rows relative to `base`)
    auto write_group = [&](uint64_t g0, uint64_t base, uint32_t nz) {
        const uint64_t o0 = g0 > a.w0 ? g0 : a.w0;
        const uint64_t o1 = g0 + FILL_SUB < a.w1 ? g0 + FILL_SUB : a.w1;
        const uint64_t ra = g0 + (uint64_t)threadIdx.x * FILL_PER_THREAD;
        const uint64_t lo = ra > o0 ? ra : o0, hi = ra + FILL_PER_THREAD < o1 ? ra + FILL_PER_THREAD : o1;
        if (lo >= hi) return;
        uint32_t wd[FILL_PER_THREAD / 4] = {0, 0, 0, 0};
        uint32_t rel = (uint32_t)(lo - base);
        uint32_t s = upper_bound_t<uint32_t>(eb, nz, rel) - 1u;      // last slot with rows that starts at or before rel (eb[0] == 0: the slot of the first row)
        uint32_t nxt = eb[s + 1], c = pcs[s];
        if (lo == ra && hi == ra + FILL_PER_THREAD && nxt >= rel + FILL_PER_THREAD) {
            wd[0] = wd[1] = wd[2] = wd[3] = c * 0x01010101u;  // the 16 rows lie inside one slot's rows (runs are ~180 rows long on a pangenome)
        } else {
#pragma unroll
            for (int j = 0; j < FILL_PER_THREAD; ++j) {
                const uint64_t o = ra + j;
                if (o >= lo && o < hi) {
                    while (rel >= nxt) { ++s; nxt = eb[s + 1]; c = pcs[s]; }      // every step moves at least one row on
                    wd[j >> 2] |= c << (8 * (j & 3));
                    ++rel;
                }
            }
        }
        uint8_t *dst = bwt + (ra - a.w0);                     // may point in front of the buffer when ra < w0: only rows in [lo, hi) are stored
        if (lo == ra && hi == ra + FILL_PER_THREAD) fill_store16(dst, wd[0], wd[1], wd[2], wd[3]);
        else for (uint64_t o = lo; o < hi; ++o) { const int j = (int)(o - ra); dst[j] = (uint8_t)(wd[j >> 2] >> (8 * (j & 3))); }
    };
    for (uint32_t ss = 0; ss < subs_per_wg; ++ss) {
        const uint64_t tstart = (st0 + ss) * SUPER;
        if (tstart >= a.w1) break;                            // uniform
        const uint64_t tend = tstart + SUPER < a.w1 ? tstart + SUPER : a.w1;
        const uint32_t ng = (uint32_t)((tend - tstart + FILL_SUB - 1) / FILL_SUB);      // groups with rows of the window
        const uint32_t *gs = sts + ss * FILL_GROUPS;
        uint32_t nz = compact(gs[0], gs[ng], tstart);
        if (nz != ~0u) {
            for (uint32_t g = 0; g < ng; ++g) write_group(tstart + (uint64_t)g * FILL_SUB, tstart, nz);
        } else {
            for (uint32_t g = 0; g < ng; ++g) {
                const uint64_t g0 = tstart + (uint64_t)g * FILL_SUB;
                nz = compact(gs[g], gs[g + 1], g0);           // at most FILL_SUB slots have rows among FILL_SUB rows
                write_group(g0, g0, nz);
            }
        }
    }
}

// run count only (no samples wanted): workgroup reduction + one atomic per workgroup.  `bwt` points at the first
// row to count; has_prev says whether bwt[-1] holds the row in front of it (slices > 0).
__global__ __launch_bounds__(BLOCK) void k_run_count(const uint8_t *bwt, uint64_t nout, int has_prev, unsigned long long *runs)
{
    __shared__ uint32_t red[4];
    const uint64_t o0 = ((uint64_t)blockIdx.x * BLOCK + threadIdx.x) * 16;
    uint32_t cntr = 0;
    if (o0 < nout) {
        uint8_t prev = (o0 || has_prev) ? *(bwt + o0 - 1) : (uint8_t)0;
        for (int k = 0; k < 16 && o0 + k < nout; ++k) { const uint8_t cc = bwt[o0 + k]; cntr += cc != prev; prev = cc; }
    }
    uint32_t tot;
    (void)block_excl_sum(cntr, red, &tot);
    if (threadIdx.x == 0 && tot) atomicAdd(runs, (unsigned long long)tot);
}
// Run starts of a window, 16 rows per thread: bit k of the result <=> row j0 + k (< rows) differs from the row in front
// of it (pbwtc starts at 0, src/pfbwt-f.cpp:304).  Reads up to 15 bytes past the window (the BWT buffer is padded).
constexpr int RUN_PER_THREAD = 16, RUN_TILE = BLOCK * RUN_PER_THREAD;
__device__ __forceinline__ uint32_t run_mask16(const uint8_t *bwt, uint64_t j0, uint64_t rows, int has_prev)
{
    if (j0 >= rows) return 0;
    const uint64_t lo = ld8(bwt + j0), hi = ld8(bwt + j0 + 8);
    const uint64_t prev = (j0 || has_prev) ? (uint64_t)*(bwt + j0 - 1) : 0ULL;
    const uint64_t xl = lo ^ ((lo << 8) | prev), xh = hi ^ ((hi << 8) | (lo >> 56));
    uint32_t m = 0;
#pragma unroll
    for (int b = 0; b < 8; ++b) { m |= ((xl >> (8 * b)) & 0xff) ? (1u << b) : 0u; m |= ((xh >> (8 * b)) & 0xff) ? (1u << (8 + b)) : 0u; }
    const uint64_t left = rows - j0;
    return left >= 16 ? m : (m & ((1u << left) - 1u));
}
// run starts per tile; rmask (nullable) keeps every thread's 16-bit mask so that the sampling pass below does not read the
// BWT bytes again
__global__ __launch_bounds__(BLOCK) void k_run_tile_count(const uint8_t *bwt, uint64_t rows, int has_prev, uint32_t *tilecnt, uint16_t *rmask)
{
    __shared__ uint32_t red[4];
    const uint64_t g = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t m = run_mask16(bwt, g * RUN_PER_THREAD, rows, has_prev);
    if (rmask) rmask[g] = (uint16_t)m;
    uint32_t tot;
    (void)block_excl_sum((uint32_t)__popc(m), red, &tot);
    if (threadIdx.x == 0) tilecnt[blockIdx.x] = tot;
}
// .ssa / .esa, src/pfbwt-f.cpp:306-315 and :325-328, for a window of rows, in two steps.  Step 1: the ROWS of the pairs --
// row index = row_base + j, run index = run_base + tilebase[tile] + rank inside the tile; the run start at row o > 0 also
// ends the previous run at row o - 1; total_rows / total_runs_plus1 - 1 describe the whole output (the last row ends the
// last run; index -1 when no run starts in this slice: esa then points one pair past the slice's first entry).
// A workgroup takes SR_TILES consecutive tiles of RUN_TILE rows (the unit tilebase counts in): their masks are requested
// together and the prefix sums run as two scans of packed 16-bit fields (one tile per workgroup: 7.8 M workgroups of ~170
// stores each per 32 G rows).
constexpr int SR_TILES = 8;
template <typename SAT> __global__ __launch_bounds__(BLOCK) void k_sample_rows(const uint16_t *rmask, uint64_t rows, uint64_t ntiles, const uint32_t *tilebase, uint64_t row_base, uint64_t run_base, uint64_t total_rows,
                                                                               uint64_t total_runs_plus1 /*0: this window does not hold the last row*/, SAT *ssa, SAT *esa)
{
    __shared__ unsigned long long red[4];
    const uint64_t t0 = (uint64_t)blockIdx.x * SR_TILES;
    uint32_t m[SR_TILES]; unsigned long long c[2] = {0ULL, 0ULL};
#pragma unroll
    for (int b = 0; b < SR_TILES; ++b) {
        const uint64_t g = (t0 + b) * BLOCK + threadIdx.x;
        m[b] = (t0 + b < ntiles && g * RUN_PER_THREAD < rows) ? (uint32_t)rmask[g] : 0u;
        c[b >> 2] |= (unsigned long long)__popc(m[b]) << (16 * (b & 3));      // a field holds at most 256 * 16 = 4096
    }
    unsigned long long tot;
    const unsigned long long e0 = block_excl_sum(c[0], red, &tot), e1 = block_excl_sum(c[1], red, &tot);
#pragma unroll
    for (int b = 0; b < SR_TILES; ++b) {
        if (t0 + b >= ntiles) break;
        const uint64_t j0 = ((t0 + b) * BLOCK + threadIdx.x) * RUN_PER_THREAD;
        uint32_t mm = m[b];
        uint64_t k = run_base + tilebase[t0 + b] + (uint32_t)(((b < 4 ? e0 : e1) >> (16 * (b & 3))) & 0xFFFFu);
        while (mm) {
            const int bit = __ffs((int)mm) - 1; mm &= mm - 1;
            const uint64_t o = row_base + j0 + bit;
            ssa[2 * k] = (SAT)o;
            if (o) esa[2 * (k - 1)] = (SAT)(o - 1);
            ++k;
        }
        if (total_runs_plus1 && j0 < rows && row_base + rows == total_rows && total_rows - 1 - row_base - j0 < RUN_PER_THREAD)
            *(esa + 2 * ((long long)total_runs_plus1 - 2)) = (SAT)(total_rows - 1);
    }
}
// Step 2: the SA VALUES of the sampled rows, one thread per run start of the window (its row and the row in front of it),
// so that the dependent gathers of all samples are in flight together (a tile of 4096 rows holds ~20 samples: inside the
// per-tile kernel of round 1 nine lanes in ten idled through four dependent loads).  sa_win != nullptr: a full SA exists,
// sa_win[o - w_first]; else the value is computed from the row's parse row q (qrow[o - w_first], or looked up -- run-aware
// emission): sa = bwsai[q] - suffix length of the slot over the row (pfbwt.hpp:87-89), row 0 := n (src/pfbwt-f.cpp:301).
template <typename SAT, typename EBT> __global__ __launch_bounds__(BLOCK) void k_sample_values(EmitArgs a, const SAT *sa_win, const uint32_t *qrow, uint64_t w_first, uint64_t rc, uint64_t run_base,
                                                                                               int has_last, uint64_t last_idx, SAT *ssa, SAT *esa)
{
    const EBT *EB = reinterpret_cast<const EBT *>(a.EB);
    auto value = [&](uint64_t o) -> SAT {
        if (sa_win) return sa_win[o - w_first];
        if (o == 0) return (SAT)a.n;
        const uint32_t slot = slot_of_row<EBT>(a, o);
        uint32_t q;
        if (!a.special) q = qrow[o - w_first];
        else {   // run-aware emission: the parse row of a sampled row is looked up, not stored per row
            const uint4 S = a.sinfo[slot]; const uint32_t fl = S.w >> 24;
            if (slot_is_special(fl)) {
                const uint32_t g = (fl & SF_MULTI) ? S.z : slot;
                q = a.qspec[(uint64_t)reinterpret_cast<const EBT *>(a.ENB)[a.cpos[g]] - a.q0 + (o - (uint64_t)EB[g])];
            } else if (!(fl & SF_MULTI)) q = a.ilist[S.x + (uint32_t)(o - (uint64_t)EB[slot])];
            else q = (o == (uint64_t)EB[S.z]) ? a.gqf[S.z] : a.gql[S.z];          // a run of one byte: only its first and last row are ever sampled
        }
        return (SAT)((SAT)a.bwsai[q] - (SAT)a.s_sl[slot]);
    };
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < rc) {
        const uint64_t k = run_base + i;
        const uint64_t o = (uint64_t)ssa[2 * k];
        ssa[2 * k + 1] = value(o);
        if (o) esa[2 * (k - 1) + 1] = value(o - 1);
    } else if (i == rc && has_last) *(esa + 2 * (long long)last_idx + 1) = value(a.nout - 1);
}

} // namespace pfp
