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
__global__ __launch_bounds__(BLOCK) void k_pbwt_rows(const uint32_t *SAP, const uint32_t *P, const uint8_t *last, const uint32_t *sai, uint64_t m,
                                                     uint8_t *bwlast, uint32_t *bwsai, uint32_t *W, uint32_t *rowid)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i > m) return;
    const uint32_t s = SAP[i];
    rowid[i] = (uint32_t)i;
    if (s == 0) { bwlast[i] = 0; if (bwsai) bwsai[i] = 0; W[i] = 0; return; }
    bwlast[i] = (s == 1) ? last[m - 1] : last[s - 2];
    if (bwsai) bwsai[i] = sai[s - 1];
    W[i] = P[s - 1];
}

__global__ __launch_bounds__(BLOCK) void k_u32_add_store(const uint32_t *in, uint64_t n, uint32_t add, uint32_t *out)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) out[i] = in[i] + add;
}

// word ranks from the sorted dictionary suffixes: flag slots whose suffix starts a word
__global__ __launch_bounds__(BLOCK) void k_wordstart_flags(const uint32_t *SA, const uint32_t *wordid, const uint32_t *ws, uint32_t dwords, uint64_t dsize, uint32_t *flag)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= dsize) return;
    const uint32_t x = SA[i], id = wordid[x];
    flag[i] = (id < dwords && ws[id] == x) ? 1u : 0u;
}
__global__ __launch_bounds__(BLOCK) void k_word_rank(const uint32_t *SA, const uint32_t *wordid, const uint32_t *flag, const uint32_t *pos, uint64_t dsize,
                                                     const uint32_t *occw, uint32_t *wrank, uint32_t *idofrank, uint32_t *occ)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= dsize || !flag[i]) return;
    const uint32_t id = wordid[SA[i]], r = pos[i];
    wrank[id] = r; idofrank[r] = id; occ[r] = occw[id];
}
__global__ __launch_bounds__(BLOCK) void k_parse_ranks(const uint32_t *pid, const uint32_t *wrank, uint64_t m, uint32_t *parse)
{
    const uint64_t j = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (j < m) parse[j] = wrank[pid[j]] + 1u;    // generate_ranks, pfparser.hpp:504-517 (1-based)
}
__global__ __launch_bounds__(BLOCK) void k_sorted_lengths(const uint32_t *ws, const uint32_t *idofrank, uint64_t dwords, uint32_t *len1, uint32_t *srcstart)
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
    const uint32_t *SA, *grank, *wordid, *ws, *wrank /*nullable*/, *occ, *F, *ilist, *bwsai /*nullable*/;
    const uint8_t *bwlast;
    const uint32_t *EB; // exclusive scan of cnt over slots
    uint64_t nout, n;
};

__device__ __forceinline__ uint32_t word_rank_of(const EmitArgs &a, uint32_t id) { return a.wrank ? a.wrank[id] : id; }

// cnt[i] = number of text rows produced by slot i (0 for suffixes no longer than w, pfbwt.hpp:114)
__global__ __launch_bounds__(BLOCK) void k_emit_count(EmitArgs a, uint32_t *cnt)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= a.dsize) return;
    const uint32_t x = a.SA[i], id = a.wordid[x];
    uint32_t c = 0;
    if (id < a.dwords) {
        const uint32_t sl = a.ws[id + 1] - 1u - x;
        if (sl > (uint32_t)a.w) c = a.occ[word_rank_of(a, id)];
    }
    cnt[i] = c;
}

__device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t *a, uint32_t n, uint32_t x)
{   // number of entries < x
    uint32_t lo = 0, hi = n;
    while (lo < hi) { uint32_t mid = lo + ((hi - lo) >> 1); if (a[mid] < x) lo = mid + 1; else hi = mid; }
    return lo;
}

template <typename SAT> __global__ __launch_bounds__(BLOCK) void k_emit(EmitArgs a, uint8_t *bwt, SAT *sa)
{
    const uint64_t o = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (o >= a.nout) return;
    // slot: last i with EB[i] <= o
    const uint32_t i = upper_bound_u32(a.EB, (uint32_t)a.dsize, (uint32_t)o) - 1u;
    const uint32_t x = a.SA[i], id = a.wordid[x];
    const uint32_t sl = a.ws[id + 1] - 1u - x;                 // suff_len, pfbwt.hpp:83-85
    const uint32_t rk = word_rank_of(a, id);
    const uint32_t r = (uint32_t)o - a.EB[i];
    const uint32_t q = a.ilist[a.F[rk] + r];                   // parse-BWT row of this occurrence
    const uint32_t g0 = a.grank[x];                            // first slot of the group of equal suffixes
    uint64_t pos = o;
    const bool multi = (i != g0) || (i + 1 < a.dsize && a.grank[a.SA[i + 1]] == g0);
    const bool self_full = (x == a.ws[id]);
    bool full_emits_eow = false;
    if (multi) {                                               // pfbwt.hpp:137-181: merge by ilist position
        // gsacak puts byte-identical suffixes in dictionary-position order, i.e. by word rank.  The
        // reference's loop starts at the FIRST member: if that one is a whole word it is emitted alone
        // (:116-128) and the rest forms its own group; otherwise all members are merged by ilist
        // position and a whole-word member contributes dict[gsa-1] == EndOfWord as its BWT byte (:140).
        uint64_t before = 0;
        uint32_t first_rk = rk, first_before = 0, first_occ = a.occ[rk]; bool first_full = self_full;
        for (uint32_t s = g0; s < a.dsize; ++s) {
            const uint32_t xs = a.SA[s];
            if (a.grank[xs] != g0) break;
            if (s == i) continue;
            const uint32_t ids = a.wordid[xs];
            const uint32_t rs = word_rank_of(a, ids);
            const uint32_t oc = a.occ[rs];
            const uint32_t lb = lower_bound_u32(a.ilist + a.F[rs], oc, q);
            before += lb;
            if (rs < first_rk) { first_rk = rs; first_before = lb; first_occ = oc; first_full = (xs == a.ws[ids]); }
        }
        const uint64_t gb = a.EB[g0];
        if (first_full) pos = (first_rk == rk) ? gb + r : gb + first_occ + (before - first_before) + r;
        else { pos = gb + before + r; full_emits_eow = self_full; }
    }
    uint8_t c;
    if (self_full) c = full_emits_eow ? EndOfWord : a.bwlast[q];   // whole word, pfbwt.hpp:116-128
    else { c = a.D[x - 1]; if (c == Dollar && x - 1 == a.ws[id]) c = 0; }   // :132 "gsa[i]-1 ? dict[..] : 0"
    bwt[pos] = c;
    if (sa) {
        SAT v = (SAT)((SAT)(a.bwsai[q]) - (SAT)sl);             // UPDATE_SA, pfbwt.hpp:87-89
        if (pos == 0) v = (SAT)a.n;                             // src/pfbwt-f.cpp:301
        sa[pos] = v;
    }
}

// number of rows that sit in multi-word groups (the reference's "hard"/EASY2 bookkeeping, pfbwt.hpp:188)
__global__ __launch_bounds__(BLOCK) void k_multi_rows(EmitArgs a, const uint32_t *cnt, uint32_t *mr)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= a.dsize) return;
    const uint32_t x = a.SA[i], g0 = a.grank[x];
    const bool multi = (i != g0) || (i + 1 < a.dsize && a.grank[a.SA[i + 1]] == g0);
    mr[i] = multi ? cnt[i] : 0u;
}

// run starts: bwt[o] != bwt[o-1] (pbwtc starts at 0, src/pfbwt-f.cpp:304)
__global__ __launch_bounds__(BLOCK) void k_run_flags(const uint8_t *bwt, uint64_t nout, uint32_t *flag)
{
    const uint64_t o = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (o < nout) flag[o] = bwt[o] != (o ? bwt[o - 1] : (uint8_t)0) ? 1u : 0u;
}
// .ssa / .esa pairs, src/pfbwt-f.cpp:306-315 and :325-328
template <typename SAT> __global__ __launch_bounds__(BLOCK) void k_samples(const uint32_t *flag, const uint32_t *runidx, const SAT *sa, uint64_t nout, uint64_t runs,
                                                                           SAT *ssa, SAT *esa)
{
    const uint64_t o = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (o >= nout) return;
    if (flag[o]) {
        const uint64_t k = runidx[o];
        ssa[2 * k] = (SAT)o; ssa[2 * k + 1] = sa[o];
        if (o) { esa[2 * (k - 1)] = (SAT)(o - 1); esa[2 * (k - 1) + 1] = sa[o - 1]; }
    }
    if (o + 1 == nout && runs) { esa[2 * (runs - 1)] = (SAT)o; esa[2 * (runs - 1) + 1] = sa[o]; }
}

} // namespace pfp
