// pfbwt-f_amd/csrc/fasta.h -- FASTA ingest on the device ("GPU newline/header stripping"): the host uploads
// RAW file bytes, these kernels drop header lines, newlines and carriage returns and append the w 'A's behind every record --
// what PfParser::add_fasta does with kseq (include/pfparser.hpp:300-337, include/kseq.h:178-228) before the parse proper.
//
// Record semantics of kseq_read that are kept: a header line starts with '>' or '@' as the FIRST character of a line (:189,
// :204) and ends at its newline; sequence lines are concatenated without their line ends (a '\r' is dropped wherever it is,
// as the host reader of pfbwtf_common.hpp does); empty lines are skipped; an empty record still gets its pad.  Not taken over:
// FASTQ -- a line that starts with '+' switches kseq to skipping quality lines (:209-221); such a line is reported
// (PFP_E_ARG) and the caller falls back to the host reader.
//
// Line structure makes the byte classes a sequential state machine (in a header / in a sequence line / at a line start);
// per 64-byte piece the transition is a function on those three states, functions compose associatively, so:
//   k_fa_scan     per 16 KiB tile: the tile's transition function, kept bytes and header starts for each entry state
//   k_fa_spine    one workgroup: entry state, kept-byte offset and header count in front of every tile
//   k_fa_compact  per tile: the kept bytes (and the pads) go to their place in the text; tiles without a header start stage
//                 their bytes in LDS and store them 16 aligned bytes per thread
// Traffic: the raw bytes are read twice, the text is written once (3 B per base); the kernels run while the next raw chunk
// is crossing PCIe, so only the last chunk's share is on the critical path.
#pragma once
#include "prims.h"

namespace pfp {

constexpr int FA_SEG = 64;                         // bytes per thread
constexpr int FA_TILE = BLOCK * FA_SEG;            // 16 KiB
enum : uint32_t { FA_H = 0, FA_S = 1, FA_L = 2 };  // in a header line / in a sequence line / at a line start
constexpr uint32_t FA_IDENT = 0x24;                // f[s] = s, two bits per entry state

__device__ __forceinline__ uint32_t fa_apply(uint32_t f, uint32_t s) { return (f >> (2 * s)) & 3u; }
// first g, then f
__device__ __forceinline__ uint32_t fa_compose(uint32_t g, uint32_t f)
{
    return fa_apply(f, fa_apply(g, 0)) | (fa_apply(f, fa_apply(g, 1)) << 2) | (fa_apply(f, fa_apply(g, 2)) << 4);
}
__device__ __forceinline__ uint32_t fa_const(uint32_t s) { return s * 0x15u; }

// 4-bit mask of the bytes of x that equal the byte replicated in pat (exact: no borrow between bytes)
__device__ __forceinline__ uint32_t eq_mask4(uint32_t x, uint32_t pat)
{
    const uint32_t t = x ^ pat;
    const uint32_t z = ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu);      // 0x80 exactly in the zero bytes of t
    return (((z >> 7) * 0x01020408u) >> 24) & 0xFu;
}
__device__ __forceinline__ uint32_t fa_byte(const uint32_t (&wd)[16], uint32_t pos)
{
    const uint32_t idx = pos >> 2;
    uint32_t v = wd[0];
#pragma unroll
    for (uint32_t k = 1; k < 16; ++k) v = idx == k ? wd[k] : v;
    return (v >> (8 * (pos & 3u))) & 0xFFu;
}
__device__ __forceinline__ unsigned long long bits_below(uint32_t k) { return k >= 64 ? ~0ULL : ((1ULL << k) - 1ULL); }

// what a thread knows about its 64 bytes without knowing the state it is entered in
struct FaSeg {
    unsigned long long nl, cr, vm;      // newline bytes, carriage returns, valid bytes
    unsigned long long hdr;             // bytes of header lines that START inside the segment (their newline included)
    unsigned long long hs;              // those header lines' first bytes
    unsigned long long plus;            // line starts inside the segment that hold '+'
    uint32_t first;                     // position of the first newline (64: none)
    uint32_t b0;                        // byte 0
    uint32_t func;                      // transition function
};
__device__ __forceinline__ bool fa_is_header(uint32_t ch) { return ch == '>' || ch == '@'; }

__device__ __forceinline__ FaSeg fa_analyze(const uint8_t *raw, uint64_t len, uint64_t pos0, uint32_t (&wd)[16])
{
    FaSeg g;
    const uint32_t valid = pos0 >= len ? 0u : (len - pos0 >= (uint64_t)FA_SEG ? (uint32_t)FA_SEG : (uint32_t)(len - pos0));
    if (valid == (uint32_t)FA_SEG) {
        const uint4 *p = reinterpret_cast<const uint4 *>(raw + pos0);
#pragma unroll
        for (int j = 0; j < 4; ++j) { const uint4 q = p[j]; wd[4 * j] = q.x; wd[4 * j + 1] = q.y; wd[4 * j + 2] = q.z; wd[4 * j + 3] = q.w; }
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            uint32_t v = 0;
            for (uint32_t k = 0; k < 4; ++k) { const uint32_t i = 4u * (uint32_t)j + k; if (i < valid) v |= (uint32_t)raw[pos0 + i] << (8 * k); }
            wd[j] = v;
        }
    }
    g.vm = bits_below(valid);
    unsigned long long nl = 0, cr = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        nl |= (unsigned long long)eq_mask4(wd[j], 0x0A0A0A0Au) << (4 * j);
        cr |= (unsigned long long)eq_mask4(wd[j], 0x0D0D0D0Du) << (4 * j);
    }
    g.nl = nl & g.vm; g.cr = cr & g.vm;
    g.first = g.nl ? (uint32_t)__builtin_ctzll(g.nl) : 64u;
    g.b0 = wd[0] & 0xFFu;
    g.hdr = 0; g.hs = 0; g.plus = 0;
    // lines that start inside the segment
    unsigned long long ls = (g.nl << 1) & g.vm;
    uint32_t last_is_hdr = 0;
    while (ls) {
        const uint32_t p = (uint32_t)__builtin_ctzll(ls);
        ls &= ls - 1;
        const uint32_t ch = fa_byte(wd, p);
        last_is_hdr = fa_is_header(ch) ? 1u : 0u;
        if (last_is_hdr) {
            const unsigned long long rest = g.nl & ~bits_below(p);                     // the line's own newline (or none: open to the end)
            const uint32_t e = rest ? (uint32_t)__builtin_ctzll(rest) + 1u : 64u;
            g.hdr |= bits_below(e) & ~bits_below(p);
            g.hs |= 1ULL << p;
        } else if (ch == '+') g.plus |= 1ULL << p;
    }
    // transition: after a newline the entry state is forgotten
    if (valid == 0) g.func = FA_IDENT;
    else if ((g.nl >> (valid - 1)) & 1ULL) g.func = fa_const(FA_L);
    else if (g.nl) g.func = fa_const(last_is_hdr ? FA_H : FA_S);
    else g.func = FA_H | (FA_S << 2) | ((fa_is_header(g.b0) ? FA_H : FA_S) << 4);
    return g;
}
// header bytes / header starts / kept bytes of the segment once the entry state is known
__device__ __forceinline__ void fa_resolve(const FaSeg &g, uint32_t st, unsigned long long *keep, unsigned long long *hs, unsigned long long *plus)
{
    const bool valid = g.vm != 0;
    const bool head0 = valid && (st == FA_H || (st == FA_L && fa_is_header(g.b0)));
    unsigned long long hdr = g.hdr;
    if (head0) hdr |= bits_below(g.first < 64u ? g.first + 1u : 64u);
    *hs = g.hs | ((valid && st == FA_L && fa_is_header(g.b0)) ? 1ULL : 0ULL);
    *plus = g.plus | ((valid && st == FA_L && g.b0 == '+') ? 1ULL : 0ULL);
    *keep = ~hdr & ~g.nl & ~g.cr & g.vm;
}

// exclusive scan of the threads' transition functions; returns the function of everything in front of this thread, *total =
// the tile's.  lds: >= 4 entries
__device__ __forceinline__ uint32_t fa_block_excl(uint32_t f, uint32_t *lds, uint32_t *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = f;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(inc, d); if (lane >= d) inc = fa_compose(y, inc); }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    uint32_t base = FA_IDENT, tot = FA_IDENT;
#pragma unroll
    for (int i = 0; i < BLOCK / WAVE; ++i) { const uint32_t s = lds[i]; if (i < wave) base = fa_compose(base, s); tot = fa_compose(tot, s); }
    __syncthreads();
    *total = tot;
    uint32_t ex = __shfl_up(inc, 1);
    if (lane == 0) ex = FA_IDENT;
    return fa_compose(base, ex);
}

struct FaTiles { uint8_t *func; uint32_t *kept /*[3][T]*/, *hdr /*[3][T]*/; uint8_t *st; unsigned long long *kbase; uint32_t *hbase; uint64_t T; };

__global__ __launch_bounds__(BLOCK) void k_fa_scan(const uint8_t *raw, uint64_t len, FaTiles t)
{
    __shared__ uint32_t red[4];
    uint32_t wd[16];
    const uint64_t pos0 = (uint64_t)blockIdx.x * FA_TILE + (uint64_t)threadIdx.x * FA_SEG;
    const FaSeg g = fa_analyze(raw, len, pos0, wd);
    uint32_t tot;
    const uint32_t pre = fa_block_excl(g.func, red, &tot);
#pragma unroll
    for (uint32_t s = 0; s < 3; ++s) {
        unsigned long long keep, hs, plus;
        fa_resolve(g, fa_apply(pre, s), &keep, &hs, &plus);
        uint32_t kt, ht;
        (void)block_excl_sum((uint32_t)__builtin_popcountll(keep), red, &kt);
        (void)block_excl_sum((uint32_t)__builtin_popcountll(hs), red, &ht);
        if (threadIdx.x == 0) { t.kept[s * t.T + blockIdx.x] = kt; t.hdr[s * t.T + blockIdx.x] = ht; }
    }
    if (threadIdx.x == 0) t.func[blockIdx.x] = (uint8_t)tot;
}

// out[0] = kept bytes of the chunk, out[1] = header starts, out[2] = state behind the chunk
__global__ __launch_bounds__(BLOCK) void k_fa_spine(FaTiles t, uint32_t st0, unsigned long long *out)
{
    __shared__ uint32_t red[4];
    __shared__ unsigned long long red64[4];
    const uint64_t per = (t.T + BLOCK - 1) / BLOCK;
    const uint64_t k0 = (uint64_t)threadIdx.x * per, k1 = (k0 + per < t.T) ? k0 + per : t.T;
    uint32_t f = FA_IDENT;
    for (uint64_t k = k0; k < k1; ++k) f = fa_compose(f, t.func[k]);
    uint32_t tot;
    const uint32_t pre = fa_block_excl(f, red, &tot);
    uint32_t st = fa_apply(pre, st0);
    unsigned long long kept = 0, hdr = 0;
    for (uint64_t k = k0; k < k1; ++k) { t.st[k] = (uint8_t)st; kept += t.kept[st * t.T + k]; hdr += t.hdr[st * t.T + k]; st = fa_apply(t.func[k], st); }
    unsigned long long ktot, htot;
    unsigned long long kb = block_excl_sum(kept, red64, &ktot);
    unsigned long long hb = block_excl_sum(hdr, red64, &htot);
    for (uint64_t k = k0; k < k1; ++k) { const uint32_t s = t.st[k]; t.kbase[k] = kb; t.hbase[k] = (uint32_t)hb; kb += t.kept[s * t.T + k]; hb += t.hdr[s * t.T + k]; }
    if (threadIdx.x == 0) { out[0] = ktot; out[1] = htot; out[2] = fa_apply(tot, st0); }
}

// text + tbase = where the chunk's first kept byte goes if no pad were due; records0 = records started before the chunk.
// A record start that is not the first of the stream is preceded by the w 'A's of pfparser.hpp:335-337.
// rec_raw / rec_pos (nullable): per header start of the chunk its offset in the chunk and the text position of its record's
// first base.  flag[0] |= 1 when a line starts with '+'.
__global__ __launch_bounds__(BLOCK) void k_fa_compact(const uint8_t *raw, uint64_t len, FaTiles t, uint8_t *text, uint64_t tbase, uint64_t records0, int w,
                                                      uint64_t *rec_raw, uint64_t *rec_pos, uint32_t *flag)
{
    __shared__ uint32_t red[4];
    __shared__ __attribute__((aligned(16))) uint8_t sbuf[FA_TILE + 32];
    uint32_t wd[16];
    const uint64_t pos0 = (uint64_t)blockIdx.x * FA_TILE + (uint64_t)threadIdx.x * FA_SEG;
    const FaSeg g = fa_analyze(raw, len, pos0, wd);
    uint32_t tot;
    const uint32_t pre = fa_block_excl(g.func, red, &tot);
    unsigned long long keep, hs, plus;
    fa_resolve(g, fa_apply(pre, t.st[blockIdx.x]), &keep, &hs, &plus);
    if (plus) atomicOr(flag, 1u);
    uint32_t ktile, htile;
    const uint32_t kex = block_excl_sum((uint32_t)__builtin_popcountll(keep), red, &ktile);
    const uint32_t hex = block_excl_sum((uint32_t)__builtin_popcountll(hs), red, &htile);
    const uint64_t kb = t.kbase[blockIdx.x];
    const uint64_t recs_before_tile = records0 + t.hbase[blockIdx.x];
    if (htile == 0) {
        // no record starts in this tile: its kept bytes are one contiguous piece of the text
        const uint64_t pads = recs_before_tile ? recs_before_tile - 1 : 0;
        uint8_t *const out = text + tbase + kb + pads * (uint64_t)w;
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(out) & 15u);
        uint32_t o = mis + kex;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t kbits = (uint32_t)(keep >> (4 * j)) & 0xFu, v = wd[j];
            if (kbits == 0xFu) { sbuf[o] = (uint8_t)v; sbuf[o + 1] = (uint8_t)(v >> 8); sbuf[o + 2] = (uint8_t)(v >> 16); sbuf[o + 3] = (uint8_t)(v >> 24); o += 4; }
            else {
                if (kbits & 1u) sbuf[o++] = (uint8_t)v;
                if (kbits & 2u) sbuf[o++] = (uint8_t)(v >> 8);
                if (kbits & 4u) sbuf[o++] = (uint8_t)(v >> 16);
                if (kbits & 8u) sbuf[o++] = (uint8_t)(v >> 24);
            }
        }
        __syncthreads();
        const uint32_t end = mis + ktile;                                  // sbuf[mis, end) -> out[0, ktile); sbuf index i <-> address out - mis + i
        uint8_t *const abase = out - mis;
        for (uint32_t j = threadIdx.x; j * 16u < end; j += BLOCK) {
            const uint32_t a = j * 16u;
            if (a >= mis && a + 16u <= end) *reinterpret_cast<uint4 *>(abase + a) = *reinterpret_cast<const uint4 *>(sbuf + a);
            else for (uint32_t i = a < mis ? mis : a; i < a + 16u && i < end; ++i) abase[i] = sbuf[i];
        }
        return;
    }
    // a tile with record starts (rare): every thread writes its own bytes, pads where its header lines begin
    uint64_t recs = recs_before_tile + hex;                                // records started in front of this thread's bytes
    uint8_t *o = text + tbase + kb + kex + (recs ? recs - 1 : 0) * (uint64_t)w;
    for (uint32_t i = 0; i < (uint32_t)FA_SEG; ++i) {
        if ((hs >> i) & 1ULL) {
            if (recs) { for (int k = 0; k < w; ++k) o[k] = 'A'; o += w; }
            const uint64_t idx = t.hbase[blockIdx.x] + hex + (uint32_t)__builtin_popcountll(hs & bits_below(i));
            if (rec_raw) { rec_raw[idx] = pos0 + i; rec_pos[idx] = (uint64_t)(o - text); }
            ++recs;
        }
        if ((keep >> i) & 1ULL) *o++ = (uint8_t)fa_byte(wd, i);
    }
}

} // namespace pfp
