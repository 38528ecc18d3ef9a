// pfbwt-f_amd/csrc/parse.h -- stage 1 on the device: trigger scan, phrase ends, phrase
// fingerprints, de-duplication, dictionary build.
//
// Replaces (reference file:line): WangHash::update include/hash.hpp:29-35, wang_hash :12-21, the
// per-base loop of PfParser::add_fasta include/pfparser.hpp:335-352, process_phrase :595-601 and the
// std::map dictionary :69-70.  Text layout in HBM: `tb` = 16 guard bytes + X + w Dollars; Y = tb+15
// is the "decorated text" (Y[0] = Dollar, Y[1+i] = X[i], Y[n+1..n+w] = Dollar): phrase j is the byte
// range Y[ys_j .. ye_j] with ys_0 = 0, ys_j = ye_{j-1} - w + 1, so the Dollar of the first phrase and
// the w Dollars of the last one (pfparser.hpp:315-318, 484-489) need no special cases.
#pragma once
#include "prims.h"

namespace pfp {

typedef uint64_t tpos_t;   // text positions (see Spans below)

// include/hash.hpp:12-21
__host__ __device__ __forceinline__ uint64_t wang_hash(uint64_t key)
{
    key = (~key) + (key << 21);
    key = key ^ (key >> 24);
    key = (key + (key << 3)) + (key << 8);
    key = key ^ (key >> 14);
    key = (key + (key << 2)) + (key << 4);
    key = key ^ (key >> 28);
    key = key + (key << 31);
    return key;
}

// toupper + optional non-ACGT->A (pfparser.hpp:337-344); idempotent by construction
__device__ __forceinline__ uint32_t norm_base(uint32_t c, bool ntoa)
{
    if (c >= 'a' && c <= 'z') c -= 32;
    if (ntoa && !(c == 'A' || c == 'C' || c == 'G' || c == 'T')) c = 'A';
    return c;
}
// `count` records of `len` bytes, `stride` apart, packed one after the other with `w` 'A's behind each (the pad of
// pfparser.hpp:335-337).  One thread moves 16 bytes; destination rows start at arbitrary byte offsets.
__global__ __launch_bounds__(BLOCK) void k_feed_batch(const uint8_t *src, uint64_t count, uint64_t len, uint64_t stride, int w, uint8_t *dst)
{
    const uint64_t per_row = (len + 15) / 16;
    const uint32_t blocks_per_row = (uint32_t)((per_row + BLOCK - 1) / BLOCK);      // a workgroup stays inside one record:
    const uint64_t row = blockIdx.x / blocks_per_row;                                // one scalar division per workgroup
    const uint64_t j = (uint64_t)(blockIdx.x - (uint32_t)row * blocks_per_row) * BLOCK + threadIdx.x;
    if (row >= count || j >= per_row) return;
    const uint8_t *s = src + row * stride + 16 * j;
    uint8_t *d = dst + row * (len + (uint64_t)w) + 16 * j;
    if (16 * j + 16 <= len) { uint4 v; __builtin_memcpy(&v, s, 16); __builtin_memcpy(d, &v, 16); }
    else for (uint64_t k = 0; 16 * j + k < len; ++k) d[k] = s[k];
    if (j == per_row - 1) for (int k = 0; k < w; ++k) dst[row * (len + (uint64_t)w) + len + k] = 'A';
}

// the w 'A's behind each of `count` records of `len` bytes that were copied in by the runtime (rows `pitch` apart)
__global__ __launch_bounds__(BLOCK) void k_pad_rows(uint8_t *dst, uint64_t count, uint64_t len, uint64_t pitch, int w)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= count * (uint64_t)w) return;
    const uint64_t row = i / (uint64_t)w, k = i - row * (uint64_t)w;
    dst[row * pitch + len + k] = 'A';
}

// seq_nt4_ntoa_table, src/utils.c:139-161 (after toupper): A,N->0 C->1 G->2 T,'-'->3 else 5
__device__ __forceinline__ uint32_t ntoa_code(uint32_t c)
{
    return (c == 'A' || c == 'N') ? 0u : (c == 'C') ? 1u : (c == 'G') ? 2u : (c == 'T' || c == '-') ? 3u : 5u;
}

// 8-entry byte table looked up for four bytes at once: byte i of the result = byte (sel.byte[i] & 7) of {hi, lo} (v_perm_b32)
__device__ __forceinline__ uint32_t lut8x4(uint32_t hi, uint32_t lo, uint32_t sel)
{
    return __builtin_amdgcn_perm(hi, lo, sel);
}
// Four bases at once (norm_base + ntoa_code of every byte, 4-way SWAR; ~9 instead of ~35 instructions per base -- the
// scan was bound by this, not by the hash).  (c >> 1) & 7 is a perfect hash of the valid symbols, either case:
// A 0, C 1, T 2, G 3, '-' 6, N 7.  Returns the four 2-bit codes (first base in bits 7..6), *bad = invalid bytes (bit b =
// byte b), *out = normalised bytes.
__device__ __forceinline__ uint32_t pack4(uint32_t x, bool ntoa, uint32_t *bad, uint32_t *out)
{
    const uint32_t h4 = (x >> 1) & 0x07070707u;
    const uint32_t exp4 = lut8x4(0x4E2D0000u, 0x47544341u, h4);            // the upper-case symbol with that hash (0: none)
    const uint32_t L = (exp4 & 0x40404040u) >> 1;                            // 0x20 where a letter is expected: lower case is accepted
    const uint32_t allowed = (x ^ exp4) & ~L;
    uint32_t nz = (((allowed & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | allowed) & 0x80808080u;   // 0x80 in every byte that is not a valid symbol
    uint32_t c4 = lut8x4(0x00030000u, 0x02030100u, h4);                    // seq_nt4_ntoa_table: A,N 0  C 1  G 2  T,'-' 3
    if (ntoa) {   // pfparser.hpp:342-344: everything that is not ACGT (N and '-' too) becomes 'A'; nothing is invalid
        nz |= (h4 & 0x04040404u) << 5;
        const uint32_t m = (nz >> 7) * 0xFFu;
        *out = (exp4 & ~m) | (0x41414141u & m); c4 &= ~m; *bad = 0;
    } else {
        const uint32_t m = (nz >> 7) * 0xFFu;
        const uint32_t t = x & 0x7F7F7F7Fu;
        const uint32_t lower = (t + 0x1F1F1F1Fu) & ~(t + 0x05050505u) & ~x & 0x80808080u;   // bytes in 'a'..'z'
        *out = (exp4 & ~m) | ((x ^ (lower >> 2)) & m);                      // invalid bytes: toupper only (pfparser.hpp:337), code 0
        c4 &= ~m;
        *bad = (((nz >> 7) * 0x01020408u) >> 24) & 0xFu;
    }
    return (c4 * 0x40100401u) >> 24;
}
__device__ __forceinline__ uint32_t pack16(const uint4 &q, bool ntoa, uint32_t *bad /*bitmask of invalid bytes*/, uint4 *normed)
{
    const uint32_t wds[4] = {q.x, q.y, q.z, q.w};
    uint32_t out[4], pk = 0, badm = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t b4;
        pk = (pk << 8) | pack4(wds[i], ntoa, &b4, &out[i]);
        badm |= b4 << (4 * i);
    }
    *bad = badm;
    if (normed) *normed = make_uint4(out[0], out[1], out[2], out[3]);
    return pk;
}

// One thread = 16 consecutive bases (one 16-byte load).  Writes the normalised bytes back, one
// 16-bit trigger mask per thread and the trigger count of the workgroup.
// X must be 16-byte aligned with capacity rounded up to the grid; positions >= n are ignored.
// The trigger test h % p == 0 (pfparser.hpp:347) is done without a division: with p = 2^k * d (d odd) and
// dinv = d^-1 mod 2^64, h is a multiple of p exactly when rotr64(h * dinv, k) <= (2^64 - 1) / p.
struct DivTest { uint64_t dinv, limit; int k; };
inline DivTest make_divtest(uint64_t p)
{
    DivTest t; t.k = 0;
    uint64_t d = p;
    while (!(d & 1)) { d >>= 1; ++t.k; }
    uint64_t x = d;                                    // Newton: 3 correct bits -> 6 -> 12 -> 24 -> 48 -> 96
    for (int i = 0; i < 5; ++i) x *= 2 - d * x;
    t.dinv = x; t.limit = ~0ULL / p;
    return t;
}
__host__ __device__ __forceinline__ bool divisible(uint64_t h, const DivTest &t)
{
    const uint64_t v = h * t.dinv;
    return (t.k ? ((v >> t.k) | (v << (64 - t.k))) : v) <= t.limit;
}
__global__ __launch_bounds__(BLOCK) void k_trigger_scan(uint8_t *X, uint64_t n, int w, DivTest p, uint64_t kmask, int ntoa,
                                                        uint16_t *mask16, uint64_t *blockcnt, unsigned long long *err_pos)
{
    __shared__ uint32_t pk[BLOCK + 2];
    __shared__ uint32_t red[4];
    const uint64_t t = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    const uint64_t base = t * 16;
    uint4 q = make_uint4(0, 0, 0, 0), nq;
    uint32_t bad = 0;
    if (base < n) q = *reinterpret_cast<const uint4 *>(X + base);
    const uint32_t mine = pack16(q, ntoa != 0, &bad, &nq);
    if (base < n) {
        const uint32_t live = (n - base >= 16) ? 0xffffu : ((1u << (unsigned)(n - base)) - 1u);
        if (live != 0xffffu) { // keep bytes beyond n untouched
            uint32_t a[4] = {q.x, q.y, q.z, q.w}, b[4] = {nq.x, nq.y, nq.z, nq.w};
            for (int i = 0; i < 16; ++i) if (!((live >> i) & 1)) { b[i >> 2] = (b[i >> 2] & ~(0xffu << (8 * (i & 3)))) | (a[i >> 2] & (0xffu << (8 * (i & 3)))); }
            nq = make_uint4(b[0], b[1], b[2], b[3]);
        }
        *reinterpret_cast<uint4 *>(X + base) = nq;
        bad &= live;
        if (bad) atomicMin(err_pos, (unsigned long long)(base + (uint64_t)(__ffs((int)bad) - 1)));
    }
    pk[threadIdx.x + 2] = mine;
    if (threadIdx.x < 2) { // halo: the 32 bases in front of the workgroup's first base
        const uint64_t first = (uint64_t)blockIdx.x * BLOCK; // in units of 16 bases
        uint32_t hv = 0;
        if (first + threadIdx.x >= 2) {
            uint32_t hb;
            uint4 hq = *reinterpret_cast<const uint4 *>(X + (first + threadIdx.x - 2) * 16);
            hv = pack16(hq, ntoa != 0, &hb, nullptr);
        }
        pk[threadIdx.x] = hv;
    }
    __syncthreads();
    uint64_t kmer = ((uint64_t)pk[threadIdx.x] << 32) | pk[threadIdx.x + 1];
    uint32_t trig = 0;
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        kmer = (kmer << 2) | ((mine >> (30 - 2 * b)) & 3u);       // hash.hpp:32
        const uint64_t pos = base + b;
        const uint64_t h = wang_hash(kmer & kmask);
        // pfparser.hpp:347: pos_ > w  <=>  pos >= w (pos_ = pos + 1 at the test)
        if (pos < n && pos >= (uint64_t)w && divisible(h, p)) trig |= 1u << b;
    }
    mask16[t] = (uint16_t)trig;
    uint32_t tot;
    (void)block_excl_sum((uint32_t)__popc(trig), red, &tot);
    if (threadIdx.x == 0) blockcnt[blockIdx.x] = tot;   // 64-bit: the scan over workgroups yields phrase indices
}

// ---- the same scan for w <= 10 (the reference's default and the only value its pipeline driver uses, vcf_to_bwt.py:118-129):
// the k-mer has at most 20 bits, so "wang_hash(kmer) % p == 0" (hash.hpp:12-21 + pfparser.hpp:347) is precomputed for all
// 4^w k-mers into a bit table of <= 128 KiB that sits in LDS for the whole scan; a base then costs one LDS read instead of
// ~60 64-bit integer operations (the hash made the scan VALU-bound: 74 ms of 32 Gbase).  One workgroup of 1024 threads per
// CU (the table takes 128 of the 160 KiB of LDS), each walks `tiles_per_wg` tiles of 16 Kbase.  The normalised bytes are
// written back only where they differ from the input (upper-case ACGT input: no stores at all).
constexpr int TS_THREADS = 1024;
constexpr int TS_MAX_W = 10;
constexpr uint32_t TS_TAB_WORDS = 1u << (2 * TS_MAX_W - 5);
__global__ __launch_bounds__(BLOCK) void k_trigger_table(int w, DivTest p, uint32_t *tab)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;           // one 32-bit word of the table = 32 k-mers
    const uint32_t words = (1u << (2 * w)) >= 32u ? (1u << (2 * w)) / 32u : 1u;
    if (i >= words) return;
    uint32_t m = 0;
    for (uint32_t b = 0; b < 32; ++b) { const uint64_t km = (uint64_t)i * 32 + b; if (km < (1ULL << (2 * w)) && divisible(wang_hash(km), p)) m |= 1u << b; }
    tab[i] = m;
}
// VIEW: the text is not in X yet -- it is `count` rows of `len` bytes, `stride` apart, in the caller's device memory
// (pfp_parse_feed_device_view), each followed by the w 'A's of pfparser.hpp:335-337 that exist nowhere: the scan reads the rows where
// they are and WRITES the normalised text (pads included) to X, which the later stages read.  Until round 4 a copy kernel laid the rows
// out in X first (k_feed_batch: 32 GB read + 32 GB written, 11.4 ms on S-32G) and the scan read X again.
struct RowView { const uint8_t *src; uint64_t count, len, stride, rowlen /*len + w*/; };
// 16 text bytes that start at offset o of row r (o may run past the row: the pad, then the next rows)
__device__ __forceinline__ uint4 view_load16(const RowView &v, uint64_t r, uint64_t o)
{
    while (o >= v.rowlen) { o -= v.rowlen; ++r; }
    if (r < v.count && o + 16 <= v.len) { uint4 q; __builtin_memcpy(&q, v.src + r * v.stride + o, 16); return q; }
    uint32_t a[4] = {0u, 0u, 0u, 0u};
    for (int k = 0; k < 16; ++k) {
        uint64_t oo = o + (uint64_t)k, rr = r;
        while (oo >= v.rowlen) { oo -= v.rowlen; ++rr; }
        const uint32_t b = rr >= v.count ? 0u : (oo < v.len ? (uint32_t)v.src[rr * v.stride + oo] : (uint32_t)'A');
        a[k >> 2] |= b << (8 * (k & 3));
    }
    return make_uint4(a[0], a[1], a[2], a[3]);
}
template <bool VIEW> __global__ __launch_bounds__(TS_THREADS) void k_trigger_scan_tab(uint8_t *X, uint64_t n, int w, const uint32_t *tab, uint32_t tabwords, uint32_t kmask, int ntoa, uint32_t tiles_per_wg,
                                                                   uint64_t nthreads_total, uint16_t *mask16, uint64_t *blockcnt /*zeroed*/, unsigned long long *err_pos, RowView rv)
{
    __shared__ uint32_t stab[TS_TAB_WORDS];
    __shared__ uint32_t pk[2][TS_THREADS + 2];                     // packed bases of the tile, [0..1] = the 32 bases in front; two tiles alternate
    for (uint32_t i = threadIdx.x; i < tabwords; i += TS_THREADS) stab[i] = tab[i];
    const int lane = threadIdx.x & 63;
    const uint64_t tile0 = (uint64_t)blockIdx.x * tiles_per_wg;
    // software pipeline: the 16 bytes of the tiles k + 1 .. k + 3 are on their way while tile k is processed.  One workgroup per CU (the
    // table fills the LDS) means 16 waves of loads per CU: with ONE tile ahead the scan that also WRITES the text (row view) took 7.0 ms per
    // 8 Gbase, with three 5.7 (the read-only scan stays at 5.0: tools/view_bench.py).  Three NAMED registers: a rotated array made the
    // compiler wait for every load where it was issued -- no gain at any depth.
    uint64_t prow = 0, po = 0;                                     // VIEW: row and offset in it of the next tile to be requested (uniform)
    if (VIEW) { const uint64_t p0 = tile0 * TS_THREADS * 16; prow = p0 / rv.rowlen; po = p0 - prow * rv.rowlen; }
    // (a tile that lies inside one row -- all but one tile in two thousand on S-32G -- is a uniform test and one address per thread)
    auto view_tile = [&](uint64_t row, uint64_t o) -> uint4 {
        if (row < rv.count && o + (uint64_t)TS_THREADS * 16 <= rv.len) { uint4 q; __builtin_memcpy(&q, rv.src + row * rv.stride + o + 16u * threadIdx.x, 16); return q; }
        return view_load16(rv, row, o + 16u * threadIdx.x);
    };
    uint32_t requested = 0;                                        // tiles of this workgroup requested so far (in order)
    auto request = [&]() -> uint4 {
        uint4 r = make_uint4(0, 0, 0, 0);
        const uint64_t b = ((tile0 + requested) * TS_THREADS + threadIdx.x) * 16;
        if (requested < tiles_per_wg && b < n) r = VIEW ? view_tile(prow, po) : *reinterpret_cast<const uint4 *>(X + b);
        if (VIEW) { po += (uint64_t)TS_THREADS * 16; while (po >= rv.rowlen) { po -= rv.rowlen; ++prow; } }
        ++requested;
        return r;
    };
    uint4 qa = request(), qb = request(), qc = request();
    for (uint32_t tl = 0; tl < tiles_per_wg; ++tl) {
        const uint64_t first = (tile0 + tl) * TS_THREADS;          // in units of 16 bases
        if (first >= nthreads_total) break;                        // uniform
        const uint64_t t = first + threadIdx.x;
        const uint64_t base = t * 16;
        const uint4 q = qa;
        qa = qb; qb = qc; qc = request();
        uint4 nq; uint32_t bad = 0;
        const uint32_t mine = pack16(q, ntoa != 0, &bad, &nq);
        if (base < n) {
            const uint32_t live = (n - base >= 16) ? 0xffffu : ((1u << (unsigned)(n - base)) - 1u);
            if (live != 0xffffu) { // keep bytes beyond n untouched
                const uint4 keep = VIEW ? *reinterpret_cast<const uint4 *>(X + base) : q;
                uint32_t a[4] = {keep.x, keep.y, keep.z, keep.w}, b[4] = {nq.x, nq.y, nq.z, nq.w};
                for (int i = 0; i < 16; ++i) if (!((live >> i) & 1)) { b[i >> 2] = (b[i >> 2] & ~(0xffu << (8 * (i & 3)))) | (a[i >> 2] & (0xffu << (8 * (i & 3)))); }
                nq = make_uint4(b[0], b[1], b[2], b[3]);
            }
            if (VIEW || nq.x != q.x || nq.y != q.y || nq.z != q.z || nq.w != q.w) *reinterpret_cast<uint4 *>(X + base) = nq;
            bad &= live;
            if (bad) atomicMin(err_pos, (unsigned long long)(base + (uint64_t)(__ffs((int)bad) - 1)));
        }
        uint32_t *cur = pk[tl & 1];
        const uint32_t *oth = pk[(tl & 1) ^ 1];
        cur[threadIdx.x + 2] = mine;
        if (threadIdx.x < 2) { // halo: the 32 bases in front of the tile -- the tail of the previous tile of this workgroup, or from memory
            uint32_t hv = 0;
            if (tl) hv = oth[TS_THREADS + threadIdx.x];
            else if (first + threadIdx.x >= 2) {
                uint32_t hb;
                const uint64_t hp = (first + threadIdx.x - 2) * 16;      // (VIEW: the neighbouring workgroup may not have written X yet -- from the rows)
                uint4 hq = VIEW ? view_load16(rv, hp / rv.rowlen, hp % rv.rowlen) : *reinterpret_cast<const uint4 *>(X + hp);
                hv = pack16(hq, ntoa != 0, &hb, nullptr);
            }
            cur[threadIdx.x] = hv;
        }
        __syncthreads();       // the one barrier per tile (tile k + 2 reuses this buffer only after every wave has passed the barrier of tile k + 1)
        // the k-mer that ends at base b is a window of the 64-bit string {the 16 bases in front : this thread's 16 bases}
        // (hash.hpp:32): its table word and its bit come straight out of that pair by constant shifts (v_alignbit_b32) -- 6
        // VALU instructions per base where shifting the k-mer along base by base took 9-10, and the scan is VALU-bound
        const uint64_t both = ((uint64_t)cur[threadIdx.x + 1] << 32) | mine;
        const uint32_t amask = (kmask >> 5) << 2, bmask = kmask & 31u;        // byte offset of the table word; bit inside it
        const uint8_t *stab8 = reinterpret_cast<const uint8_t *>(stab);
        uint32_t trig = 0;
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            const uint32_t word = *reinterpret_cast<const uint32_t *>(stab8 + ((uint32_t)(both >> (33 - 2 * b)) & amask));
            const uint32_t bit = (uint32_t)(both >> (30 - 2 * b)) & bmask;
            trig |= ((word >> bit) & 1u) << b;
        }
        // pfparser.hpp:347: pos_ > w  <=>  pos >= w; nothing at or behind n
        if (base < (uint64_t)w) trig &= ~((1u << (unsigned)((uint64_t)w - base > 16 ? 16 : (uint64_t)w - base)) - 1u);
        if (base >= n) trig = 0; else if (n - base < 16) trig &= (1u << (unsigned)(n - base)) - 1u;
        if (t < nthreads_total) mask16[t] = (uint16_t)trig;
        // trigger count of every group of 256 threads (the unit k_phrase_ends works in): one atomic per wave
        uint32_t cnt = (uint32_t)__popc(trig);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d);
        if (lane == 0 && cnt && t < nthreads_total) atomicAdd(reinterpret_cast<unsigned long long *>(&blockcnt[t / BLOCK]), (unsigned long long)cnt);
    }
}

// ye[j] = (trigger position e_j) + 1 for every trigger, in text order.  A workgroup takes PE_BLOCKS consecutive groups of
// 256 mask words (a group = the unit blockoff counts in, ~40 triggers at p = 100): all its masks are requested up front and
// the 8 prefix sums run as two scans of packed 16-bit fields -- one group per workgroup was 7.8 M workgroups of a few
// hundred bytes each (6.1 ms on S-32G for 6.6 GB).
constexpr int PE_BLOCKS = 8;
__global__ __launch_bounds__(BLOCK) void k_phrase_ends(const uint16_t *mask16, const uint64_t *blockoff, uint64_t ngroups, tpos_t *ye)
{
    __shared__ unsigned long long red[4];
    const uint64_t g0 = (uint64_t)blockIdx.x * PE_BLOCKS;
    uint32_t m[PE_BLOCKS]; unsigned long long c[2] = {0ULL, 0ULL};
#pragma unroll
    for (int b = 0; b < PE_BLOCKS; ++b) {
        m[b] = g0 + b < ngroups ? mask16[(g0 + b) * BLOCK + threadIdx.x] : 0u;
        c[b >> 2] |= (unsigned long long)__popc(m[b]) << (16 * (b & 3));      // a field holds at most 256 * 16 = 4096
    }
    unsigned long long tot;
    const unsigned long long e0 = block_excl_sum(c[0], red, &tot), e1 = block_excl_sum(c[1], red, &tot);
#pragma unroll
    for (int b = 0; b < PE_BLOCKS; ++b) {
        if (g0 + b >= ngroups) break;
        uint32_t mm = m[b];
        uint64_t o = blockoff[g0 + b] + (uint32_t)(((b < 4 ? e0 : e1) >> (16 * (b & 3))) & 0xFFFFu);
        const uint64_t t = (g0 + b) * BLOCK + threadIdx.x;
        while (mm) { const int bit = __ffs((int)mm) - 1; mm &= mm - 1; ye[o++] = (tpos_t)(t * 16 + bit + 1); }
    }
}

// Where the byte strings to be de-duplicated live: string j = Y[ys_j .. ye[j]].  Text mode (ys == nullptr):
// consecutive phrases overlap by w bytes, ys_j = ye[j-1] - w + 1, ys_0 = 0.  Word mode (ys != nullptr): explicit
// starts (dictionary words of several shards laid out in one buffer, see pfp_merge_shards).
// Text positions are 64-bit (tpos_t): a 1000-haplotype collection has tens of Gbases while its dictionary and its
// parse stay far below 2^32 entries.
struct Spans { const tpos_t *ye; const uint32_t *ys32; const uint32_t *ye32; int w; };
__device__ __forceinline__ void phrase_span(const Spans &sp, uint64_t j, tpos_t *ys, uint32_t *len)
{
    if (sp.ys32) { const uint32_t s = sp.ys32[j]; *ys = s; *len = sp.ye32[j] - s + 1u; return; }
    const tpos_t s = j ? sp.ye[j - 1] - (tpos_t)sp.w + 1u : (tpos_t)0;
    *ys = s; *len = (uint32_t)(sp.ye[j] - s + 1u);
}

// unaligned 8-byte read (gfx950 global loads need no alignment; callers keep 7 readable bytes behind every string)
__device__ __forceinline__ uint64_t ld8(const uint8_t *p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }

constexpr uint32_t LONG_PHRASE = 2048; // phrases longer than this go to the workgroup-per-phrase kernels

// ---- exact de-duplication: a hash table of representatives ----------------------------------------------------------
// The reference keeps the phrases in a std::map<std::string, Freq> (pfparser.hpp:69-70, 595-597).  Here every phrase is
// looked up in an open-addressing table whose entries are {32 bits of the phrase's hash | index of a representative
// phrase}: an entry matches when the filter bits agree AND the bytes of the two phrases are equal (compared 8 at a time),
// so the result is exact whatever the hash does.  The phrases are visited in text order (their own bytes stream through
// once), the representatives they are compared with are the ~dictionary-sized hot set.  Entries never change once
// written, so the plain (possibly stale, per-XCD L2) read in front of the compare-and-swap is safe: a stale EMPTY is
// corrected by the value the CAS returns.  Round 1 sorted (fingerprint, phrase) pairs and compared neighbours instead:
// 8 radix passes over all phrases and two random phrase reads per phrase.
constexpr uint64_t HT_EMPTY = ~0ULL, HT_NOINFO = ~0ULL;
constexpr uint32_t HT_MAX_PROBES = 1u << 16;
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) { z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }
// content hash of a string of len bytes whose 8-byte little-endian words come from ld(byte offset): FOUR interleaved
// multiply-xorshift chains (word q feeds chain q & 3), the last word masked.  One chain is a dependent sequence of
// 64-bit multiplies as long as the phrase -- and a wave waits for the longest of its 64 phrases.
template <typename NEXT> __device__ __forceinline__ uint64_t hash_words(uint32_t len, uint64_t seed, NEXT next /* the string's words, in order */)
{
    const uint64_t M = 0xD6E8FEB86659FD93ULL;
    uint64_t h0 = seed ^ ((uint64_t)len * 0x9E3779B97F4A7C15ULL), h1 = h0 ^ 0xA0761D6478BD642FULL, h2 = h0 ^ 0xE7037ED1A0B428DBULL, h3 = h0 ^ 0x8EBC6AF09C88C6E3ULL;
    uint32_t i = 0;
    for (; i + 32 <= len; i += 32) {
        const uint64_t a = next(), b = next(), c = next(), d = next();
        h0 = (h0 ^ a) * M; h1 = (h1 ^ b) * M; h2 = (h2 ^ c) * M; h3 = (h3 ^ d) * M;
        h0 ^= h0 >> 32; h1 ^= h1 >> 32; h2 ^= h2 >> 32; h3 ^= h3 >> 32;
    }
    // the last 1..31 bytes: up to four more words, the final one masked
    if (i < len) { uint64_t a = next(); if (len - i < 8) a &= (1ULL << (8 * (len - i))) - 1ULL; h0 = (h0 ^ a) * M; h0 ^= h0 >> 32; }
    if (i + 8 < len) { uint64_t a = next(); if (len - i - 8 < 8) a &= (1ULL << (8 * (len - i - 8))) - 1ULL; h1 = (h1 ^ a) * M; h1 ^= h1 >> 32; }
    if (i + 16 < len) { uint64_t a = next(); if (len - i - 16 < 8) a &= (1ULL << (8 * (len - i - 16))) - 1ULL; h2 = (h2 ^ a) * M; h2 ^= h2 >> 32; }
    if (i + 24 < len) { uint64_t a = next(); a &= (1ULL << (8 * (len - i - 24))) - 1ULL; h3 = (h3 ^ a) * M; h3 ^= h3 >> 32; }
    return mix64(h0 ^ ((h1 << 17) | (h1 >> 47)) ^ ((h2 << 31) | (h2 >> 33)) ^ ((h3 << 47) | (h3 >> 17)));
}
__device__ __forceinline__ uint64_t str_hash(const uint8_t *s, uint32_t len, uint64_t seed) { uint32_t o = 0; return hash_words(len, seed, [s, &o]() { const uint64_t v = ld8(s + o); o += 8; return v; }); }
__device__ __forceinline__ bool str_equal(const uint8_t *a, const uint8_t *b, uint32_t len)
{
    uint32_t i = 0;
    for (; i + 8 <= len; i += 8) if (ld8(a + i) != ld8(b + i)) return false;
    if (i < len) return ((ld8(a + i) ^ ld8(b + i)) & ((1ULL << (8 * (len - i))) - 1ULL)) == 0;
    return true;
}
// One entry = one 32-byte piece of a line: what a lookup touches -- the filter | representative word, where the
// representative's bytes are, the occurrence counter -- arrives with ONE memory transaction (three arrays cost three
// random sectors per phrase, and this kernel is bound by exactly that traffic: 190 GB per 325 M phrases before).
// The table is cleared to all-ones: tab == HT_EMPTY, rinfo == HT_NOINFO (its creator has not stored it yet: read the
// spans), cnt == 2^32 - 1 (occurrences = cnt + 1, wrapping).
struct DedupEntry { unsigned long long tab, rinfo; uint32_t cnt, kidx, pad[2]; };
// slotof[j] (the entry of string j) holds the entry's DENSE index kidx (the order in which the entries were created:
// k_dedup_ids then gathers from an array of nd ids that stays in L2, not from the table), or, when the creator's store of
// it was not visible yet, HT_BYSLOT | slot.
constexpr uint32_t HT_NOIDX = ~0u, HT_BYSLOT = 0x80000000u;
__device__ __forceinline__ uint32_t entry_ref(const DedupEntry *ent, uint64_t slot) { const uint32_t k = ent[slot].kidx; return k != HT_NOIDX ? k : (HT_BYSLOT | (uint32_t)slot); }
struct DedupTable {
    DedupEntry *ent; uint64_t mask;                               // entries, table size - 1
    uint32_t *slotof;                                             // per phrase: its entry
    uint32_t *dslot; uint64_t *dhash; uint32_t *nd; uint32_t limit;   // the entries in use (appended by whoever created them) and their hashes
    uint32_t *overflow;                                           // != 0: more distinct phrases than `limit` (or a probe sequence too long): retry with a larger table
    uint32_t *abandon;                                            // the same news on a cache line of its own: read by every workgroup when it starts (`overflow` shares its line with the
                                                                  // counters the insertions bump -- 1.3 M workgroup starts reading THAT line cost 8 ms on S-32G)
};
__device__ __forceinline__ void ht_give_up(const DedupTable &t, uint32_t code) { *t.overflow = code; *t.abandon = 1u; }
// phrase j (hash h, bytes Y[ys..ys+len)) enters the table or finds its representative there
__device__ __forceinline__ void dedup_find_or_insert(const uint8_t *Y, const Spans &sp, const DedupTable &t, uint32_t j, tpos_t ys, uint32_t len, uint64_t h)
{
    const uint64_t filt = h >> 32;
    uint64_t slot = h & t.mask;
    for (uint32_t probe = 0; probe < HT_MAX_PROBES; ++probe, slot = (slot + 1) & t.mask) {
        unsigned long long cur = t.ent[slot].tab;
        if ((probe & 7u) == 7u && *t.overflow) return;          // the table is being abandoned (it may be full: no EMPTY entry would end the walk)
        if (cur == HT_EMPTY) {
            if (*t.overflow) return;
            cur = atomicCAS(&t.ent[slot].tab, (unsigned long long)HT_EMPTY, (unsigned long long)((filt << 32) | j));
            if (cur == HT_EMPTY) {                               // this phrase is the representative of a new entry
                const uint32_t k = atomicAdd(t.nd, 1u);
                if (k >= t.limit) { ht_give_up(t, 1u); return; }
                t.dslot[k] = (uint32_t)slot; t.dhash[k] = h; t.ent[slot].kidx = k;
                t.slotof[j] = k; atomicAdd(&t.ent[slot].cnt, 1u);
                return;
            }
        }
        if ((cur >> 32) == filt) {
            tpos_t rs; uint32_t rlen; phrase_span(sp, (uint32_t)cur, &rs, &rlen);
            if (rlen == len && str_equal(Y + ys, Y + rs, len)) { t.slotof[j] = entry_ref(t.ent, slot); atomicAdd(&t.ent[slot].cnt, 1u); return; }
        }
    }
    ht_give_up(t, 2u);
}
// One thread per phrase; phrases longer than LONG_PHRASE are only listed (a workgroup each, below).  The 256 phrases of a
// workgroup are neighbours in the text (~26 KB at p = 100): their bytes are brought into LDS by coalesced 16-byte loads
// and hashed / compared from there, so that every phrase crosses HBM once; what remains random is the read of the
// representative (a dictionary-sized hot set).  A window that does not fit (long phrases, scattered words) falls back to
// reading the phrase from memory.
constexpr uint32_t DD_TILE_BYTES = 32768 - 64;
// the 8-byte words of a string that starts at an arbitrary byte offset of an LDS array of 32-bit words, one after the
// other: two new words of LDS per 8 bytes (the third a shifted read needs is the one kept from the step before) -- the
// threads' offsets are ~110 bytes apart, i.e. these reads hit random banks, and they were what bounded the kernel
struct LdsWords {
    const uint32_t *t; uint32_t idx, sh, carry;
    __device__ __forceinline__ LdsWords(const uint32_t *t32, uint32_t off) : t(t32), idx(off >> 2), sh((off & 3u) * 8u), carry(t32[off >> 2]) {}
    __device__ __forceinline__ uint64_t next()
    {
        const uint32_t w1 = t[idx + 1], w2 = t[idx + 2];
        const uint32_t lo = (uint32_t)((((uint64_t)w1 << 32) | carry) >> sh), hi = (uint32_t)((((uint64_t)w2 << 32) | w1) >> sh);
        carry = w2; idx += 2;
        return ((uint64_t)hi << 32) | lo;
    }
};
// The order in which the workgroups visit the text (round 4).  A collection of similar sequences is a matrix: sequence h, locus c.  In text order the
// workgroups that run at the same time cover ~27 MB of text, one whole sequence: the representative a phrase is compared with (the same locus of the FIRST
// sequence, mostly) was last touched a sequence ago and comes from the MALL or HBM -- that read is 24 of the kernel's 42 ms on S-32G (stage by stage,
// DESIGN.md section 4).  Hardware workgroup b runs on XCD b mod 8.  With `period` = workgroups per sequence (an estimate: text workgroups / sequences fed),
// XCD x visits the loci c = x, x + 8, ... (columns of `chunk` workgroups) and, inside a column, sequence after sequence: the representatives of a column
// (~1 MB) stay in that XCD's 4 MB L2 while all `rows` sequences pass.  Only the ORDER changes -- any estimate gives the same table; a bad one just does
// not help.  period == 0: text order.  Grid = 8 * ceil(period / (8 * chunk)) * chunk * rows workgroups; those that fall outside the matrix return at once.
struct DedupOrder { uint32_t period, chunk, rows; };
// hardware workgroup b -> the workgroup of the text (or of any array laid out like the text) it works on; false: none (it lies outside the matrix)
__device__ __forceinline__ bool dedup_order_block(const DedupOrder &ord, uint32_t b, uint64_t nblk, uint64_t *B)
{
    *B = b;
    if (!ord.period) return true;
    const uint32_t x = b & 7u, tt = b >> 3, cell = ord.chunk * ord.rows;
    const uint32_t k = tt / cell, r = tt - k * cell, hrow = r / ord.chunk, i = r - hrow * ord.chunk;
    const uint64_t col = (uint64_t)(k * 8u + x) * ord.chunk + i;
    if (col >= ord.period) return false;
    *B = (uint64_t)hrow * ord.period + col;
    return *B < nblk;
}
// host: the order for nb workgroups of a text fed as nseq sequences; *grid = workgroups to launch.  period_req / chunk_req: the dedup_period / dedup_chunk switches
// (period 0 = nb / nseq when that is at least min_period workgroups, < 0 = text order).
inline DedupOrder make_dedup_order(uint64_t nb, uint64_t nseq, int64_t period_req, int64_t chunk_req, uint64_t min_period, uint64_t *grid)
{
    DedupOrder ord = {0u, 0u, 0u};
    *grid = nb;
    uint64_t period = period_req > 0 ? (uint64_t)period_req : (period_req == 0 && nseq >= 8 ? (nb + nseq / 2) / nseq : 0);
    if (period_req == 0 && period < min_period) period = 0;
    if (period && period < nb) {
        uint64_t chunk = chunk_req > 0 ? (uint64_t)chunk_req : 0;
        if (!chunk) { const uint64_t q0 = (period + 128) / 256 ? (period + 128) / 256 : 1; chunk = (period + 8 * q0 - 1) / (8 * q0); }
        if (chunk > period) chunk = period;
        const uint64_t q = (period + 8 * chunk - 1) / (8 * chunk), rows = (nb + period - 1) / period;
        const uint64_t g = 8 * q * chunk * rows;
        if (g < 0x7FFFFFFFULL && chunk * rows < 0xFFFFFFFFULL) { ord.period = (uint32_t)period; ord.chunk = (uint32_t)chunk; ord.rows = (uint32_t)rows; *grid = g; }
    }
    return ord;
}
// COOP (round 4): the representatives' bytes are read by the wave together (see below); false = every lane reads its own representative (rounds 2-3).
// The host takes COOP for a collection (>= 8 sequences fed) while its first, small table lasts: on a text of DISTINCT phrases the cooperative kernel is 2.4 x slower
// (S-3G: 80 against 34 ms; none of its parts explains it when switched off one by one -- r04ag_s3g_exp.log -- so the cause is open).
// phase (nullable, experiments): wall-clock ticks (10 ns) that thread 0 of a workgroup saw between the kernel's stages, summed into 64 x 8 counters.
template <bool COOP>
__global__ __launch_bounds__(BLOCK) void k_dedup_insert(const uint8_t *Y, Spans sp, uint64_t m, uint64_t seed, DedupTable t, uint32_t *longlist, uint32_t *nlong, uint8_t *last /*nullable: last[j] = Y[ye[j] - w], pfparser.hpp:599*/,
                                                        unsigned long long *phase, DedupOrder ord)
{
    constexpr uint32_t TILE_BYTES = DD_TILE_BYTES, NT = BLOCK;
    constexpr uint32_t CHUNK = 128u;       // bytes of the representative in flight per round trip (256: no gain, r04dv)
    __shared__ uint32_t tile[TILE_BYTES / 4 + 20];
    unsigned long long tk[6] = {0, 0, 0, 0, 0, 0};
    if (phase) tk[0] = wall_clock64();
    uint64_t B;
    if (!dedup_order_block(ord, blockIdx.x, (m + NT - 1) / NT, &B)) return;      // (see DedupOrder)
    const uint64_t j = B * NT + threadIdx.x;
    const bool live = j < m;
    tpos_t ys = 0; uint32_t len = 0;
    unsigned long long lo, hi;
    bool tiled;
    uint64_t base;
    {
        __shared__ unsigned long long wlo[BLOCK / WAVE], whi[BLOCK / WAVE];
        __shared__ uint32_t abandoned;
        // a table that is being abandoned (a non-repetitive text fills the first, small table after an eighth of its phrases): the workgroups
        // that have not started yet return at once instead of hashing their phrases for nothing (S-3G: 19 of 38 ms)
        if (threadIdx.x == 0) abandoned = *(volatile uint32_t *)t.abandon;
        __syncthreads();
        if (abandoned) return;
        if (phase) tk[1] = wall_clock64();
        if (live) phrase_span(sp, j, &ys, &len);
        const bool lng0 = live && len > LONG_PHRASE;
        if (lng0) longlist[atomicAdd(nlong, 1u)] = (uint32_t)j;
        // window of the workgroup's (short) phrases
        lo = (live && !lng0) ? (unsigned long long)ys : ~0ULL; hi = (live && !lng0) ? (unsigned long long)ys + len : 0ULL;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const unsigned long long a = __shfl_xor(lo, d), b = __shfl_xor(hi, d); lo = a < lo ? a : lo; hi = b > hi ? b : hi; }
        if ((threadIdx.x & 63) == 0) { wlo[threadIdx.x >> 6] = lo; whi[threadIdx.x >> 6] = hi; }
        __syncthreads();
        lo = wlo[0]; hi = whi[0];
#pragma unroll
        for (int v = 1; v < BLOCK / WAVE; ++v) { lo = wlo[v] < lo ? wlo[v] : lo; hi = whi[v] > hi ? whi[v] : hi; }
        if (hi == 0) return;                                     // nothing but long phrases (or nothing at all)
    }
    const bool lng = live && len > LONG_PHRASE;
    if (phase) tk[2] = wall_clock64();
    base = lo & ~15ULL;                                       // Y + base is 16-byte aligned when Y is: not assumed -- the loads below are byte-exact
    tiled = hi - base <= TILE_BYTES;
    if (tiled) {
        const uint32_t nb = (uint32_t)(hi - base);
        // the text buffer is padded in front of Y[0] and behind its end, so whole 16-byte pieces may be read
        for (uint32_t o = threadIdx.x * 16u; o < nb + 8u; o += NT * 16u) {
            uint4 v; __builtin_memcpy(&v, Y + base + o, 16);
            tile[o / 4] = v.x; tile[o / 4 + 1] = v.y; tile[o / 4 + 2] = v.z; tile[o / 4 + 3] = v.w;
        }
    }
    __syncthreads();
    if (phase) tk[3] = wall_clock64();
    const bool act = live && !lng;
    // the character in front of the phrase's closing window, while the phrase is at hand (in LDS, normally): a kernel of its own
    // read one sector per phrase for this byte
    if (live && last && (lng || !tiled)) last[j] = Y[ys + len - 1u - (uint32_t)sp.w];
    if (!tiled) { if (act) dedup_find_or_insert(Y, sp, t, (uint32_t)j, ys, len, str_hash(Y + ys, len, seed)); return; }
    const bool actm = act;
    const uint32_t off = actm ? (uint32_t)(ys - base) : 0u;
    if (actm && last) { const uint32_t o = off + len - 1u - (uint32_t)sp.w; last[j] = (uint8_t)(tile[o >> 2] >> (8u * (o & 3u))); }
    uint64_t h = 0;
    if (actm) { LdsWords hw(tile, off); h = hash_words(len, seed, [&hw]() { return hw.next(); }); }
    if (phase) tk[4] = wall_clock64();
    // the table: my phrase finds its entry or becomes one
    auto lookup = [&]() __attribute__((always_inline)) -> void {
        const uint64_t filt = h >> 32;
        uint64_t slot = h & t.mask;
        for (uint32_t probe = 0; probe < HT_MAX_PROBES; ++probe, slot = (slot + 1) & t.mask) {
            unsigned long long cur = t.ent[slot].tab;
            unsigned long long ri = t.ent[slot].rinfo;                    // requested together with the entry: the representative's bytes are one dependent load away, not three
            if ((probe & 7u) == 7u && *t.overflow) return;          // the table is being abandoned (it may be full: no EMPTY entry would end the walk)
            if (cur == HT_EMPTY) {
                if (*t.overflow) return;
                cur = atomicCAS(&t.ent[slot].tab, (unsigned long long)HT_EMPTY, (unsigned long long)((filt << 32) | (uint32_t)j));
                if (cur == HT_EMPTY) {
                    const uint32_t k = atomicAdd(t.nd, 1u);
                    if (k >= t.limit) { ht_give_up(t, 1u); return; }
                    t.ent[slot].rinfo = ((unsigned long long)ys << 16) | len;
                    t.dslot[k] = (uint32_t)slot; t.dhash[k] = h; t.ent[slot].kidx = k;
                    t.slotof[j] = k; atomicAdd(&t.ent[slot].cnt, 1u);
                    return;
                }
                ri = HT_NOINFO;
            }
            if ((cur >> 32) == filt) {
                tpos_t rs; uint32_t rlen;
                if (ri != HT_NOINFO) { rs = (tpos_t)(ri >> 16); rlen = (uint32_t)(ri & 0xFFFFu); }
                else phrase_span(sp, (uint32_t)cur, &rs, &rlen);
                if (rlen == len) {
                    // no early exit (an entry whose filter bits agree is the same phrase all but never), and the representative's
                    // bytes are requested CHUNK at a time: a wave is as slow as the longest of its 64 phrases (~470 bytes at
                    // p = 100), which was 15 dependent round trips when 32 bytes were in flight
                    const uint8_t *r = Y + rs;
                    uint64_t diff = 0;
                    LdsWords cw(tile, off);
                    for (uint32_t c0 = 0; c0 < len; c0 += CHUNK) {
                        uint64_t a[CHUNK / 8];
#pragma unroll
                        for (int q = 0; q < (int)(CHUNK / 8); q += 2) {      // 16 bytes per request: every lane reads another line, and the requests, not the bytes, are what the memory pipeline counts (callers keep 15 readable bytes behind every string)
                            const uint32_t i = c0 + 8u * (uint32_t)q;
                            uint4 v = make_uint4(0u, 0u, 0u, 0u);
                            if (i < len) __builtin_memcpy(&v, r + i, 16);
                            a[q] = ((uint64_t)v.y << 32) | v.x; a[q + 1] = ((uint64_t)v.w << 32) | v.z;
                        }
#pragma unroll
                        for (int q = 0; q < (int)(CHUNK / 8); ++q) {
                            const uint32_t i = c0 + 8u * (uint32_t)q;
                            if (i < len) { uint64_t x = a[q] ^ cw.next(); if (len - i < 8) x &= (1ULL << (8 * (len - i))) - 1ULL; diff |= x; }
                        }
                    }
                    if (!diff) { t.slotof[j] = entry_ref(t.ent, slot); atomicAdd(&t.ent[slot].cnt, 1u); return; }
                }
            }
        }
        ht_give_up(t, 2u);
    };
    if constexpr (!COOP) { if (actm) lookup(); }
    else {
        // the representatives' bytes are read by the wave TOGETHER.  A lane that reads the 7 x 16 bytes of its own representative touches a cache
        // line of its own with every request (64 lines per load instruction); here the 8 lanes of a group read 128 contiguous bytes of ONE representative
        // (1-2 lines), eight representatives per instruction, and compare them with the phrase's bytes in LDS; a ballot brings the verdict back to the
        // phrase's own lane.  Every lane walks its probe sequence as before; the wave meets for the comparison whenever some lane has a candidate.
        const uint32_t lane = threadIdx.x & 63u, grp = lane >> 3, sub = lane & 7u;
        const uint64_t filt = h >> 32;
        uint64_t slot = h & t.mask;
        uint32_t probe = 0;
        bool active = actm;
        while (__any(active ? 1 : 0)) {
            bool cand = false, won = false; tpos_t rs = 0; uint32_t kx = HT_NOIDX;
            if (active) {
                unsigned long long cur = t.ent[slot].tab;
                unsigned long long ri = t.ent[slot].rinfo;
                kx = t.ent[slot].kidx;                                   // (the same 32-byte sector: no further round trip when the phrase turns out to be this entry's)
                bool fin = false;
                if ((probe & 7u) == 7u && *t.overflow) fin = true;       // the table is being abandoned (it may be full: no EMPTY entry would end the walk)
                else if (cur == HT_EMPTY) {
                    if (*t.overflow) fin = true;
                    else {
                        cur = atomicCAS(&t.ent[slot].tab, (unsigned long long)HT_EMPTY, (unsigned long long)((filt << 32) | (uint32_t)j));
                        if (cur == HT_EMPTY) { won = true; fin = true; }
                        ri = HT_NOINFO; kx = HT_NOIDX;
                    }
                }
                if (fin) active = false;
                else if ((cur >> 32) == filt) {
                    uint32_t rlen;
                    if (ri != HT_NOINFO) { rs = (tpos_t)(ri >> 16); rlen = (uint32_t)(ri & 0xFFFFu); }
                    else phrase_span(sp, (uint32_t)cur, &rs, &rlen);
                    cand = rlen == len;
                }
            }
            // the wave's new entries take their dense indices with ONE update of the counter (a text of distinct phrases -- S-3G -- is bound by exactly
            // that counter when every lane bumps it by itself: 31 M atomics on one address)
            const unsigned long long wm = __ballot(won ? 1 : 0);
            if (wm) {
                const int leader = __ffsll((long long)wm) - 1;
                uint32_t kbase = 0;
                if ((int)lane == leader) kbase = atomicAdd(t.nd, (uint32_t)__popcll(wm));
                kbase = __shfl(kbase, leader);
                if (won) {
                    const uint32_t k = kbase + (uint32_t)__popcll(wm & ((1ULL << lane) - 1ULL));
                    if (k >= t.limit) ht_give_up(t, 1u);
                    else {
                        t.ent[slot].rinfo = ((unsigned long long)ys << 16) | len;
                        t.dslot[k] = (uint32_t)slot; t.dhash[k] = h; t.ent[slot].kidx = k;
                        t.slotof[j] = k; atomicAdd(&t.ent[slot].cnt, 1u);
                    }
                }
            }
            uint32_t bad = 0;
            const uint32_t clen = cand ? len : 0u;
            uint32_t maxlen = clen;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = __shfl_xor(maxlen, d); maxlen = o > maxlen ? o : maxlen; }
            for (uint32_t c0 = 0; c0 < maxlen; c0 += 128u) {
                const uint32_t o = c0 + sub * 16u;
                uint4 a[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int p = r * 8 + (int)grp;
                    const uint32_t plen = __shfl(clen, p);
                    const unsigned long long prs = __shfl((unsigned long long)rs, p);
                    a[r] = make_uint4(0u, 0u, 0u, 0u);
                    if (o < plen) __builtin_memcpy(&a[r], Y + prs + o, 16);      // (callers keep 15 readable bytes behind every string)
                }
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int p = r * 8 + (int)grp;
                    const uint32_t plen = __shfl(clen, p), poff = __shfl(off, p);
                    bool d = false;
                    if (o < plen) {
                        const uint32_t bo = poff + o, wi = bo >> 2, sh = (bo & 3u) * 8u;
                        const uint32_t w0 = tile[wi], w1 = tile[wi + 1], w2 = tile[wi + 2], w3 = tile[wi + 3], w4 = tile[wi + 4];
                        uint32_t d0 = (uint32_t)((((uint64_t)w1 << 32) | w0) >> sh) ^ a[r].x, d1 = (uint32_t)((((uint64_t)w2 << 32) | w1) >> sh) ^ a[r].y;
                        uint32_t d2 = (uint32_t)((((uint64_t)w3 << 32) | w2) >> sh) ^ a[r].z, d3 = (uint32_t)((((uint64_t)w4 << 32) | w3) >> sh) ^ a[r].w;
                        const uint32_t nv = plen - o;                              // bytes of this piece that belong to the phrase (>= 1)
                        if (nv < 16u) {
                            if (nv <= 12u) d3 = 0; else d3 &= (1u << (8u * (nv - 12u))) - 1u;
                            if (nv <= 8u) d2 = 0; else if (nv < 12u) d2 &= (1u << (8u * (nv - 8u))) - 1u;
                            if (nv <= 4u) d1 = 0; else if (nv < 8u) d1 &= (1u << (8u * (nv - 4u))) - 1u;
                            if (nv < 4u) d0 &= (1u << (8u * nv)) - 1u;
                        }
                        d = (d0 | d1 | d2 | d3) != 0u;
                    }
                    const unsigned long long bm = __ballot(d ? 1 : 0);
                    if ((int)grp == r) bad |= (uint32_t)((bm >> (8u * sub)) & 0xFFull);
                }
            }
            if (cand && !bad) {
                t.slotof[j] = kx != HT_NOIDX ? kx : (HT_BYSLOT | (uint32_t)slot);
                atomicAdd(&t.ent[slot].cnt, 1u);
                active = false;
            }
            if (active) {
                ++probe; slot = (slot + 1) & t.mask;
                if (probe >= HT_MAX_PROBES) { ht_give_up(t, 2u); active = false; }
            }
        }
    }
    if (phase && threadIdx.x == 0) {
        tk[5] = wall_clock64();
        unsigned long long *ph = phase + (blockIdx.x & 63u) * 8u;
        for (int k = 0; k < 5; ++k) atomicAdd(&ph[k], tk[k + 1] - tk[k]);
        atomicAdd(&ph[5], 1ULL);
    }
}
// ---- long phrases (e.g. a 10 Mbp run of N is ONE phrase: wang_hash(0) % 100 != 0): one workgroup
// per phrase hashes and compares it cooperatively; thread 0 walks the probe sequence
constexpr int DL_THREADS = 1024;       // the waves of one whole CU stream the phrase
__global__ __launch_bounds__(DL_THREADS) void k_dedup_insert_long(const uint8_t *Y, Spans sp, const uint32_t *longlist, uint64_t seed, DedupTable t)
{
    __shared__ uint64_t part[DL_THREADS];
    __shared__ unsigned long long s_cur, s_h;
    __shared__ int s_state;                                  // 0: next probe, 1: done, 2: compare with the entry's representative
    __shared__ uint32_t s_diff;
    const uint32_t j = longlist[blockIdx.x];
    tpos_t ys; uint32_t len; phrase_span(sp, j, &ys, &len);
    {   // thread t chains the 8-byte words t, t + 1024, t + 2048, ... (coalesced reads), thread 0 folds the partial hashes
        uint64_t hp = seed + threadIdx.x;
        for (uint64_t k = 8ULL * threadIdx.x; k < len; k += 8ULL * DL_THREADS) {
            uint64_t v = ld8(Y + ys + k);
            if (len - k < 8) v &= (1ULL << (8 * (len - k))) - 1ULL;
            hp = (hp ^ v) * 0xD6E8FEB86659FD93ULL; hp ^= hp >> 32;
        }
        part[threadIdx.x] = hp;
    }
    if (threadIdx.x == 0) s_diff = 0;
    __syncthreads();
    if (threadIdx.x == 0) { uint64_t hh = seed ^ len; for (int k = 0; k < DL_THREADS; ++k) hh = mix64(hh ^ part[k]); s_h = hh; }
    __syncthreads();
    const uint64_t h = s_h;
    const uint64_t filt = h >> 32;
    uint64_t slot = h & t.mask;
    for (uint32_t probe = 0; probe < HT_MAX_PROBES; ++probe) {
        if (threadIdx.x == 0) {
            int st = 0;
            unsigned long long cur = t.ent[slot].tab;
            if (cur == HT_EMPTY) {
                if (*t.overflow) st = 1;
                else {
                    cur = atomicCAS(&t.ent[slot].tab, (unsigned long long)HT_EMPTY, (unsigned long long)((filt << 32) | j));
                    if (cur == HT_EMPTY) {
                        const uint32_t k = atomicAdd(t.nd, 1u);
                        if (k >= t.limit) ht_give_up(t, 1u);
                        else { t.dslot[k] = (uint32_t)slot; t.dhash[k] = h; t.ent[slot].kidx = k; t.slotof[j] = k; atomicAdd(&t.ent[slot].cnt, 1u); }
                        st = 1;
                    }
                }
            }
            if (!st && (cur >> 32) == filt) st = 2;
            s_cur = cur; s_state = st;
        }
        __syncthreads();
        const int st = s_state;
        if (st == 1) return;
        if (st == 2) {
            tpos_t rs; uint32_t rlen; phrase_span(sp, (uint32_t)s_cur, &rs, &rlen);
            if (rlen != len) { if (threadIdx.x == 0) s_diff = 1; }
            else {
                uint64_t d = 0;
                for (uint64_t k = 8ULL * threadIdx.x; k < len; k += 8ULL * DL_THREADS) {
                    uint64_t v = ld8(Y + ys + k) ^ ld8(Y + rs + k);
                    if (len - k < 8) v &= (1ULL << (8 * (len - k))) - 1ULL;
                    d |= v;
                }
                if (d) s_diff = 1;
            }
            __syncthreads();
            const bool same = s_diff == 0;
            __syncthreads();
            if (same) { if (threadIdx.x == 0) { t.slotof[j] = entry_ref(t.ent, slot); atomicAdd(&t.ent[slot].cnt, 1u); } return; }
            if (threadIdx.x == 0) s_diff = 0;
        }
        slot = (slot + 1) & t.mask;
        __syncthreads();
    }
    if (threadIdx.x == 0) ht_give_up(t, 2u);
}
// ids = position of an entry's hash among the sorted hashes of the entries in use (deterministic whatever thread created
// the entry): rep[id] = its representative phrase, occw[id] = its occurrences; the entry then holds the id
__global__ __launch_bounds__(BLOCK) void k_dedup_assign(const uint32_t *order, uint64_t nd, DedupTable t, uint32_t *rep, uint32_t *occw, uint32_t *idofk)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= nd) return;
    const uint32_t k = order[i], slot = t.dslot[k];
    rep[i] = (uint32_t)t.ent[slot].tab; occw[i] = t.ent[slot].cnt + 1u;
    t.ent[slot].tab = (unsigned long long)i; idofk[k] = (uint32_t)i;
}
__global__ __launch_bounds__(BLOCK) void k_dedup_ids(const DedupEntry *ent, const uint32_t *idofk, const uint32_t *slotof, uint64_t m, uint32_t *pid)
{
    const uint64_t j = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (j >= m) return;
    const uint32_t v = slotof[j];
    pid[j] = (v & HT_BYSLOT) ? (uint32_t)ent[v & ~HT_BYSLOT].tab : idofk[v];
}
__global__ __launch_bounds__(BLOCK) void k_iota_u32(uint32_t *v, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) v[i] = (uint32_t)i;
}
// wlen1[id] = phrase length + 1 (EndOfWord)
__global__ __launch_bounds__(BLOCK) void k_word_lengths(Spans sp, const uint32_t *rep, uint64_t dwords, uint32_t *wlen1)
{
    const uint64_t id = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (id >= dwords) return;
    tpos_t ys; uint32_t len; phrase_span(sp, rep[id], &ys, &len);
    wlen1[id] = len + 1;
}

__device__ __forceinline__ uint32_t upper_bound_u32(const uint32_t *a, uint32_t n, uint32_t x)
{   // first index with a[idx] > x
    uint32_t lo = 0, hi = n;
    while (lo < hi) { uint32_t mid = lo + ((hi - lo) >> 1); if (a[mid] <= x) lo = mid + 1; else hi = mid; }
    return lo;
}

// D' : words in id order, each followed by EndOfWord, then EndOfDict.  One thread = 16 output bytes.
// srcstart[id] = Y offset of the word's first byte (or offset into another dictionary image).
__global__ __launch_bounds__(BLOCK) void k_dict_build(const uint8_t *src, const tpos_t *srcstart, const uint32_t *ws, uint32_t dwords, uint64_t dsize,
                                                      uint8_t *dict, uint32_t *wordid)
{
    const uint64_t x0 = ((uint64_t)blockIdx.x * BLOCK + threadIdx.x) * 16;
    if (x0 >= dsize) return;
    uint32_t id = upper_bound_u32(ws, dwords + 1, (uint32_t)x0) - 1; // ws[id] <= x0
    uint32_t wbeg = id < dwords ? ws[id] : 0u, wend = id < dwords ? ws[id + 1] : 0xFFFFFFFFu;
    tpos_t sbeg = id < dwords ? srcstart[id] : 0;
    if (x0 + 16 <= dsize) {        // whole 16-byte piece: bytes and word ids are assembled in registers and stored as 16-byte vectors
        uint32_t bytes[4] = {0, 0, 0, 0}, ids[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint32_t x = (uint32_t)x0 + k;
            while (id < dwords && x >= wend) { ++id; if (id < dwords) { wbeg = wend; wend = ws[id + 1]; sbeg = srcstart[id]; } }
            uint32_t c;
            if (id >= dwords) c = EndOfDict;
            else c = (x + 1 == wend) ? (uint32_t)EndOfWord : (uint32_t)src[sbeg + (x - wbeg)];
            bytes[k >> 2] |= c << (8 * (k & 3));
            ids[k] = id;
        }
        *reinterpret_cast<uint4 *>(dict + x0) = make_uint4(bytes[0], bytes[1], bytes[2], bytes[3]);
        if (wordid) {
            uint4 *wo = reinterpret_cast<uint4 *>(wordid + x0);
#pragma unroll
            for (int k = 0; k < 4; ++k) wo[k] = make_uint4(ids[4 * k], ids[4 * k + 1], ids[4 * k + 2], ids[4 * k + 3]);
        }
        return;
    }
    for (uint64_t x = x0; x < dsize; ++x) {
        while (id < dwords && x >= ws[id + 1]) ++id;
        uint8_t c;
        if (id >= dwords) c = EndOfDict;
        else { const uint32_t off = (uint32_t)x - ws[id]; c = (x + 1 == ws[id + 1]) ? EndOfWord : src[srcstart[id] + off]; }
        dict[x] = c;
        if (wordid) wordid[x] = id;
    }
}
__global__ __launch_bounds__(BLOCK) void k_rep_starts(Spans sp, const uint32_t *rep, uint64_t dwords, tpos_t *srcstart)
{
    const uint64_t id = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (id >= dwords) return;
    tpos_t ys; uint32_t len; phrase_span(sp, rep[id], &ys, &len);
    srcstart[id] = ys;
}
__global__ __launch_bounds__(BLOCK) void k_fill_u8(uint8_t *p, uint64_t n, uint8_t v)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ __launch_bounds__(BLOCK) void k_set_u32(uint32_t *p, uint64_t idx, uint32_t v)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) p[idx] = v;
}
__global__ __launch_bounds__(BLOCK) void k_set_u64(uint64_t *p, uint64_t idx, uint64_t v)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) p[idx] = v;
}

} // namespace pfp
