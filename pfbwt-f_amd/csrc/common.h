// pfbwt-f_amd/csrc/common.h -- context, device arena, launch + profiling helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <string>
#include <vector>
#include "../../include/pfbwt_hip.h"
#include "devmem.h"
#include "../../include/pfbwt_hip_dev.h"

namespace pfp {

typedef uint32_t idx_t;           // device index type of this build: texts / dictionaries < 2^32

constexpr int BLOCK = 256;        // 4 wave64 per workgroup
constexpr int WAVE = 64;

// Reference constants, include/utils.h:8-10
constexpr uint8_t Dollar = 2, EndOfWord = 1, EndOfDict = 0;

// ---- kernel ids for the profile table ----------------------------------------------------------
enum KernelId {
    K_TRIGGER_SCAN, K_PHRASE_ENDS, K_PHRASE_HASH, K_PHRASE_HASH_LONG, K_DEDUP_HEADS, K_DEDUP_LONG,
    K_DICT_BUILD, K_RADIX_HIST, K_RADIX_SCATTER, K_SCAN_REDUCE, K_SCAN_SPINE, K_SCAN_APPLY,
    K_SS_INIT_KEYS, K_SS_HEADS, K_SS_MAKE_KEYS, K_SS_WRITE_RANK, K_SS_FLAG_ACTIVE, K_COMPACT,
    K_WORD_RANK, K_PARSE_RANKS, K_DICT_SORTED, K_PBWT_ROWS, K_EMIT_COUNT, K_EMIT, K_RUNS, K_SAMPLES, K_MISC, K_EMIT_BIG, K_FILL, K_CLASS_SORT, K_FASTA, K_EMIT_LARGE, K_REC_PARSE, K_REC_DEDUP, K_REC_ASSEMBLE,
    K_COUNT_
};
static const char *const kernel_names[K_COUNT_] = {
    "trigger_scan", "phrase_ends", "phrase_hash", "phrase_hash_long", "dedup_heads", "dedup_long",
    "dict_build", "radix_hist", "radix_scatter", "scan_reduce", "scan_spine", "scan_apply",
    "ss_init_keys", "ss_heads", "ss_make_keys", "ss_write_rank", "ss_flag_active", "compact",
    "word_rank", "parse_ranks", "dict_sorted", "pbwt_rows", "emit_count", "emit", "runs", "samples", "misc", "emit_big", "fill", "class_sort", "fasta_strip", "emit_large", "rec_parse", "rec_dedup", "rec_assemble"};

struct ProfRec { uint64_t launches = 0; double ms = 0, bytes = 0; };

// Two-ended bump allocator over one address range (devmem.h: physical memory is committed as the two ends advance): results
// of a stage live at the low end, scratch at the high end (released with mark/release).
struct Arena {
    VmRegion vm;
    char *base = nullptr;
    size_t cap = 0, lo = 0, hi = 0, want = 0;
    size_t peak_lo = 0, peak_hi_bytes = 0;      // high-water marks (PFP_VERBOSE)
    bool failed = false;
    void reset() { lo = 0; hi = cap; failed = false; }
    void *alloc_lo(size_t bytes)
    {
        size_t a = (lo + 255) & ~(size_t)255;
        if (a + bytes > hi || !vm.commit(a, a + bytes)) { failed = true; want += bytes; return nullptr; }
        lo = a + bytes; if (lo > peak_lo) peak_lo = lo;
        return base + a;
    }
    void *alloc_hi(size_t bytes)
    {
        size_t b = (bytes + 255) & ~(size_t)255;
        if (b > hi || hi - b < lo || !vm.commit(hi - b, hi)) { failed = true; want += bytes; return nullptr; }
        hi -= b; if (cap - hi > peak_hi_bytes) peak_hi_bytes = cap - hi;
        return base + hi;
    }
    // address space only: the caller commits what it is going to touch (commit_range) -- arrays sized for the worst case of
    // which a fraction is used (run samples: r is not known in advance)
    void *reserve_lo(size_t bytes)
    {
        size_t a = (lo + 255) & ~(size_t)255;
        if (a + bytes > hi) { failed = true; want += bytes; return nullptr; }
        lo = a + bytes;
        return base + a;
    }
    bool commit_range(const void *p, size_t bytes) { const size_t o = (size_t)((const char *)p - base); return vm.commit(o, o + bytes); }
    size_t offset_of(const void *p) const { return (size_t)((const char *)p - base); }
    size_t mark_hi() const { return hi; }
    void release_hi(size_t m) { hi = m; }
    size_t mark_lo() const { return lo; }
    void release_lo(size_t m) { lo = m; }
};

// Route and tuning switches of a context.  The defaults are the product's; tests and A/B measurements change them with
// pfp_debug_set (include/pfbwt_hip_dev.h) or -- only in a process started with PFP_TEST_HOOKS=1 -- through PFP_<NAME>
// environment variables that pfp_create reads into the new context.  Nothing here is latched per process.
struct Tunables {
    int verbose = 0;                   // PFP_VERBOSE (diagnostics on stderr; honoured without PFP_TEST_HOOKS: it changes no route)
    int seg_grid = 768;                // workgroups per pass of the global radix sort (3 per CU of a 256-CU device)
    int seg_stage = 1;                 // global radix scatter staged through LDS
    int sort_k = 0;                    // 1: plain doubling in every refinement round, 3: three ranks in every round but the run round
    int sort_no_table = 0;             // the K = 3 rounds follow the chains themselves
    uint32_t class_sort_maxrange = 0;  // smaller LDS class-sort ranges (reaches the large-class route on small inputs)
    int dedup_table_log2 = 0;          // a first phrase table that overflows
    int no_trigger_table = 0;          // trigger test by hashing every window
    uint64_t emit_chunk_rows = 3ULL << 30;   // rows per emission window (32-bit offsets inside a window: < 2^32 with room for a straddling group).  2^30 until round 3: every window pays ~0.5 ms of small launches and host round trips (S-32G: 30 windows 96.8 ms, 15 windows 87.1, 8 windows 82.0)
    uint32_t fill_subs = 2;            // super-tiles (4 x 4096 rows) per workgroup of k_fill
    uint64_t sample_cap = ~0ULL;       // cap of the one-pass run-sample arrays (forces the two-pass fallback)
    int no_runaware = 0;               // -r with every row enumerated, as with a full SA
    long big_group_members = -2;       // -2: BIG_GROUP_MEMBERS (emit.h); < 0 otherwise: never take the sort route
    int force_wide_rows = 0;           // 64-bit row counters on small texts
    uint64_t ingest_block_bytes = 0;   // block size of the file reader (0: 64 MiB)
    int expand_dma = 1;                // pfp_bwt_get_expanded: the copy engine takes blocks from the back while host threads write runs from the front (page-locked destinations)
    int ingest_readers = 0;            // pread threads of the file reader (0: one per CPU, 4 / 8 / 16)
    uint32_t emit_group_rows = 4096;   // rows per batch of the group-stationary emission of the special rows (0: every special row through k_emit; smaller: more groups left to k_emit)
    int dict_text_rounds = -1;         // dictionary suffix sort by text rounds: -1 = when the collection is not repetitive (dictionary > text / 8), 0 never, 1 always
    int force_run_round = 0;           // the run round of the dictionary sort even without a run of 256 equal bytes (tests)
    int int_key_symbols = 3;           // symbols of the parse in the initial sort key: 3 where 3 x symbol bits <= 64 (S-32G: 21-bit symbols, 8 radix passes instead of 6, one refinement round less to pay for: parse BWT 135.7 -> 131.9 ms), else 2
    int no_slot_records = 0;           // k_emit_slots by two gathers (word id | preceding byte, then the word record): the route of dictionaries with words of 64 Mbase and more
    uint64_t fasta_chunk_bytes = 0;    // size of the raw-FASTA device buffers (0: 1 MiB ... 256 MiB by the size of the first call)
    int parse_rec = -1;                // suffix sort of the parse through a level-2 prefix-free parse (recsort.h): -1 = when the parse is long and repetitive, 0 never, 1 wherever the route can run
    int parse_rec_p2 = 4;              // its modulus: one symbol in p2 ends a level-2 phrase
    uint64_t parse_rec_min = 1u << 21; // shortest parse that takes it (below: launch latencies, not data, bound either route)
    int parse_rec_depth = 1;           // levels (the names of one level are as long as a second level's dictionary on the collections measured)
    uint32_t parse_rec_tile_rows = 0;  // rows per assembly batch (0: a full LDS tile; smaller: reaches the large-class route on small inputs)
    int dict_rec = -1;                 // suffix sort of the dictionary through a level-2 parse of the dictionary (dictrec.h): -1 = when the collection is repetitive, 0 never, 1 whenever the route can run
    int dict_rec_p2 = 16;              // its modulus (windows of four bytes)
    int dedup_variant = -1;            // k_dedup_insert<COOP> (parse.h): 1 = the representatives read by the wave together, 0 = by every lane for itself (rounds 2-3), -1 = 1 for a collection while its first table lasts
    int64_t dedup_period = 0;          // k_dedup_insert, order of the workgroups (parse.h, DedupOrder): workgroups per sequence; 0 = text workgroups / sequences fed, -1 = text order
    int64_t dedup_chunk = 0;           // workgroups per column (0 = about 32)
    int dedup_phases = 0;              // != 0: the stages of k_dedup_insert timed inside the kernel and printed (experiments)
    int parse_rec_table_log2 = 0;      // log2 of the level-2 phrase table (tests: a table that overflows -> doubling route)
};

} // namespace pfp

struct pfp_ctx {
    int w = 10; uint64_t p = 100; unsigned flags = 0; int device = 0;
    hipStream_t stream = nullptr;
    pfp::Arena arena;
    size_t arena_request = 0;
    // error detail
    uint64_t err_pos = 0; int err_ch = 0;
    // --- text staging (device): tb = 16 guard bytes (tb[15] = Dollar) + X + w Dollars + slack
    pfp::VmRegion text;          // address range of tb; committed as the text grows (never re-allocated, never copied)
    uint8_t *tb = nullptr; size_t tb_cap = 0; uint64_t n = 0; uint64_t text_hint = 0; uint64_t nseq = 0 /* sequences fed (a hint for the order in which the de-duplication visits the text) */;
    struct RowViewPending { const uint8_t *src = nullptr; uint64_t count = 0, len = 0, stride = 0; } view;      // pfp_parse_feed_device_view: the text is still the caller's rows (read in place by the trigger scan of pfp_parse_finalize)
    uint64_t left_ctx = 0;       // bytes of left context fed in front of this shard's text (pfp_parse_feed_left_context)
    uint64_t tb_n = 0;           // bytes of tb that hold the text of the current parse (0: none -- merged or loaded state)
    // --- parse results (device, arena low end)
    int stage = 0;             // 0 feeding, 1 parsed, 2 parse-bwt done
    uint64_t m = 0, dwords = 0, dsize = 0;
    uint64_t *d_ye = nullptr;       // m: Y-coordinate of each phrase's last byte (= sai); text positions are 64-bit
    uint32_t *d_pid = nullptr;      // m: phrase -> dictionary word id (D' order)
    uint32_t *d_parse = nullptr;    // m: 1-based ranks
    uint8_t *d_last = nullptr;      // m
    uint8_t *d_dict = nullptr;      // D': dictionary in word-id order (dsize bytes)
    uint32_t *d_ws = nullptr;       // dwords+1 word starts in D'
    uint32_t *d_wordid = nullptr;   // dsize: word id of each D' offset
    uint32_t *d_wrank = nullptr;    // dwords: word id -> 0-based rank (nullptr: identity)
    uint32_t *d_occ = nullptr;      // dwords, by rank
    uint8_t *d_sdict = nullptr;     // sorted .dict image (dsize), by rank
    uint32_t *d_gsa = nullptr;      // dsize: suffix array of D'
    uint8_t *d_sflag = nullptr;     // dsize (text-round sort only): per slot, 1 = the suffix there starts a word
    uint32_t *d_srank = nullptr;    // dsize: per suffix-array SLOT, the first slot of its class of equal suffixes
    uint2 *d_grank = nullptr;       // dsize: { class-head slot, covered-prefix end } of each D' offset (sufsort.h)
    bool gsa_valid = false;
    // --- parse-BWT results
    uint64_t nrows = 0;
    uint8_t *d_bwlast = nullptr; uint32_t *d_ilist = nullptr; uint64_t *d_bwsai = nullptr;
    uint8_t *d_bwl_il = nullptr;    // bwlast in ilist order, bwl_il[k] = bwlast[ilist[k]] (nullptr: not made -- loaded files, 2^29 parse rows or more)
    // --- BWT results
    uint64_t nout = 0, runs = 0, esa_pairs = 0, easy = 0, hard = 0, slice_begin = 0, slice_rows = 0;
    uint8_t *d_bwt = nullptr; void *d_sa = nullptr; void *d_ssa = nullptr; void *d_esa = nullptr;
    bool have_sa = false, have_rssa = false;
    uint64_t *d_ma = nullptr; uint64_t ma_words = 0;      // marker array (pfp_marker_array)
    size_t ma_lo_mark = (size_t)-1, ma_lo_end = 0;        // where its result sits at the low end of the arena (released by the next call)
    size_t lo_after_parse = 0, lo_after_pbwt = 0, emit_scratch_mark = 0;
    // --- instrumentation
    bool prof_on = false; uint64_t prof_mask = ~0ULL;
    pfp::ProfRec prof[pfp::K_COUNT_];
    struct PendingEv { hipEvent_t a, b; int id; double bytes; };
    std::vector<PendingEv> pending;
    std::vector<hipEvent_t> ev_pool;
    double stage_ms[3] = {0, 0, 0};
    int hip_err = 0;
    // --- ingest: pageable host memory goes through two pinned staging buffers, so that the host-side copy of chunk
    //     k + 1 overlaps the DMA of chunk k (pinned sources are copied directly)
    uint8_t *hstage[2] = {nullptr, nullptr}; hipEvent_t hstage_ev[2] = {nullptr, nullptr}; bool hstage_used[2] = {false, false};
    pfp::Tunables tun;
    // --- raw FASTA ingest (csrc/fasta.h): two device buffers for raw chunks (the upload of one overlaps the stripping of the other)
    struct FastaIngest {
        uint8_t *raw[2] = {nullptr, nullptr}; size_t rawcap = 0; bool used[2] = {false, false};
        hipStream_t copy = nullptr; bool copy_ready = false; hipEvent_t ev_copied[2] = {nullptr, nullptr}, ev_free[2] = {nullptr, nullptr};
        uint8_t *tiles = nullptr; size_t tiles_cap = 0;                 // per-tile summaries of one chunk
        unsigned long long *d_tot = nullptr, *h_tot = nullptr;          // device / page-locked host: kept bytes, header starts, end state, flags
        uint32_t state = 2; bool started = false; uint64_t records = 0; // the stream's state machine (2 = at a line start)
        std::vector<uint64_t> rec_raw, rec_pos;                         // records that started in the last pfp_parse_feed_fasta call
    } fa;
    uint8_t *h_bwt = nullptr; void *h_sa = nullptr;                      // pfp_bwt_build_stream: host destinations, filled window by window during the emission
    uint8_t *ing_buf[16] = {};                                           // page-locked blocks of the file reader (csrc/ingest.h)
    uint64_t ing_next_off = 0;
    std::vector<std::string> doc_names; std::vector<uint64_t> doc_starts;   // records of the last pfp_parse_feed_fasta_file(PFP_FASTA_RECORDS)
    uint64_t hash_seed = 0x9E3779B97F4A7C15ULL;
    uint32_t *d_trigtab = nullptr;      // w <= 10: one bit per k-mer, "wang_hash(kmer) % p == 0" (128 KiB for w = 10; lives in LDS during the trigger scan)
};

namespace pfp {

inline hipEvent_t ev_get(pfp_ctx *c)
{
    if (!c->ev_pool.empty()) { hipEvent_t e = c->ev_pool.back(); c->ev_pool.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}
inline void prof_collect(pfp_ctx *c)
{
    if (c->pending.empty()) return;
    (void)hipStreamSynchronize(c->stream);
    for (auto &pe : c->pending) {
        float ms = 0; (void)hipEventElapsedTime(&ms, pe.a, pe.b);
        c->prof[pe.id].launches++; c->prof[pe.id].ms += ms; c->prof[pe.id].bytes += pe.bytes;
        c->ev_pool.push_back(pe.a); c->ev_pool.push_back(pe.b);
    }
    c->pending.clear();
}
struct ProfScope {
    pfp_ctx *c; int id; double bytes; hipEvent_t a = nullptr; bool on = false;
    ProfScope(pfp_ctx *c_, int id_, double bytes_) : c(c_), id(id_), bytes(bytes_)
    {
        on = c->prof_on && ((c->prof_mask >> id) & 1ULL);
        if (on) { a = ev_get(c); (void)hipEventRecord(a, c->stream); }
    }
    ~ProfScope()
    {
        if (on) {
            hipEvent_t b = ev_get(c); (void)hipEventRecord(b, c->stream);
            c->pending.push_back({a, b, id, bytes});
            if (c->pending.size() > 4096) prof_collect(c);
        }
    }
};

// LAUNCH(ctx, kernel-id, algorithmic bytes, kernel, grid, args...)
#define PFP_LAUNCH(ctx, id, bytes, kernel, grid, ...)                                             \
    do {                                                                                          \
        pfp::ProfScope ps_((ctx), (id), (double)(bytes));                                         \
        if ((ctx)->tun.verbose >= 3) { fprintf(stderr, "[pfbwt_hip] launch %s grid %u\n", #kernel, (unsigned)(grid)); fflush(stderr); } \
        hipLaunchKernelGGL(kernel, dim3((unsigned)(grid)), dim3(pfp::BLOCK), 0, (ctx)->stream, __VA_ARGS__); \
        hipError_t le_ = hipGetLastError();            /* a rejected launch (grid, LDS size) must not pass as stale output */ \
        if ((ctx)->tun.verbose >= 3 && le_ == hipSuccess) le_ = hipStreamSynchronize((ctx)->stream);   /* fault hunting: one kernel at a time */ \
        if (le_ != hipSuccess) { (ctx)->hip_err = (int)le_; (ctx)->err_ch = (int)le_;                 \
            fprintf(stderr, "[pfbwt_hip] launch of %s failed: %s (%s:%d)\n", #kernel, hipGetErrorString(le_), __FILE__, __LINE__); \
            return PFP_E_HIP; }                                                                       \
    } while (0)

// the same with an explicit workgroup size
#define PFP_LAUNCH_B(ctx, id, bytes, kernel, grid, block, ...)                                    \
    do {                                                                                          \
        pfp::ProfScope ps_((ctx), (id), (double)(bytes));                                         \
        hipLaunchKernelGGL(kernel, dim3((unsigned)(grid)), dim3((unsigned)(block)), 0, (ctx)->stream, __VA_ARGS__); \
        hipError_t le_ = hipGetLastError();                                                       \
        if (le_ != hipSuccess) { (ctx)->hip_err = (int)le_; (ctx)->err_ch = (int)le_;                 \
            fprintf(stderr, "[pfbwt_hip] launch of %s failed: %s (%s:%d)\n", #kernel, hipGetErrorString(le_), __FILE__, __LINE__); \
            return PFP_E_HIP; }                                                                       \
    } while (0)

#define PFP_HIP(ctx, expr)                                                                        \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) { (ctx)->hip_err = (int)e_; (ctx)->err_ch = (int)e_;                \
            fprintf(stderr, "[pfbwt_hip] %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return PFP_E_HIP; }                                                                   \
    } while (0)

#define PFP_TRY(expr) do { int rc_ = (expr); if (rc_ != PFP_OK) return rc_; } while (0)

// typed arena allocation; on exhaustion the enclosing function returns PFP_E_NOMEM
#define PFP_ALLOC_HI(ctx, ptr, T, count)                                                          \
    do { (ptr) = (T *)(ctx)->arena.alloc_hi(sizeof(T) * (size_t)((count) ? (count) : 1));         \
         if (!(ptr)) return PFP_E_NOMEM; } while (0)
#define PFP_ALLOC_LO(ctx, ptr, T, count)                                                          \
    do { (ptr) = (T *)(ctx)->arena.alloc_lo(sizeof(T) * (size_t)((count) ? (count) : 1));         \
         if (!(ptr)) return PFP_E_NOMEM; } while (0)

inline unsigned nblocks(uint64_t items, uint64_t per_block) { return (unsigned)((items + per_block - 1) / per_block); }
inline int bits_for(uint64_t maxval) { int b = 1; while (b < 64 && (maxval >> b)) ++b; return b; }

struct HostTimer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

} // namespace pfp
