// pfbwt-f_amd/csrc/pfbwt_hip.hip -- the C ABI of include/pfbwt_hip.h: host orchestration of the
// gfx950 kernels in parse.h / sufsort.h / emit.h / prims.h.  Built by hipcc into
// pfbwt-f_amd/lib/libpfbwt_hip.so.  There is no CPU fallback in this library.
#include "common.h"
#include "prims.h"
#include "parse.h"
#include "fasta.h"
#include "sufsort.h"
#include "recsort.h"
#include "dictrec.h"
#include "emit.h"
#include "markers.h"
#include <map>
#include <sched.h>
#include <thread>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

using namespace pfp;

#ifndef PFP_BACKEND_NAME
#define PFP_BACKEND_NAME "hip-gfx950"
#endif

// -------------------------------------------------------------------------------------------------
// The workspace is an ADDRESS range of `want` bytes (what the stages of a text of n_hint bytes could ask for, at most 15/16 of
// the card); HBM is committed only where the two ends of the arena really get to (csrc/devmem.h).
static int ensure_arena(pfp_ctx *c, uint64_t n_hint)
{
    size_t want = c->arena_request ? c->arena_request : (size_t)(96ULL * n_hint + (64ULL << 20));
    if (!c->arena_request) {   // never ask for more than the device can give (large inputs run with what there is)
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess) {
            const size_t avail = fr + c->arena.vm.committed;
            const size_t cap = avail - avail / 16;
            if (want > cap) want = cap;
        }
    }
    want &= ~(size_t)4095;     // both ends of the arena hand out 256-byte aligned blocks
    if (c->arena.base && c->arena.cap >= want) return PFP_OK;
    // a larger range for a context that already has one (the same context parsed a shard and now merges all of them): the new range
    // is reserved BEFORE the old one is given back, so that it never sits at the addresses kernels of this process used a moment ago
    // (seen on the MI355X box: memory access faults / garbage in the stages behind such a re-reservation at the same address)
    VmRegion nv;
    if (nv.reserve(want, c->device) != hipSuccess) { c->arena.want = want; return PFP_E_NOMEM; }
    if (c->arena.base) { PFP_HIP(c, hipStreamSynchronize(c->stream)); (void)hipDeviceSynchronize(); c->arena.vm.destroy(); c->arena.base = nullptr; c->arena.cap = 0; }
    c->arena.vm = nv; nv.base = nullptr;
    c->arena.base = c->arena.vm.base; c->arena.cap = want; c->arena.reset();
    return PFP_OK;
}

// The text buffer: 16 guard bytes + text + w Dollars + slack for whole trigger-scan tiles.  Its address range is reserved once
// (text_hint bytes if the caller announced a size with pfp_parse_reserve, else four times the first feed, doubled when outgrown) and
// committed as the text grows: feeding record by record rarely moves the text and never re-allocates piecemeal.
static int ensure_text(pfp_ctx *c, uint64_t need_n)
{
    const size_t need = 16 + (((size_t)need_n + 4095) / 4096) * 4096 + 4096 + 64;
    if (c->tb && c->tb_cap >= need) return PFP_OK;
    if (!c->text.live() || need > c->text.va_bytes) {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); tot = (size_t)1 << 38; }
        // An announced size (pfp_parse_reserve: the file reader knows the file's size) is reserved as it is.  Without one the range
        // is four times what the first feed needs (at least 64 MiB) and doubles when the text outgrows it: the text is then moved, a
        // device-to-device copy that costs less than 1 ms per GB and at most twice the text in total.  (Until round 3 every context
        // reserved the card's whole size and therefore committed HBM in 2 GiB pieces: a 100-byte text of a test created, mapped and
        // unmapped 2 GiB -- thousands of such contexts per process are what the rare host-side crashes inside pfp_destroy on the
        // GPU box point at.)
        size_t va = c->text_hint ? (size_t)c->text_hint + (size_t)c->text_hint / 8 + ((size_t)64 << 20) : (4 * need > ((size_t)64 << 20) ? 4 * need : ((size_t)64 << 20));
        if (!c->text_hint && c->text.live() && va < 2 * c->text.va_bytes) va = 2 * c->text.va_bytes;
        if (!c->text_hint && va > tot && tot > need + need / 4) va = tot;
        if (va < need) va = need + need / 4;
        VmRegion nr;
        if (nr.reserve(va, c->device) != hipSuccess) return PFP_E_NOMEM;
        if (c->tb) {      // the announced size was too small: move to a larger range (the one case in which text is copied)
            if (!nr.commit(0, 16 + (size_t)c->n)) { nr.destroy(); return PFP_E_NOMEM; }
            PFP_HIP(c, hipMemcpyAsync(nr.base, c->tb, 16 + (size_t)c->n, hipMemcpyDeviceToDevice, c->stream));
            PFP_HIP(c, hipStreamSynchronize(c->stream));
            c->text.destroy();
        }
        c->text = nr; nr.base = nullptr;
        c->tb = (uint8_t *)c->text.base; c->tb_cap = 0;
    }
    if (!c->text.commit(0, need)) return PFP_E_NOMEM;
    c->tb_cap = c->text.vmm ? (c->text.lo_edge < c->text.va_bytes ? c->text.lo_edge : c->text.va_bytes) : c->text.va_bytes;
    return PFP_OK;
}

static void reset_results(pfp_ctx *c)
{
    c->stage = 0; c->n = 0; c->nseq = 0; c->tb_n = 0; c->left_ctx = 0; c->view.src = nullptr; c->m = c->dwords = c->dsize = 0; c->nrows = 0; c->nout = c->runs = c->esa_pairs = 0;
    c->gsa_valid = false; c->d_wrank = nullptr; c->d_bwt = nullptr; c->d_sa = c->d_ssa = c->d_esa = nullptr;
    c->d_bwlast = nullptr; c->d_ilist = nullptr; c->d_bwsai = nullptr; c->d_bwl_il = nullptr;
    c->d_ma = nullptr; c->ma_words = 0; c->ma_lo_mark = (size_t)-1;
    c->d_ye = nullptr; c->d_pid = nullptr; c->d_parse = nullptr; c->d_last = nullptr; c->d_dict = nullptr; c->d_ws = nullptr; c->d_wordid = nullptr;
    c->d_occ = nullptr; c->d_sdict = nullptr; c->d_gsa = nullptr; c->d_grank = nullptr; c->d_srank = nullptr; c->d_sflag = nullptr;
    c->arena.reset();
    c->fa.started = false; c->fa.state = 2; c->fa.records = 0; c->fa.rec_raw.clear(); c->fa.rec_pos.clear();
}

// ---- route / tuning switches (pfbwt_hip_dev.h) -----------------------------------------------------
static const char *const tunable_names[] = {"verbose", "seg_grid", "seg_stage", "sort_k", "sort_no_table", "class_sort_maxrange", "dedup_table_log2", "no_trigger_table",
                                            "emit_chunk_rows", "fill_subs", "sample_cap", "no_runaware", "big_group_members", "force_wide_rows", "fasta_chunk_bytes", "ingest_block_bytes", "emit_group_rows", "no_slot_records", "dict_text_rounds", "int_key_symbols", "force_run_round",
                                            "ingest_readers", "expand_dma", "parse_rec", "parse_rec_p2", "parse_rec_min", "parse_rec_depth", "parse_rec_tile_rows", "parse_rec_table_log2", "dict_rec", "dict_rec_p2", "dedup_variant", "dedup_phases", "dedup_period", "dedup_chunk"};
static int set_tunable(pfp_ctx *c, const char *key, long long v)
{
    Tunables &t = c->tun;
    if (!strcmp(key, "verbose")) t.verbose = (int)v;
    else if (!strcmp(key, "seg_grid")) t.seg_grid = (int)v;
    else if (!strcmp(key, "seg_stage")) t.seg_stage = (int)v;
    else if (!strcmp(key, "sort_k")) t.sort_k = (int)v;
    else if (!strcmp(key, "sort_no_table")) t.sort_no_table = (int)v;
    else if (!strcmp(key, "class_sort_maxrange")) t.class_sort_maxrange = (uint32_t)v;
    else if (!strcmp(key, "dedup_table_log2")) t.dedup_table_log2 = (int)v;
    else if (!strcmp(key, "no_trigger_table")) t.no_trigger_table = (int)v;
    else if (!strcmp(key, "emit_chunk_rows")) t.emit_chunk_rows = v > 0 ? (uint64_t)v : (3ULL << 30);
    else if (!strcmp(key, "fill_subs")) t.fill_subs = (uint32_t)v;
    else if (!strcmp(key, "sample_cap")) t.sample_cap = v < 0 ? ~0ULL : (uint64_t)v;
    else if (!strcmp(key, "no_runaware")) t.no_runaware = (int)v;
    else if (!strcmp(key, "big_group_members")) t.big_group_members = (long)v;
    else if (!strcmp(key, "force_wide_rows")) t.force_wide_rows = (int)v;
    else if (!strcmp(key, "fasta_chunk_bytes")) t.fasta_chunk_bytes = v > 0 ? (uint64_t)v : 0;
    else if (!strcmp(key, "ingest_block_bytes")) t.ingest_block_bytes = v > 0 ? (uint64_t)v : 0;
    else if (!strcmp(key, "emit_group_rows")) t.emit_group_rows = v > 0 ? (uint32_t)v : 0u;
    else if (!strcmp(key, "no_slot_records")) t.no_slot_records = (int)v;
    else if (!strcmp(key, "dict_text_rounds")) t.dict_text_rounds = (int)v;
    else if (!strcmp(key, "int_key_symbols")) t.int_key_symbols = (int)v;
    else if (!strcmp(key, "force_run_round")) t.force_run_round = (int)v;
    else if (!strcmp(key, "ingest_readers")) t.ingest_readers = (int)v;
    else if (!strcmp(key, "expand_dma")) t.expand_dma = (int)v;
    else if (!strcmp(key, "parse_rec")) t.parse_rec = (int)v;
    else if (!strcmp(key, "parse_rec_p2")) t.parse_rec_p2 = (int)v;
    else if (!strcmp(key, "parse_rec_min")) t.parse_rec_min = v > 0 ? (uint64_t)v : 0;
    else if (!strcmp(key, "parse_rec_depth")) t.parse_rec_depth = (int)v;
    else if (!strcmp(key, "parse_rec_tile_rows")) t.parse_rec_tile_rows = v > 0 ? (uint32_t)v : 0u;
    else if (!strcmp(key, "parse_rec_table_log2")) t.parse_rec_table_log2 = (int)v;
    else if (!strcmp(key, "dict_rec")) t.dict_rec = (int)v;
    else if (!strcmp(key, "dict_rec_p2")) t.dict_rec_p2 = (int)v;
    else if (!strcmp(key, "dedup_variant")) t.dedup_variant = (int)v;
    else if (!strcmp(key, "dedup_phases")) t.dedup_phases = (int)v;
    else if (!strcmp(key, "dedup_period")) t.dedup_period = (int64_t)v;
    else if (!strcmp(key, "dedup_chunk")) t.dedup_chunk = (int64_t)v;
    else return PFP_E_ARG;
    return PFP_OK;
}
// PFP_VERBOSE is honoured always (it changes no route); every other PFP_<NAME> only in a process started with PFP_TEST_HOOKS=1,
// so that a user's environment cannot silently change routes or tile sizes
static void load_tunables_from_env(pfp_ctx *c)
{
    if (getenv("PFP_VERBOSE")) c->tun.verbose = 1;
    const char *hooks = getenv("PFP_TEST_HOOKS");
    if (!hooks || strcmp(hooks, "1")) return;
    for (const char *name : tunable_names) {
        char env[64] = "PFP_"; size_t k = 4;
        for (const char *q = name; *q && k + 1 < sizeof env; ++q) env[k++] = (char)((*q >= 'a' && *q <= 'z') ? *q - 32 : *q);
        env[k] = 0;
        const char *e = getenv(env);
        if (e) (void)set_tunable(c, name, *e ? atoll(e) : 1LL);
    }
}

extern "C" {

const char *pfp_backend(void) { return PFP_BACKEND_NAME; }
int pfp_debug_set(pfp_ctx *c, const char *key, long long value) { return (c && key) ? set_tunable(c, key, value) : PFP_E_ARG; }

const char *pfp_strerror(int s)
{
    switch (s) {
    case PFP_OK: return "ok";
    case PFP_E_ARG: return "invalid argument";
    case PFP_E_INVALID_CHAR: return "error, invalid character";
    case PFP_E_TOO_LARGE: return "input too long for 32-bit device indices";
    case PFP_E_NOMEM: return "device workspace exhausted";
    case PFP_E_HIP: return "HIP runtime error";
    case PFP_E_ONE_WORD: return "error: only one dict word total. Re-run with a smaller p modulus";
    case PFP_E_STATE: return "call order violated";
    case PFP_E_CORRUPT: return "something went wrong!";
    case PFP_E_IO: return "failed to open file!";
    default: return "unknown status";
    }
}

pfp_ctx *pfp_create(int w, uint64_t p, unsigned flags, int device, uint64_t workspace_bytes, int *status)
{
    int st = PFP_OK;
    pfp_ctx *c = nullptr;
    if (w < 1 || w > 32 || p == 0) st = PFP_E_ARG;           // check_w, pfparser.hpp:371-376
    else {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) st = PFP_E_HIP;
        else if (hipSetDevice(device) != hipSuccess) st = PFP_E_HIP;
        else {
            c = new pfp_ctx();
            c->w = w; c->p = p; c->flags = flags; c->device = device; c->arena_request = (size_t)workspace_bytes;
            load_tunables_from_env(c);
            if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; c = nullptr; st = PFP_E_HIP; }
        }
    }
    if (status) *status = st;
    return c;
}

void pfp_destroy(pfp_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    (void)hipDeviceSynchronize();      // other contexts may still read this one's text / shard view on their streams: unmapping does not wait for them
    prof_collect(c);
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    c->text.destroy();
    for (int k = 0; k < 2; ++k) { if (c->fa.raw[k]) (void)hipFree(c->fa.raw[k]); if (c->fa.ev_copied[k]) (void)hipEventDestroy(c->fa.ev_copied[k]); if (c->fa.ev_free[k]) (void)hipEventDestroy(c->fa.ev_free[k]); }
    if (c->fa.tiles) (void)hipFree(c->fa.tiles);
    if (c->fa.d_tot) (void)hipFree(c->fa.d_tot);
    if (c->fa.h_tot) (void)hipHostFree(c->fa.h_tot);
    if (c->fa.copy_ready) (void)hipStreamDestroy(c->fa.copy);
    for (auto &q : c->ing_buf) if (q) (void)hipHostFree(q);
    if (c->d_trigtab) (void)hipFree(c->d_trigtab);
    for (int k = 0; k < 2; ++k) { if (c->hstage[k]) (void)hipHostFree(c->hstage[k]); if (c->hstage_ev[k]) (void)hipEventDestroy(c->hstage_ev[k]); }
    c->arena.vm.destroy();
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int pfp_error_detail(pfp_ctx *c, uint64_t *pos, int *ch)
{
    if (!c) return PFP_E_ARG;
    if (pos) *pos = c->err_pos;
    if (ch) *ch = c->err_ch;
    return PFP_OK;
}
uint64_t pfp_workspace_needed(pfp_ctx *c) { return c ? (uint64_t)(c->arena.cap + c->arena.want) : 0; }
int pfp_reset(pfp_ctx *c)
{
    if (!c) return PFP_E_ARG;
    PFP_HIP(c, hipSetDevice(c->device));
    PFP_HIP(c, hipStreamSynchronize(c->stream));
    reset_results(c);
    return PFP_OK;
}

int pfp_profile_enable(pfp_ctx *c, int on) { if (!c) return PFP_E_ARG; prof_collect(c); c->prof_on = on != 0; c->prof_mask = ~0ULL; return PFP_OK; }
int pfp_profile_select(pfp_ctx *c, const char *kernel)
{
    if (!c || !kernel) return PFP_E_ARG;
    prof_collect(c);
    for (int i = 0; i < K_COUNT_; ++i) if (!strcmp(kernel, kernel_names[i])) { c->prof_on = true; c->prof_mask = 1ULL << i; return PFP_OK; }
    return PFP_E_ARG;
}
int pfp_profile_reset(pfp_ctx *c) { if (!c) return PFP_E_ARG; prof_collect(c); for (auto &r : c->prof) r = ProfRec(); return PFP_OK; }
int pfp_profile_get(pfp_ctx *c, int idx, const char **name, uint64_t *launches, double *ms, double *bytes)
{
    if (!c || idx < 0 || idx >= K_COUNT_) return PFP_E_ARG;
    prof_collect(c);
    if (name) *name = kernel_names[idx];
    if (launches) *launches = c->prof[idx].launches;
    if (ms) *ms = c->prof[idx].ms;
    if (bytes) *bytes = c->prof[idx].bytes;
    return PFP_OK;
}
int pfp_stage_ms(pfp_ctx *c, double out[3]) { if (!c || !out) return PFP_E_ARG; for (int i = 0; i < 3; ++i) out[i] = c->stage_ms[i]; return PFP_OK; }

// ---- stage 1: feeding ---------------------------------------------------------------------------
constexpr size_t STAGE_BYTES = (size_t)32 << 20;
static bool host_pointer_is_pinned(const void *p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) == hipSuccess) return a.type == hipMemoryTypeHost;
    (void)hipGetLastError();      // an ordinary malloc'd pointer is reported as an error
    return false;
}
// host memory -> device: pinned sources directly, pageable ones through the staging ring (kseq's 16 KiB reads of
// include/kseq.h:228 become 32 MiB DMA transfers; the memcpy into one buffer overlaps the transfer of the other)
// *in_flight (nullable) = the caller's buffer may still be read when this returns (pinned source: the caller must wait for
// the stream); copies through the staging ring are complete as far as the caller's buffer is concerned.
static int h2d_copy(pfp_ctx *c, uint8_t *dst, const uint8_t *src, uint64_t len, bool *in_flight = nullptr)
{
    if (in_flight) *in_flight = false;
    if (!len) return PFP_OK;
    if (host_pointer_is_pinned(src)) { PFP_HIP(c, hipMemcpyAsync(dst, src, (size_t)len, hipMemcpyHostToDevice, c->stream)); if (in_flight) *in_flight = true; else PFP_HIP(c, hipStreamSynchronize(c->stream)); return PFP_OK; }
    if (len <= ((size_t)1 << 16)) { PFP_HIP(c, hipMemcpyAsync(dst, src, (size_t)len, hipMemcpyHostToDevice, c->stream)); PFP_HIP(c, hipStreamSynchronize(c->stream)); return PFP_OK; }
    for (int k = 0; k < 2; ++k) if (!c->hstage[k]) {
        PFP_HIP(c, hipHostMalloc((void **)&c->hstage[k], STAGE_BYTES, hipHostMallocDefault));
        PFP_HIP(c, hipEventCreate(&c->hstage_ev[k]));
    }
    int k = 0;
    for (uint64_t off = 0; off < len; off += STAGE_BYTES, k ^= 1) {
        const size_t chunk = (size_t)(len - off < STAGE_BYTES ? len - off : STAGE_BYTES);
        if (c->hstage_used[k]) PFP_HIP(c, hipEventSynchronize(c->hstage_ev[k]));      // the transfer that last used this buffer is done
        memcpy(c->hstage[k], src + off, chunk);
        PFP_HIP(c, hipMemcpyAsync(dst + off, c->hstage[k], chunk, hipMemcpyHostToDevice, c->stream));
        PFP_HIP(c, hipEventRecord(c->hstage_ev[k], c->stream));
        c->hstage_used[k] = true;
    }
    return PFP_OK;
}
static int feed_common(pfp_ctx *c, const void *src, uint64_t len, int end_of_seq, hipMemcpyKind kind)
{
    if (!c || (!src && len)) return PFP_E_ARG;
    PFP_HIP(c, hipSetDevice(c->device));
    if (c->stage != 0) reset_results(c);
    const uint64_t add = len + (end_of_seq ? (uint64_t)c->w : 0);
    // pfparser.hpp:326-331: the 32-bit build stops at 2^32 bases; the 64-bit build here at 2^40 (device positions are 64-bit)
    if (c->n + add + (uint64_t)c->w + 64 >= ((c->flags & PFP_FLAG_U64) ? (1ULL << 40) : 0xFFFFFFFFULL)) return PFP_E_TOO_LARGE;
    PFP_TRY(ensure_text(c, c->n + add + (uint64_t)c->w));
    bool in_flight = false;
    if (len && kind == hipMemcpyHostToDevice) PFP_TRY(h2d_copy(c, c->tb + 16 + c->n, (const uint8_t *)src, len, &in_flight));
    else if (len) PFP_HIP(c, hipMemcpyAsync(c->tb + 16 + c->n, src, (size_t)len, kind, c->stream));
    c->n += len;
    if (end_of_seq) {   // the w 'A's of pfparser.hpp:335-337
        ++c->nseq;
        PFP_HIP(c, hipMemsetAsync(c->tb + 16 + c->n, 'A', (size_t)c->w, c->stream));
        c->n += (uint64_t)c->w;
    }
    // the caller may reuse its buffer on return: a pageable source went through the staging ring (its DMA transfers overlap
    // the caller's next read / decompression, include/kseq.h:228), only a page-locked source is read in place
    // (a device source may belong to another context that its owner destroys right after this call -- PfParser::operator+= with a
    // temporary right-hand side, src/merge_pfp.cpp:100-112: unmapping on-demand committed memory does not wait for copies in flight)
    if (in_flight || (len && kind == hipMemcpyDeviceToDevice)) PFP_HIP(c, hipStreamSynchronize(c->stream));
    c->tb_n = c->n;
    return PFP_OK;
}
int pfp_parse_reopen(pfp_ctx *c)
{
    if (!c) return PFP_E_ARG;
    PFP_HIP(c, hipSetDevice(c->device));
    if (c->stage == 0) return PFP_OK;
    if (!c->tb || !c->tb_n || c->tb_n != c->n) return PFP_E_STATE;   // a context filled by pfp_merge_shards / pfp_bwt_load holds no text
    PFP_HIP(c, hipStreamSynchronize(c->stream));
    const uint64_t n = c->n, nseq = c->nseq;
    reset_results(c);
    c->nseq = nseq;
    c->n = c->tb_n = n;                                         // the (normalised) text is still in place; more can be appended
    return PFP_OK;
}
static int flush_view(pfp_ctx *c);
int pfp_text_view(pfp_ctx *c, const uint8_t **d_text, uint64_t *n)
{
    if (!c || !d_text || !n) return PFP_E_ARG;
    PFP_TRY(flush_view(c));
    if (!c->tb || c->tb_n != c->n) { *d_text = nullptr; *n = 0; return PFP_OK; }
    PFP_HIP(c, hipSetDevice(c->device));
    PFP_HIP(c, hipStreamSynchronize(c->stream));
    *d_text = c->tb + 16; *n = c->n;
    return PFP_OK;
}
int pfp_parse_feed_left_context(pfp_ctx *c)
{
    if (!c) return PFP_E_ARG;
    if (c->stage != 0) { PFP_HIP(c, hipSetDevice(c->device)); reset_results(c); }
    if (c->n != 0) return PFP_E_STATE;                            // the context must be the first thing a shard is fed
    const std::vector<uint8_t> a((size_t)c->w, (uint8_t)'A');
    PFP_TRY(feed_common(c, a.data(), a.size(), 0, hipMemcpyHostToDevice));
    c->left_ctx = (uint64_t)c->w;
    return PFP_OK;
}
// a pending row view (pfp_parse_feed_device_view) becomes text in X: what every entry point that appends to or hands out the text
// does first (the fused scan of pfp_parse_finalize is the one consumer that does not need it)
static int flush_view(pfp_ctx *c)
{
    if (!c->view.src) return PFP_OK;
    const uint64_t count = c->view.count, len = c->view.len;
    const uint64_t blocks_per_row = ((len + 15) / 16 + BLOCK - 1) / BLOCK;
    PFP_LAUNCH(c, K_MISC, 2 * count * len, k_feed_batch, count * blocks_per_row, c->view.src, count, len, c->view.stride, c->w, c->tb + 16);
    c->view.src = nullptr;
    return PFP_OK;
}
int pfp_parse_feed_device_view(pfp_ctx *c, const void *d_bases, uint64_t count, uint64_t len, uint64_t stride)
{
    if (!c || !d_bases || !count || !len || stride < len) return PFP_E_ARG;
    PFP_HIP(c, hipSetDevice(c->device));
    if (c->stage != 0) reset_results(c);
    if (c->n != 0) return PFP_E_STATE;                                   // the view is the whole text of a parse
    const uint64_t pitch = len + (uint64_t)c->w, add = count * pitch;
    if (add + (uint64_t)c->w + 64 >= ((c->flags & PFP_FLAG_U64) ? (1ULL << 40) : 0xFFFFFFFFULL)) return PFP_E_TOO_LARGE;
    const uint64_t blocks_per_row = ((len + 15) / 16 + BLOCK - 1) / BLOCK;
    if (count * blocks_per_row >= 0x7FFFFFFFULL) return PFP_E_TOO_LARGE;
    PFP_TRY(ensure_text(c, add + (uint64_t)c->w));
    c->view.src = (const uint8_t *)d_bases; c->view.count = count; c->view.len = len; c->view.stride = stride;
    c->n = add; c->tb_n = c->n; c->nseq = count;
    return PFP_OK;
}
int pfp_parse_feed(pfp_ctx *c, const uint8_t *bases, uint64_t len, int end_of_seq) { if (c) PFP_TRY(flush_view(c)); return feed_common(c, bases, len, end_of_seq, hipMemcpyHostToDevice); }
int pfp_parse_feed_device(pfp_ctx *c, const void *d_bases, uint64_t len, int end_of_seq) { if (c) PFP_TRY(flush_view(c)); return feed_common(c, d_bases, len, end_of_seq, hipMemcpyDeviceToDevice); }
int pfp_parse_feed_device_batch(pfp_ctx *c, const void *d_bases, uint64_t count, uint64_t len, uint64_t stride)
{
    if (!c || (!d_bases && count && len) || stride < len) return PFP_E_ARG;
    if (!count) return PFP_OK;
    PFP_TRY(flush_view(c));
    if (!len) { for (uint64_t k = 0; k < count; ++k) PFP_TRY(feed_common(c, nullptr, 0, 1, hipMemcpyDeviceToDevice)); return PFP_OK; }
    PFP_HIP(c, hipSetDevice(c->device));
    if (c->stage != 0) reset_results(c);
    const uint64_t pitch = len + (uint64_t)c->w, add = count * pitch;
    if (c->n + add + (uint64_t)c->w + 64 >= ((c->flags & PFP_FLAG_U64) ? (1ULL << 40) : 0xFFFFFFFFULL)) return PFP_E_TOO_LARGE;
    PFP_TRY(ensure_text(c, c->n + add + (uint64_t)c->w));
    uint8_t *dst = c->tb + 16 + c->n;
    const uint64_t blocks_per_row = ((len + 15) / 16 + BLOCK - 1) / BLOCK;
    if (count * blocks_per_row >= 0x7FFFFFFFULL) return PFP_E_TOO_LARGE;      // grid limit (2^31 workgroups = 8 Tbase)
    PFP_LAUNCH(c, K_MISC, 2 * count * len, k_feed_batch, count * blocks_per_row, (const uint8_t *)d_bases, count, len, stride, c->w, dst);
    c->n += add; c->tb_n = c->n; c->nseq += count;
    return PFP_OK;
}

int pfp_parse_feed_batch(pfp_ctx *c, const uint8_t *bases, uint64_t count, uint64_t len, uint64_t stride)
{
    if (!c || (!bases && count && len) || stride < len) return PFP_E_ARG;
    if (!count) return PFP_OK;
    PFP_TRY(flush_view(c));
    if (!len || !host_pointer_is_pinned(bases)) {      // pageable memory: record by record through the staging ring
        for (uint64_t k = 0; k < count; ++k) PFP_TRY(feed_common(c, bases + k * stride, len, 1, hipMemcpyHostToDevice));
        return PFP_OK;
    }
    PFP_HIP(c, hipSetDevice(c->device));
    if (c->stage != 0) reset_results(c);
    const uint64_t pitch = len + (uint64_t)c->w, add = count * pitch;
    if (c->n + add + (uint64_t)c->w + 64 >= ((c->flags & PFP_FLAG_U64) ? (1ULL << 40) : 0xFFFFFFFFULL)) return PFP_E_TOO_LARGE;
    PFP_TRY(ensure_text(c, c->n + add + (uint64_t)c->w));
    uint8_t *dst = c->tb + 16 + c->n;
    // pinned host memory: one DMA transfer per record, queued back to back (the runtime's 2-D copy rejects a destination that
    // spans several pieces of the on-demand committed text, csrc/devmem.h), the pads of pfparser.hpp:335-337 by one small kernel
    for (uint64_t k = 0; k < count; ++k) PFP_HIP(c, hipMemcpyAsync(dst + k * pitch, bases + k * stride, (size_t)len, hipMemcpyHostToDevice, c->stream));
    PFP_LAUNCH(c, K_MISC, count * (uint64_t)c->w, k_pad_rows, nblocks(count * (uint64_t)c->w, BLOCK), dst, count, len, pitch, c->w);
    PFP_HIP(c, hipStreamSynchronize(c->stream));      // the caller may reuse its buffer
    c->n += add; c->tb_n = c->n; c->nseq += count;
    return PFP_OK;
}

// ---- raw FASTA bytes (csrc/fasta.h) ---------------------------------------------------------------------------------
constexpr size_t FA_RAW_MAX = (size_t)256 << 20, FA_RAW_MIN = (size_t)1 << 20;
int pfp_parse_reserve(pfp_ctx *c, uint64_t text_bytes)
{
    if (!c) return PFP_E_ARG;
    if (!c->text.live()) c->text_hint = text_bytes;      // sizes the address range of the text; ignored once text has been fed
    return PFP_OK;
}
static int ensure_copy_stream(pfp_ctx *c)
{
    auto &f = c->fa;
    if (f.copy_ready) return PFP_OK;
    PFP_HIP(c, hipStreamCreateWithFlags(&f.copy, hipStreamNonBlocking));
    f.copy_ready = true;
    for (int k = 0; k < 2; ++k) { PFP_HIP(c, hipEventCreateWithFlags(&f.ev_copied[k], hipEventDisableTiming)); PFP_HIP(c, hipEventCreateWithFlags(&f.ev_free[k], hipEventDisableTiming)); }
    PFP_HIP(c, hipMalloc((void **)&f.d_tot, 64));
    PFP_HIP(c, hipHostMalloc((void **)&f.h_tot, 64, hipHostMallocDefault));
    PFP_HIP(c, hipMemsetAsync(f.d_tot, 0, 64, c->stream));
    return PFP_OK;
}
static int fasta_buffers(pfp_ctx *c, uint64_t len)
{
    auto &f = c->fa;
    size_t want = FA_RAW_MIN; while (want < FA_RAW_MAX && want < len) want <<= 1;
    if (c->tun.fasta_chunk_bytes) want = (size_t)(c->tun.fasta_chunk_bytes < 64 ? 64 : c->tun.fasta_chunk_bytes);      // tests: many chunks on small inputs
    PFP_TRY(ensure_copy_stream(c));
    if (f.rawcap >= want || (f.rawcap && want < 4 * f.rawcap)) return PFP_OK;      // (a much larger call than the first one: take larger buffers once)
    PFP_HIP(c, hipStreamSynchronize(c->stream)); PFP_HIP(c, hipStreamSynchronize(f.copy));
    for (int k = 0; k < 2; ++k) { if (f.raw[k]) PFP_HIP(c, hipFree(f.raw[k])); f.raw[k] = nullptr; f.used[k] = false; }
    if (f.tiles) { PFP_HIP(c, hipFree(f.tiles)); f.tiles = nullptr; }
    for (int k = 0; k < 2; ++k) if (hipMalloc((void **)&f.raw[k], want) != hipSuccess) { (void)hipGetLastError(); f.rawcap = 0; return PFP_E_NOMEM; }
    const size_t T = want / FA_TILE + 1;
    f.tiles_cap = T * (1 + 12 + 12 + 1 + 8 + 4) + 256;
    if (hipMalloc((void **)&f.tiles, f.tiles_cap) != hipSuccess) { (void)hipGetLastError(); f.rawcap = 0; return PFP_E_NOMEM; }
    f.rawcap = want;
    return PFP_OK;
}
// kseq_read, include/kseq.h:186-190: everything in front of the first '>' or '@' of a stream is skipped.  Returns the offset
// of the first byte that counts (len: none yet).
static uint64_t fa_skip_preamble(pfp_ctx *c, const uint8_t *raw, uint64_t len)
{
    auto &f = c->fa;
    if (f.started || !len) return 0;
    const uint8_t *a = (const uint8_t *)memchr(raw, '>', (size_t)len), *b = (const uint8_t *)memchr(raw, '@', (size_t)len);
    const uint8_t *q = (a && b) ? (a < b ? a : b) : (a ? a : b);
    if (!q) return len;
    f.started = true; f.state = FA_L; f.records = 0;
    return (uint64_t)(q - raw);
}
// upload of one raw piece (at most rawcap bytes) into device buffer `slot`, once the kernels that read its previous content are done
static int fa_issue(pfp_ctx *c, const uint8_t *src, uint64_t len, int slot)
{
    auto &f = c->fa;
    if (f.used[slot]) PFP_HIP(c, hipStreamWaitEvent(f.copy, f.ev_free[slot], 0));
    PFP_HIP(c, hipMemcpyAsync(f.raw[slot], src, (size_t)len, hipMemcpyHostToDevice, f.copy));
    PFP_HIP(c, hipEventRecord(f.ev_copied[slot], f.copy));
    return PFP_OK;
}
// strip the piece in device buffer `slot` into the text.  Returns with the upload of the piece complete (the host waits for
// the piece's totals, which wait for the upload): the caller's buffer may be reused.  rec_base: offset of the piece in the
// caller's coordinate system (record offsets are reported in it).
static int fa_process(pfp_ctx *c, uint64_t len, int slot, bool want_recs, uint64_t rec_base, uint64_t *started)
{
    auto &f = c->fa;
    const uint64_t w = (uint64_t)c->w;
    const uint64_t limit = (c->flags & PFP_FLAG_U64) ? (1ULL << 40) : 0xFFFFFFFFULL;
    PFP_HIP(c, hipStreamWaitEvent(c->stream, f.ev_copied[slot], 0));
    FaTiles t; t.T = (len + FA_TILE - 1) / FA_TILE;
    uint8_t *q = f.tiles;
    t.kbase = (unsigned long long *)q; q += t.T * 8; t.kept = (uint32_t *)q; q += t.T * 12; t.hdr = (uint32_t *)q; q += t.T * 12; t.hbase = (uint32_t *)q; q += t.T * 4; t.func = q; q += t.T; t.st = q;
    PFP_LAUNCH(c, K_FASTA, len, k_fa_scan, t.T, (const uint8_t *)f.raw[slot], len, t);
    PFP_LAUNCH(c, K_MISC, t.T * 40, k_fa_spine, 1, t, f.state, f.d_tot);
    PFP_HIP(c, hipMemcpyAsync(f.h_tot, f.d_tot, 24, hipMemcpyDeviceToHost, c->stream));
    PFP_HIP(c, hipStreamSynchronize(c->stream));
    const uint64_t kept = f.h_tot[0], hdr = f.h_tot[1], rec0 = f.records;
    const uint64_t tbase = c->n - w * (rec0 ? rec0 - 1 : 0);
    const uint64_t n_new = tbase + kept + w * ((rec0 + hdr) ? rec0 + hdr - 1 : 0);
    if (n_new + 2 * w + 64 >= limit) return PFP_E_TOO_LARGE;                      // pfparser.hpp:326-331
    PFP_TRY(ensure_text(c, n_new + 2 * w));
    uint64_t *d_rr = nullptr, *d_rp = nullptr;
    struct DevFree { uint64_t *&p; ~DevFree() { if (p) (void)hipFree(p); } } rr_guard{d_rr};      // freed on every path out of this function
    if (want_recs && hdr) { PFP_HIP(c, hipMalloc((void **)&d_rr, hdr * 16)); d_rp = d_rr + hdr; }
    PFP_LAUNCH(c, K_FASTA, 2 * len, k_fa_compact, t.T, (const uint8_t *)f.raw[slot], len, t, c->tb + 16, tbase, rec0, c->w, d_rr, d_rp, (uint32_t *)(f.d_tot + 4));
    PFP_HIP(c, hipEventRecord(f.ev_free[slot], c->stream)); f.used[slot] = true;
    if (d_rr) {
        const size_t b = f.rec_raw.size();
        f.rec_raw.resize(b + hdr); f.rec_pos.resize(b + hdr);
        PFP_HIP(c, hipMemcpyAsync(f.rec_raw.data() + b, d_rr, hdr * 8, hipMemcpyDeviceToHost, c->stream));      // (the context's stream does not synchronise with the null stream)
        PFP_HIP(c, hipMemcpyAsync(f.rec_pos.data() + b, d_rp, hdr * 8, hipMemcpyDeviceToHost, c->stream));
        PFP_HIP(c, hipStreamSynchronize(c->stream));
        for (size_t i = b; i < b + hdr; ++i) f.rec_raw[i] += rec_base;
    }
    c->n = n_new; c->tb_n = c->n; f.records = rec0 + hdr; f.state = (uint32_t)f.h_tot[2];
    if (started) *started += hdr;
    return PFP_OK;
}
// end of a stream: the pad of its last record; a '+' line seen on the way makes the whole stream PFP_E_ARG
static int fa_finish_stream(pfp_ctx *c)
{
    auto &f = c->fa;
    const uint64_t w = (uint64_t)c->w;
    if (f.started && f.records) {
        PFP_TRY(ensure_text(c, c->n + 2 * w));
        PFP_HIP(c, hipMemsetAsync(c->tb + 16 + c->n, 'A', (size_t)w, c->stream));
        c->n += w; c->tb_n = c->n;
    }
    uint32_t fl = 0;
    if (f.d_tot) { PFP_HIP(c, hipMemcpyAsync(&fl, f.d_tot + 4, 4, hipMemcpyDeviceToHost, c->stream)); PFP_HIP(c, hipStreamSynchronize(c->stream)); PFP_HIP(c, hipMemsetAsync(f.d_tot + 4, 0, 8, c->stream)); }
    c->nseq += f.records;      // (the hint the text de-duplication orders its workgroups by)
    f.started = false; f.state = FA_L; f.records = 0;
    if (fl & 1u) { c->err_ch = '+'; return PFP_E_ARG; }                             // a FASTQ quality section: not handled on the device
    return PFP_OK;
}
static int feed_fasta_pieces(pfp_ctx *c, const uint8_t *raw, uint64_t len, unsigned flags, uint64_t *nrec);
// The header promises that the caller's buffer may be reused (freed, unregistered) when the call returns.  On the good path every
// piece's upload has completed by then (fa_process waits for the piece's totals, which wait for its upload); on an error return the
// upload of piece k + 1 may still be reading the buffer -- PFP_E_TOO_LARGE / PFP_E_NOMEM of piece k come after it was issued --
// so every error path drains both streams first.
static int feed_fasta_impl(pfp_ctx *c, const uint8_t *raw, uint64_t len, unsigned flags, uint64_t *nrec)
{
    PFP_TRY(flush_view(c));
    const int rc = feed_fasta_pieces(c, raw, len, flags, nrec);
    if (rc != PFP_OK) { if (c->fa.copy_ready) (void)hipStreamSynchronize(c->fa.copy); (void)hipStreamSynchronize(c->stream); }
    return rc;
}
static int feed_fasta_pieces(pfp_ctx *c, const uint8_t *raw, uint64_t len, unsigned flags, uint64_t *nrec)
{
    auto &f = c->fa;
    const bool want_recs = (flags & PFP_FASTA_RECORDS) != 0;
    f.rec_raw.clear(); f.rec_pos.clear();
    uint64_t started = 0;
    const uint64_t off = fa_skip_preamble(c, raw, len);
    if (off < len) {
        PFP_TRY(fasta_buffers(c, len - off));
        const uint64_t cap = f.rawcap, nsub = (len - off + cap - 1) / cap;
        auto piece = [&](uint64_t k, uint64_t *o, uint64_t *cl) { *o = off + k * cap; *cl = (len - *o < cap) ? len - *o : cap; };
        uint64_t o, cl;
        piece(0, &o, &cl); PFP_TRY(fa_issue(c, raw + o, cl, 0));
        for (uint64_t k = 0; k < nsub; ++k) {
            if (k + 1 < nsub) { piece(k + 1, &o, &cl); PFP_TRY(fa_issue(c, raw + o, cl, (int)((k + 1) & 1))); }      // its upload overlaps the stripping of piece k
            piece(k, &o, &cl);
            PFP_TRY(fa_process(c, cl, (int)(k & 1), want_recs, o, &started));
        }
    }
    if (flags & PFP_FASTA_FINAL) PFP_TRY(fa_finish_stream(c));
    if (nrec) *nrec = started;
    return PFP_OK;
}
int pfp_parse_feed_fasta(pfp_ctx *c, const uint8_t *raw, uint64_t len, unsigned flags, uint64_t *nrec)
{
    if (!c || (!raw && len)) return PFP_E_ARG;
    PFP_HIP(c, hipSetDevice(c->device));
    if (c->stage != 0) reset_results(c);
    return feed_fasta_impl(c, raw, len, flags, nrec);
}
} // extern "C"
#include "ingest.h"
extern "C" {
int pfp_parse_feed_fasta_file(pfp_ctx *c, const char *path, unsigned flags, pfp_ingest_info *info)
{
    if (!c || !path) return PFP_E_ARG;
    PFP_HIP(c, hipSetDevice(c->device));
    if (c->stage != 0) reset_results(c);
    IngestStats st;
    const int rc = ingest_file(c, path, flags, &st);
    if (info) { info->raw_bytes = st.raw_bytes; info->records = st.records; info->n = c->n; info->read_wait_ms = st.read_wait_ms; info->total_ms = st.total_ms; info->mode = st.mode; }
    return rc;
}
int pfp_bwt_write(pfp_ctx *c, int fd_bwt, int fd_sa, int fd_ssa, int fd_esa)
{
    if (!c) return PFP_E_ARG;
    if (c->stage < 3) return PFP_E_STATE;
    PFP_HIP(c, hipSetDevice(c->device));
    PFP_HIP(c, hipStreamSynchronize(c->stream));
    const size_t U = (c->flags & PFP_FLAG_U64) ? 8 : 4;
    if (fd_bwt >= 0) PFP_TRY(write_device_to_fd(c, c->d_bwt, c->slice_rows, fd_bwt));
    if (fd_sa >= 0) { if (!c->d_sa) return PFP_E_STATE; PFP_TRY(write_device_to_fd(c, c->d_sa, c->slice_rows * U, fd_sa)); }
    if (fd_ssa >= 0) { if (!c->d_ssa) return PFP_E_STATE; PFP_TRY(write_device_to_fd(c, c->d_ssa, c->runs * 2 * U, fd_ssa)); }
    if (fd_esa >= 0) { if (!c->d_esa) return PFP_E_STATE; PFP_TRY(write_device_to_fd(c, c->d_esa, c->esa_pairs * 2 * U, fd_esa)); }
    return PFP_OK;
}
// .bwt over the slow link in its run-length form: with the run samples at hand, .ssa[k].row starts run k, so ONE byte per run (r bytes:
// 84 MB on S-32G) crosses PCIe instead of n + 1 (32 GB), and host threads write the runs out (memset) at memory bandwidth.
extern "C++" {
template <typename SAT> __global__ __launch_bounds__(BLOCK) void k_run_heads(const uint8_t *bwt, const SAT *ssa, uint64_t r, uint8_t *heads)
{
    const uint64_t k = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (k < r) heads[k] = bwt[(uint64_t)ssa[2 * k]];
}
}
// The runs of one block of the .bwt into host memory.  A run is a few hundred bytes on a pangenome.  Written run by run, every run
// touches two partial cache lines -- with ordinary stores every line of the 32 GB output is first READ from memory to be owned, with
// streaming stores a partially written line costs the memory controller a read-modify-write (measured on the pool's boxes: 130 ns per
// run and thread, 1.5 GB/s per thread).  So the runs are expanded into a 4 KiB buffer that lives in L1 and leave it as whole, 64-byte
// aligned lines of streaming stores; a run longer than the buffer goes out directly.
struct RunWriter {
    uint8_t *dst;                 // next byte of the destination that has not been written
    alignas(64) uint8_t buf[4096]; size_t fill = 0;
    explicit RunWriter(uint8_t *d) : dst(d) {}
    static void stream(uint8_t *d, const uint8_t *s, size_t len)      // d 16-byte aligned, len a multiple of 16
    {
#if defined(__SSE2__)
        for (size_t i = 0; i < len; i += 16) _mm_stream_si128(reinterpret_cast<__m128i *>(d + i), _mm_load_si128(reinterpret_cast<const __m128i *>(s + i)));
#else
        memcpy(d, s, len);
#endif
    }
    void flush_all()              // everything buffered goes out (the tail that is not a whole line with ordinary stores)
    {
        const size_t body = fill & ~(size_t)63;
        if (((uintptr_t)dst & 63u) == 0) { stream(dst, buf, body); memcpy(dst + body, buf + body, fill - body); }
        else memcpy(dst, buf, fill);
        dst += fill; fill = 0;
    }
    void put(uint8_t v, size_t len)
    {
        if (fill == 0 && ((uintptr_t)dst & 63u)) {      // bring the destination to a line boundary first
            const size_t head = (size_t)(-(uintptr_t)dst & 63u) < len ? (size_t)(-(uintptr_t)dst & 63u) : len;
            memset(dst, v, head); dst += head; len -= head;
        }
        while (len) {
            if (fill == 0 && len >= sizeof buf) {      // a long run: whole lines straight to the destination
                const size_t body = len & ~(size_t)63;
#if defined(__SSE2__)
                const __m128i x = _mm_set1_epi8((char)v);
                for (size_t i = 0; i < body; i += 16) _mm_stream_si128(reinterpret_cast<__m128i *>(dst + i), x);
#else
                memset(dst, v, body);
#endif
                dst += body; len -= body;
                continue;
            }
            const size_t k = len < sizeof buf - fill ? len : sizeof buf - fill;
            memset(buf + fill, v, k); fill += k; len -= k;
            if (fill == sizeof buf) { stream(dst, buf, sizeof buf); dst += sizeof buf; fill = 0; }
        }
    }
};
int pfp_bwt_get_expanded(pfp_ctx *c, uint8_t *host_bwt, const void *ssa_host, int threads)
{
    if (!c || !host_bwt) return PFP_E_ARG;
    if (c->stage < 3 || !c->d_ssa || !c->d_bwt || c->slice_rows != c->nout || c->slice_begin != 0) return PFP_E_STATE;      // needs the whole output's run samples
    PFP_HIP(c, hipSetDevice(c->device));
    const bool u64 = (c->flags & PFP_FLAG_U64) != 0;
    const uint64_t r = c->runs, nout = c->nout;
    const size_t mk = c->arena.mark_hi();
    uint8_t *d_heads; PFP_ALLOC_HI(c, d_heads, uint8_t, r);
    if (u64) PFP_LAUNCH(c, K_MISC, r * 9, (k_run_heads<uint64_t>), nblocks(r, BLOCK), (const uint8_t *)c->d_bwt, (const uint64_t *)c->d_ssa, r, d_heads);
    else PFP_LAUNCH(c, K_MISC, r * 5, (k_run_heads<uint32_t>), nblocks(r, BLOCK), (const uint8_t *)c->d_bwt, (const uint32_t *)c->d_ssa, r, d_heads);
    std::vector<uint8_t> heads((size_t)r);
    PFP_HIP(c, hipMemcpyAsync(heads.data(), d_heads, (size_t)r, hipMemcpyDeviceToHost, c->stream));
    std::vector<uint8_t> own;
    if (!ssa_host) { own.resize((size_t)r * 2 * (u64 ? 8 : 4)); PFP_HIP(c, hipMemcpyAsync(own.data(), c->d_ssa, own.size(), hipMemcpyDeviceToHost, c->stream)); ssa_host = own.data(); }
    PFP_HIP(c, hipStreamSynchronize(c->stream));
    c->arena.release_hi(mk);
    if (threads < 1) threads = usable_cpus();      // 0 / negative: one thread per CPU this process may use
    if (threads > 64) threads = 64;
    auto row = [&](uint64_t k) -> uint64_t { return k >= r ? nout : (u64 ? ((const uint64_t *)ssa_host)[2 * k] : (uint64_t)((const uint32_t *)ssa_host)[2 * k]); };
    // The output is cut into blocks of BYTES (runs of a collection's BWT are anything from 1 to millions of rows long; equal run counts
    // gave threads unequal shares -- VERDICT r3).  Host threads claim blocks from the FRONT and write the runs that reach into them
    // (the first one is found by bisection over the run starts).  When the destination is page-locked, the copy engine claims
    // blocks from the BACK and moves those rows over PCIe as they are: host stores and DMA writes use different paths into host memory
    // (16 cores of two CCDs reach ~65 GB/s of streaming stores on the pool's boxes, the link 55 GB/s), so the two meet somewhere in the
    // middle -- wherever this box's ratio puts it -- instead of the slower one doing everything (S-32G: 472 -> ~270 ms).
    const uint64_t BLKB = c->tun.expand_dma >= 2 ? ((uint64_t)1 << 20) : ((uint64_t)64 << 20);      // (2: tests -- 1 MiB blocks, so that small outputs are split between threads and link too)
    const uint64_t nblk = (nout + BLKB - 1) / BLKB;
    std::mutex mu; uint64_t front = 0, back = nblk;      // blocks [front, back) are unclaimed
    auto claim = [&](bool from_back, uint64_t *blk) -> bool { std::lock_guard<std::mutex> g(mu); if (front >= back) return false; *blk = from_back ? --back : front++; return true; };
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t)
        th.emplace_back([&] {
            uint64_t blk;
            while (claim(false, &blk)) {
                const uint64_t b0 = blk * BLKB, b1 = b0 + BLKB < nout ? b0 + BLKB : nout;
                uint64_t lo = 0, hi = r;                    // last run that starts at or before b0 (run 0 starts at row 0)
                while (hi - lo > 1) { const uint64_t mid = lo + (hi - lo) / 2; if (row(mid) <= b0) lo = mid; else hi = mid; }
                uint64_t a = b0;
                RunWriter wr(host_bwt + b0);
                for (uint64_t k = lo; k < r && a < b1; ++k) { uint64_t e = row(k + 1); if (e > b1) e = b1; if (e > a) { wr.put(heads[(size_t)k], (size_t)(e - a)); a = e; } }
                wr.flush_all();
            }
#if defined(__SSE2__)
            _mm_sfence();      // the streaming stores of this thread are globally visible before it ends
#endif
        });
    int dma_rc = PFP_OK; uint64_t dma_blocks = 0;
    if (nblk >= (c->tun.expand_dma >= 2 ? 2u : 8u) && c->tun.expand_dma != 0 && host_pointer_is_pinned(host_bwt) && ensure_copy_stream(c) == PFP_OK) {
        uint64_t blk;
        while (dma_rc == PFP_OK && claim(true, &blk)) {
            const uint64_t b0 = blk * BLKB, b1 = b0 + BLKB < nout ? b0 + BLKB : nout;
            if (hipMemcpyAsync(host_bwt + b0, (const uint8_t *)c->d_bwt + b0, (size_t)(b1 - b0), hipMemcpyDeviceToHost, c->fa.copy) != hipSuccess || hipStreamSynchronize(c->fa.copy) != hipSuccess) { (void)hipGetLastError(); dma_rc = PFP_E_HIP; }
            ++dma_blocks;
        }
    }
    for (auto &t : th) t.join();
    if (c->tun.verbose) fprintf(stderr, "[pfbwt_hip] .bwt from its runs: %llu blocks of 64 MiB, %llu of them over the link\n", (unsigned long long)nblk, (unsigned long long)dma_blocks);
    return dma_rc;
}
int pfp_parse_docs(pfp_ctx *c, uint64_t *count) { if (!c || !count) return PFP_E_ARG; *count = c->doc_names.size(); return PFP_OK; }
int pfp_parse_doc_get(pfp_ctx *c, uint64_t i, const char **name, uint64_t *start)
{
    if (!c || i >= c->doc_names.size()) return PFP_E_ARG;
    if (name) *name = c->doc_names[(size_t)i].c_str();
    if (start) *start = c->doc_starts[(size_t)i];
    return PFP_OK;
}
int pfp_parse_fasta_records(pfp_ctx *c, uint64_t *raw_off, uint64_t *text_pos)
{
    if (!c) return PFP_E_ARG;
    for (size_t i = 0; i < c->fa.rec_raw.size(); ++i) { if (raw_off) raw_off[i] = c->fa.rec_raw[i]; if (text_pos) text_pos[i] = c->fa.rec_pos[i]; }
    return PFP_OK;
}

// ---- dictionary suffix sort (shared by the parse and the --pfbwt-only path) ---------------------
static int sort_dict_suffixes(pfp_ctx *c)
{
    const uint64_t N = c->dsize;
    const size_t mk = c->arena.mark_hi();
    uint64_t *k0, *k1; uint32_t *v0, *v1;
    PFP_ALLOC_LO(c, c->d_gsa, uint32_t, N);
    PFP_ALLOC_LO(c, c->d_srank, uint32_t, N);
    const size_t lo_state = c->arena.mark_lo();
    PFP_ALLOC_HI(c, k0, uint64_t, N); PFP_ALLOC_HI(c, k1, uint64_t, N);
    PFP_ALLOC_HI(c, v0, uint32_t, N); PFP_ALLOC_HI(c, v1, uint32_t, N);
    BitRange full = {0, DK_KEY_BITS};
    int rounds = 0;
    // A collection that is not repetitive has a dictionary about as large as its text whose suffixes are told apart by their first
    // few dozen characters: the rounds then read the text itself (sufsort.h, text rounds) and no per-offset rank array exists -- the
    // emission takes the class heads per SLOT (d_srank), the word ranks come from the word-start flags that travel with the
    // suffixes (d_sflag).  A repetitive collection (pangenome: dictionary << text, variant words share long prefixes) keeps the
    // rank-based rounds, whose covered prefix doubles / quadruples; so does a dictionary on which the text rounds are given up.
    // A repetitive collection's dictionary is repetitive itself: sorted through its own level-2 parse (dictrec.h); outputs as of the text rounds
    if (c->tun.dict_rec > 0 || (c->tun.dict_rec < 0 && c->n != 0 && N <= c->n / 8 && N >= ((uint64_t)1 << 22))) {
        PFP_ALLOC_LO(c, c->d_sflag, uint8_t, N);
        int taken = 0;
        PFP_TRY(dict_sort_pfp(c, c->d_gsa, c->d_srank, c->d_sflag, &taken));
        if (taken) { c->d_grank = nullptr; c->arena.release_hi(mk); c->gsa_valid = true; return PFP_OK; }
        c->d_sflag = nullptr; c->arena.release_lo(lo_state);
    }
    const int want = c->tun.dict_text_rounds;
    bool text_mode = want > 0 || (want < 0 && c->n != 0 && N > c->n / 8);
    if (text_mode) {
        PFP_ALLOC_LO(c, c->d_sflag, uint8_t, N);
        c->d_grank = nullptr;
        PFP_LAUNCH(c, K_SS_INIT_KEYS, N * 13, k_dict_init_keys, nblocks(N, DK_TILE), (const uint8_t *)c->d_dict, N, k0, v0);
        int conv = 1;
        PFP_TRY(suffix_sort_doubling<true>(c, N, k0, v0, k1, v1, &full, 1, DK_CHARS, c->d_dict, c->d_gsa, (uint32_t *)nullptr, (uint2 *)nullptr, &rounds, 9, c->d_srank, c->d_sflag, 1, &conv));
        if (!conv) { text_mode = false; c->d_sflag = nullptr; c->arena.release_lo(lo_state); }
    }
    if (!text_mode) {
        PFP_ALLOC_LO(c, c->d_grank, uint2, N);
        PFP_LAUNCH(c, K_SS_INIT_KEYS, N * 13, k_dict_init_keys, nblocks(N, DK_TILE), (const uint8_t *)c->d_dict, N, k0, v0);
        PFP_TRY(suffix_sort_doubling<true>(c, N, k0, v0, k1, v1, &full, 1, DK_CHARS, c->d_dict, c->d_gsa, (uint32_t *)nullptr, c->d_grank, &rounds, 9, c->d_srank));
    }
    c->arena.release_hi(mk);
    c->gsa_valid = true;
    return PFP_OK;
}

// De-duplicates the m byte strings described by (Y, sp) -- the std::map of pfparser.hpp:69-70, 595-597 -- exactly, with a
// hash table of representatives (parse.h).  Outputs: number of distinct strings, d_id[j] = id of string j (ids follow the
// sorted hashes of the distinct strings), rep[id] = a string with that id, occw[id] = how many strings have it.
static int dedup_strings(pfp_ctx *c, const uint8_t *Y, Spans sp, uint64_t m, uint64_t total_bytes, uint32_t *d_id, uint64_t *ndistinct, uint32_t **rep_out, uint32_t **occw_out, uint8_t *last_out)
{
    uint32_t *longlist, *d_u32, *slotof;
    const size_t mk0 = c->arena.mark_hi();
    const uint64_t maxlong = total_bytes / LONG_PHRASE + 2;
    PFP_ALLOC_HI(c, longlist, uint32_t, maxlong);
    PFP_ALLOC_HI(c, d_u32, uint32_t, 8);
    uint32_t *d_abandon; PFP_ALLOC_HI(c, d_abandon, uint32_t, 64);      // a 256-byte block of its own
    PFP_ALLOC_HI(c, slotof, uint32_t, m);
    const unsigned gm = nblocks(m, BLOCK);
    const int force_small = c->tun.dedup_table_log2;      // tests: a first table that overflows
    DedupTable t; uint32_t nd = 0;
    for (int attempt = 0;; ++attempt) {
        if (attempt == 2) return PFP_E_CORRUPT;
        const size_t mk = c->arena.mark_hi();
        // first guess: a collection is repetitive (distinct strings << strings); if more than half of that table fills up, the
        // second table holds twice the number of strings, which cannot overflow
        uint64_t want = attempt == 0 ? m / 4 : 2 * m;
        const uint64_t floor_ = 4 * m < (1ULL << 20) ? 4 * m : (1ULL << 20);     // small inputs: 4 entries per string (cannot overflow); else at least 2^20
        if (want < floor_) want = floor_;
        int lg = 10; while ((1ULL << lg) < want) ++lg;
        if (attempt == 0 && force_small) lg = force_small;
        const uint64_t T = 1ULL << lg;
        const uint64_t limit = T >= 2 * m ? m + 1 : T / 2;
        if (limit >= 0x7FFFFFFFULL || T > 0x80000000ULL) return PFP_E_TOO_LARGE;      // dense entry indices and slots share 31 bits of slotof[]
        PFP_ALLOC_HI(c, t.ent, DedupEntry, T);
        PFP_ALLOC_HI(c, t.dslot, uint32_t, limit); PFP_ALLOC_HI(c, t.dhash, uint64_t, limit);
        t.mask = T - 1; t.slotof = slotof; t.nd = d_u32; t.limit = (uint32_t)limit; t.overflow = d_u32 + 1; t.abandon = d_abandon;
        PFP_HIP(c, hipMemsetAsync(d_abandon, 0, 4, c->stream));
        PFP_HIP(c, hipMemsetAsync(t.ent, 0xFF, T * sizeof(DedupEntry), c->stream));
        PFP_HIP(c, hipMemsetAsync(d_u32, 0, 32, c->stream));
        unsigned long long *d_phase = nullptr;
        if (c->tun.dedup_phases) { PFP_ALLOC_HI(c, d_phase, unsigned long long, 512); PFP_HIP(c, hipMemsetAsync(d_phase, 0, 4096, c->stream)); }
        // the order of the workgroups (parse.h, DedupOrder): columns of loci per XCD when the text is a collection of similar sequences
        const uint64_t nb = gm;
        uint64_t grid = nb;
        const DedupOrder ord = make_dedup_order(nb, sp.ys32 ? 0 : c->nseq + c->fa.records, c->tun.dedup_period, c->tun.dedup_chunk, 64 /* sequences of less than ~1.6 Mbase: nothing to gain */, &grid);
        if (c->tun.verbose && ord.period) fprintf(stderr, "[pfbwt_hip] text de-duplication: %llu workgroups visited as %u sequences x %u loci, columns of %u per XCD (grid %llu)\n", (unsigned long long)nb, ord.rows, ord.period, ord.chunk, (unsigned long long)grid);
        // variant: 1 = cooperative, 0 = per lane, -1 (default) = cooperative for a collection (>= 8 sequences fed) as long as its first table lasts (parse.h)
        const bool coop = c->tun.dedup_variant > 0 || (c->tun.dedup_variant < 0 && attempt == 0 && !sp.ys32 && c->nseq + c->fa.records >= 8);
        if (!coop) PFP_LAUNCH(c, K_PHRASE_HASH, 2 * total_bytes + m * 24, k_dedup_insert<false>, grid, Y, sp, m, c->hash_seed, t, longlist, d_u32 + 2, last_out, d_phase, ord);
        else PFP_LAUNCH(c, K_PHRASE_HASH, 2 * total_bytes + m * 24, k_dedup_insert<true>, grid, Y, sp, m, c->hash_seed, t, longlist, d_u32 + 2, last_out, d_phase, ord);
        if (d_phase) {
            unsigned long long hp[512], tot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            PFP_HIP(c, hipMemcpyAsync(hp, d_phase, 4096, hipMemcpyDeviceToHost, c->stream));
            PFP_HIP(c, hipStreamSynchronize(c->stream));
            for (int q = 0; q < 64; ++q) for (int k = 0; k < 8; ++k) tot[k] += hp[q * 8 + k];
            const double nwg = tot[5] ? (double)tot[5] : 1.0;
            fprintf(stderr, "[pfbwt_hip] k_dedup_insert (%s), thread 0 of %llu workgroups, mean us per stage: abandon flag %.2f, spans + window bounds %.2f, window into LDS %.2f, hash %.2f, table + compare %.2f\n",
                    coop ? "cooperative" : "per lane", tot[5], tot[0] / nwg / 100.0, tot[1] / nwg / 100.0, tot[2] / nwg / 100.0, tot[3] / nwg / 100.0, tot[4] / nwg / 100.0);
        }
        uint32_t h3[3];
        PFP_HIP(c, hipMemcpyAsync(h3, d_u32, 12, hipMemcpyDeviceToHost, c->stream));
        PFP_HIP(c, hipStreamSynchronize(c->stream));
        if (!h3[1] && h3[2]) {
            PFP_LAUNCH_B(c, K_PHRASE_HASH_LONG, 2.0 * total_bytes / 64, k_dedup_insert_long, h3[2], DL_THREADS, Y, sp, (const uint32_t *)longlist, c->hash_seed, t);
            PFP_HIP(c, hipMemcpyAsync(h3, d_u32, 12, hipMemcpyDeviceToHost, c->stream));
            PFP_HIP(c, hipStreamSynchronize(c->stream));
        }
        if (!h3[1]) { nd = h3[0]; break; }
        if (c->tun.verbose) fprintf(stderr, "[pfbwt_hip] phrase table of 2^%d entries overflowed (%u distinct so far, state %u): retry\n", lg, h3[0], h3[1]);
        c->arena.release_hi(mk);
    }
    // ids in the order of the hashes of the distinct strings
    uint64_t *k1, *sk; uint32_t *v0, *v1, *sv, *rep, *occw;
    PFP_ALLOC_HI(c, rep, uint32_t, (size_t)nd + 1); PFP_ALLOC_HI(c, occw, uint32_t, (size_t)nd + 1);
    PFP_ALLOC_HI(c, k1, uint64_t, nd); PFP_ALLOC_HI(c, v0, uint32_t, nd); PFP_ALLOC_HI(c, v1, uint32_t, nd);
    PFP_LAUNCH(c, K_MISC, nd * 4, k_iota_u32, nblocks(nd, BLOCK), v0, (uint64_t)nd);
    BitRange full = {0, 64};
    PFP_TRY(radix_sort_pairs<uint64_t>(c, t.dhash, v0, k1, v1, nd, &full, 1, &sk, &sv));
    uint32_t *idofk;
    PFP_ALLOC_HI(c, idofk, uint32_t, nd);
    PFP_LAUNCH(c, K_DEDUP_HEADS, (uint64_t)nd * 44, k_dedup_assign, nblocks(nd, BLOCK), (const uint32_t *)sv, (uint64_t)nd, t, rep, occw, idofk);
    PFP_LAUNCH(c, K_DEDUP_HEADS, m * 12, k_dedup_ids, gm, (const DedupEntry *)t.ent, (const uint32_t *)idofk, (const uint32_t *)slotof, m, d_id);
    // the table (32 bytes per entry) is dead now: give its space back and keep only rep / occw, moved to the top of what it
    // occupied (they were allocated below it, and are smaller than it: the two regions cannot overlap)
    c->arena.release_hi(mk0);
    uint32_t *rep2, *occw2;
    PFP_ALLOC_HI(c, rep2, uint32_t, (size_t)nd + 1); PFP_ALLOC_HI(c, occw2, uint32_t, (size_t)nd + 1);
    if ((char *)occw2 < (char *)(rep + nd + 1) + ((size_t)nd + 1) * 4) return PFP_E_CORRUPT;      // cannot happen (see above)
    PFP_HIP(c, hipMemcpyAsync(rep2, rep, ((size_t)nd + 1) * 4, hipMemcpyDeviceToDevice, c->stream));
    PFP_HIP(c, hipMemcpyAsync(occw2, occw, ((size_t)nd + 1) * 4, hipMemcpyDeviceToDevice, c->stream));
    *ndistinct = nd; *rep_out = rep2; *occw_out = occw2;
    return PFP_OK;
}

// The dictionary D' (distinct strings in id order, each + EndOfWord, then EndOfDict), its word starts and the
// word id of every offset.
static int build_dictionary(pfp_ctx *c, const uint8_t *Y, Spans sp, const uint32_t *rep, uint64_t dwords)
{
    uint32_t *wlen1; tpos_t *srcstart;
    PFP_ALLOC_HI(c, wlen1, uint32_t, dwords); PFP_ALLOC_HI(c, srcstart, tpos_t, dwords);
    const unsigned gd = nblocks(dwords, BLOCK);
    PFP_LAUNCH(c, K_MISC, dwords * 16, k_word_lengths, gd, sp, rep, dwords, wlen1);
    PFP_ALLOC_LO(c, c->d_ws, uint32_t, dwords + 1);
    PFP_TRY((device_scan<uint32_t, 0>(c, wlen1, c->d_ws, dwords, c->d_ws + dwords)));
    uint32_t dsm1 = 0; PFP_TRY(d2h_u32(c, c->d_ws + dwords, &dsm1));
    const uint64_t dsize = (uint64_t)dsm1 + 1;
    c->dwords = dwords; c->dsize = dsize;
    if (dsize + 64 >= 0xFFFFFFFFULL) return PFP_E_TOO_LARGE;
    PFP_ALLOC_LO(c, c->d_dict, uint8_t, dsize + 16);
    PFP_ALLOC_LO(c, c->d_wordid, uint32_t, dsize);
    PFP_LAUNCH(c, K_MISC, dwords * 12, k_rep_starts, gd, sp, rep, dwords, srcstart);
    PFP_LAUNCH(c, K_DICT_BUILD, dsize * 6, k_dict_build, nblocks(dsize, 16 * BLOCK), Y, (const tpos_t *)srcstart, (const uint32_t *)c->d_ws, (uint32_t)dwords, dsize, c->d_dict, c->d_wordid);
    return PFP_OK;
}

// From the dictionary to ranks (sort_dict + generate_ranks, pfparser.hpp:494-517), occ (:471-480), the parse
// and the sorted .dict image.  Needs c->m, d_pid, dwords/dsize/d_ws/d_dict/d_wordid; occw = occurrences per word id.
static int finish_parse(pfp_ctx *c, const uint32_t *occw)
{
    const uint64_t m = c->m, dwords = c->dwords, dsize = c->dsize;
    const unsigned gm = nblocks(m, BLOCK), gd = nblocks(dwords, BLOCK);
    // suffix sort of the dictionary: gives word ranks now and the emission order later
    PFP_TRY(sort_dict_suffixes(c));
    uint32_t *wk0, *wk1, *wv0, *wv1, *idofrank, *len1; tpos_t *srcstart;
    PFP_ALLOC_HI(c, wk0, uint32_t, dwords); PFP_ALLOC_HI(c, wk1, uint32_t, dwords); PFP_ALLOC_HI(c, wv0, uint32_t, dwords); PFP_ALLOC_HI(c, wv1, uint32_t, dwords);
    PFP_ALLOC_HI(c, idofrank, uint32_t, dwords); PFP_ALLOC_HI(c, len1, uint32_t, dwords + 1); PFP_ALLOC_HI(c, srcstart, tpos_t, dwords);
    PFP_ALLOC_LO(c, c->d_wrank, uint32_t, dwords);
    PFP_ALLOC_LO(c, c->d_occ, uint32_t, dwords);
    PFP_ALLOC_LO(c, c->d_parse, uint32_t, m + 1);
    PFP_ALLOC_LO(c, c->d_sdict, uint8_t, dsize + 16);
    if (c->d_grank) {
        PFP_LAUNCH(c, K_WORD_RANK, dwords * 16, k_wordstart_keys, gd, (const uint32_t *)c->d_ws, (const uint2 *)c->d_grank, dwords, wk0, wv0);
        BitRange br = {0, bits_for(dsize)};
        uint32_t *sk32, *sv32;
        PFP_TRY(radix_sort_pairs<uint32_t>(c, wk0, wv0, wk1, wv1, dwords, &br, 1, &sk32, &sv32));
        PFP_LAUNCH(c, K_WORD_RANK, dwords * 20, k_word_rank, gd, (const uint32_t *)sv32, dwords, occw, c->d_wrank, idofrank, c->d_occ);
    } else {
        // text-round sort: the suffixes that start a word carry a flag to their slots; in slot order they ARE the words in
        // lexicographic order (two distinct words are never byte-identical), so their word ids, compacted, are the ids by rank
        const size_t mkf = c->arena.mark_hi();
        const uint64_t nt = nblocks(dsize, WF_TILE);
        uint32_t *tcnt, *tbase, *d_cnt;
        PFP_ALLOC_HI(c, tcnt, uint32_t, nt); PFP_ALLOC_HI(c, tbase, uint32_t, nt); PFP_ALLOC_HI(c, d_cnt, uint32_t, 1);
        PFP_LAUNCH(c, K_WORD_RANK, dsize, k_flag_tile_count, nt, (const uint8_t *)c->d_sflag, dsize, tcnt);
        PFP_TRY((device_scan<uint32_t, 0>(c, tcnt, tbase, nt, d_cnt)));
        uint32_t nst = 0; PFP_TRY(d2h_u32(c, d_cnt, &nst));
        if (nst != dwords) return PFP_E_CORRUPT;
        PFP_LAUNCH(c, K_WORD_RANK, dsize + dwords * 24, k_word_rank_flags, nt, (const uint8_t *)c->d_sflag, (const uint32_t *)tbase, (const uint32_t *)c->d_gsa, (const uint32_t *)c->d_wordid, dsize, (uint32_t)dwords,
                   occw, c->d_wrank, idofrank, c->d_occ);
        c->arena.release_hi(mkf);
    }
    PFP_LAUNCH(c, K_PARSE_RANKS, m * 12, k_parse_ranks, gm, (const uint32_t *)c->d_pid, (const uint32_t *)c->d_wrank, m, c->d_parse);
    PFP_LAUNCH(c, K_DICT_SORTED, dwords * 12, k_sorted_lengths, gd, (const uint32_t *)c->d_ws, (const uint32_t *)idofrank, dwords, len1, srcstart);
    PFP_TRY((device_scan<uint32_t, 0>(c, len1, len1, dwords, len1 + dwords)));
    PFP_LAUNCH(c, K_DICT_SORTED, dsize * 2, k_dict_build, nblocks(dsize, 16 * BLOCK), (const uint8_t *)c->d_dict, (const tpos_t *)srcstart, (const uint32_t *)len1, (uint32_t)dwords, dsize,
               c->d_sdict, (uint32_t *)nullptr);
    return PFP_OK;
}

// A stage that fails (PFP_E_NOMEM, PFP_E_INVALID_CHAR, ...) must leave the workspace as it found it, so that the
// natural retry (a cheaper request, a larger workspace) does not start from a leaked high-water mark.
struct ArenaGuard {
    pfp_ctx *c; size_t lo, hi; bool armed = true;
    explicit ArenaGuard(pfp_ctx *c_) : c(c_), lo(c_->arena.lo), hi(c_->arena.hi) {}
    int done(int rc)
    {
        if (rc != PFP_OK && armed && c->arena.base) { (void)hipStreamSynchronize(c->stream); c->arena.lo = lo; c->arena.hi = hi; c->arena.failed = false; }
        return rc;
    }
};

static int parse_finalize_impl(pfp_ctx *c, pfp_parse_sizes *out, bool shard_only);
static int parse_finalize_entry(pfp_ctx *c, pfp_parse_sizes *out, bool shard_only)
{
    if (!c) return PFP_E_ARG;
    if (c->stage != 0) return PFP_E_STATE;
    if (c->n == 0) return PFP_E_ARG;
    PFP_HIP(c, hipSetDevice(c->device));
    ArenaGuard g(c);
    const int rc = g.done(parse_finalize_impl(c, out, shard_only));
    // after a failure the fed text is still there (stage 0): finalize can be retried (more workspace), more text can be
    // appended, or pfp_reset drops it
    if (rc != PFP_OK) { c->m = c->dwords = c->dsize = 0; c->gsa_valid = false; }
    return rc;
}
int pfp_parse_finalize(pfp_ctx *c, pfp_parse_sizes *out) { return parse_finalize_entry(c, out, false); }
int pfp_parse_finalize_shard(pfp_ctx *c, pfp_parse_sizes *out) { return parse_finalize_entry(c, out, true); }
static int parse_finalize_impl(pfp_ctx *c, pfp_parse_sizes *out, bool shard_only)
{
    HostTimer timer;
    const uint64_t n = c->n; const int w = c->w;
    PFP_TRY(ensure_arena(c, n));
    c->arena.reset();
    uint8_t *X = c->tb + 16; const uint8_t *Y = c->tb + 15;
    // Dollar in front, w Dollars behind (pfparser.hpp:315-318, 484-489)
    PFP_HIP(c, hipMemsetAsync(c->tb, Dollar, 16, c->stream));
    PFP_HIP(c, hipMemsetAsync(X + n, Dollar, (size_t)w, c->stream));
    const size_t mk = c->arena.mark_hi();

    // 1. trigger scan
    const unsigned gts = nblocks(n, 16 * BLOCK);
    uint16_t *mask16; uint64_t *blockcnt; uint32_t *d_u32; unsigned long long *d_err;
    PFP_ALLOC_HI(c, mask16, uint16_t, (size_t)gts * BLOCK);
    PFP_ALLOC_HI(c, blockcnt, uint64_t, (size_t)gts + 1);
    PFP_ALLOC_HI(c, d_u32, uint32_t, 8);
    PFP_ALLOC_HI(c, d_err, unsigned long long, 1);
    PFP_HIP(c, hipMemsetAsync(d_err, 0xff, 8, c->stream));
    PFP_HIP(c, hipMemsetAsync(d_u32, 0, 32, c->stream));
    const uint64_t kmask = (w == 32) ? 0ULL : ((1ULL << (2 * w)) - 1ULL);   // hash.hpp:26 (w == 32: observed x86 value)
    const bool no_trigtab = c->tun.no_trigger_table != 0;      // tests / measurements: the hash evaluated per base
    if (c->view.src && !(w <= TS_MAX_W && !no_trigtab)) PFP_TRY(flush_view(c));      // the hash-per-window scan reads the text from X
    if (w <= TS_MAX_W && !no_trigtab) {
        const uint32_t tabwords = (1u << (2 * w)) >= 32u ? (1u << (2 * w)) / 32u : 1u;
        if (!c->d_trigtab) {      // w and p are fixed for the life of a context
            PFP_HIP(c, hipMalloc((void **)&c->d_trigtab, (size_t)TS_TAB_WORDS * 4));
            PFP_LAUNCH(c, K_MISC, tabwords * 4, k_trigger_table, nblocks(tabwords, BLOCK), w, make_divtest(c->p), c->d_trigtab);
        }
        const uint64_t nthreads_total = (uint64_t)gts * BLOCK;
        const uint64_t tiles = (nthreads_total + TS_THREADS - 1) / TS_THREADS;
        uint32_t tpw = (uint32_t)(tiles / 1024); if (tpw < 1) tpw = 1; if (tpw > 64) tpw = 64;      // enough workgroups to fill 256 CUs, the table load amortised
        PFP_HIP(c, hipMemsetAsync(blockcnt, 0, ((size_t)gts + 1) * 8, c->stream));
        if (c->view.src) {      // the rows of pfp_parse_feed_device_view are read where they are; the scan writes the text (2 B per base instead of 1 + a copy pass)
            const RowView rv = {c->view.src, c->view.count, c->view.len, c->view.stride, c->view.len + (uint64_t)w};
            PFP_LAUNCH_B(c, K_TRIGGER_SCAN, 2 * n + n / 8, (k_trigger_scan_tab<true>), (tiles + tpw - 1) / tpw, TS_THREADS, X, n, w, (const uint32_t *)c->d_trigtab, tabwords, (uint32_t)kmask,
                         (int)((c->flags & PFP_FLAG_NON_ACGT_TO_A) != 0), tpw, nthreads_total, mask16, blockcnt, d_err, rv);
            c->view.src = nullptr;      // X holds the text from here on
        } else
        PFP_LAUNCH_B(c, K_TRIGGER_SCAN, n + n / 8, (k_trigger_scan_tab<false>), (tiles + tpw - 1) / tpw, TS_THREADS, X, n, w, (const uint32_t *)c->d_trigtab, tabwords, (uint32_t)kmask,
                     (int)((c->flags & PFP_FLAG_NON_ACGT_TO_A) != 0), tpw, nthreads_total, mask16, blockcnt, d_err, RowView{nullptr, 0, 0, 0, 1});
    } else
    PFP_LAUNCH(c, K_TRIGGER_SCAN, n * 2 + n / 8, k_trigger_scan, gts, X, n, w, make_divtest(c->p), kmask, (int)((c->flags & PFP_FLAG_NON_ACGT_TO_A) != 0), mask16, blockcnt, d_err);
    PFP_TRY((device_scan<uint64_t, 0>(c, blockcnt, blockcnt, gts, blockcnt + gts)));
    uint64_t ntrig = 0; unsigned long long herr = 0;
    PFP_HIP(c, hipMemcpyAsync(&herr, d_err, 8, hipMemcpyDeviceToHost, c->stream));
    PFP_HIP(c, hipMemcpyAsync(&ntrig, blockcnt + gts, 8, hipMemcpyDeviceToHost, c->stream));
    PFP_HIP(c, hipStreamSynchronize(c->stream));
    if (herr != ~0ULL) {   // hash.hpp:31
        uint8_t ch = 0;
        PFP_HIP(c, hipMemcpy(&ch, X + herr, 1, hipMemcpyDeviceToHost));
        c->err_pos = herr; c->err_ch = ch;
        return PFP_E_INVALID_CHAR;
    }
    const uint64_t m = ntrig + 1;
    if (m > 0xFFFFFFFEULL - 64) return PFP_E_TOO_LARGE;       // pfparser.hpp:399-404: more than 2^32-2 phrases is a hard limit of the reference too
    c->m = m;
    PFP_ALLOC_LO(c, c->d_ye, tpos_t, m);
    PFP_LAUNCH(c, K_PHRASE_ENDS, n / 8 + m * 4, k_phrase_ends, nblocks(gts, PE_BLOCKS), (const uint16_t *)mask16, (const uint64_t *)blockcnt, (uint64_t)gts, c->d_ye);
    PFP_LAUNCH(c, K_MISC, 8, k_set_u64, 1, c->d_ye, m - 1, (uint64_t)(n + (uint64_t)w));

    // 2. distinct phrases, dictionary
    Spans sp; sp.ye = c->d_ye; sp.ys32 = nullptr; sp.ye32 = nullptr; sp.w = w;
    uint64_t dwords = 0; uint32_t *rep, *occw;
    PFP_ALLOC_LO(c, c->d_pid, uint32_t, m);
    PFP_ALLOC_LO(c, c->d_last, uint8_t, m);
    PFP_TRY(dedup_strings(c, Y, sp, m, n + (uint64_t)w + 1 + m * (uint64_t)w, c->d_pid, &dwords, &rep, &occw, c->d_last));      // + last[j] = Y[ye[j] - w], pfparser.hpp:599
    PFP_TRY(build_dictionary(c, Y, sp, rep, dwords));
    // 3. dictionary suffix sort, ranks, occ, parse, sorted .dict image -- not for a shard that is only going to be merged: the
    //    merge sorts the united dictionary, and a shard's dictionary is nearly as large as the whole collection's
    if (!shard_only) PFP_TRY(finish_parse(c, occw));
    PFP_HIP(c, hipStreamSynchronize(c->stream));
    c->arena.release_hi(mk);
    c->stage = 1;
    c->lo_after_parse = c->arena.mark_lo();
    c->stage_ms[0] = timer.ms();
    if (out) { out->n = n; out->m = m; out->dwords = c->dwords; out->dsize = c->dsize; }
    return PFP_OK;
}

// ---- multi-GPU: merge of shard parses (semantics of PfParser::operator+=, pfparser.hpp:194-263) -----
// Shard r > 0 was parsed as the text  A^w + X_r  (the w 'A's that end shard r-1, pfparser.hpp:335-337, as left
// context), so its closed phrases are exactly the phrases of the whole text; only its first phrase (Dollar +
// context + head) and, for r < N-1, its last phrase (tail + w Dollars) are fragments: tail_r + head_{r+1} is
// the phrase that straddles the boundary.  All dictionaries are laid out in one buffer, the two fragment
// words of a boundary are re-pointed at the junction word, and the union is de-duplicated like phrases are.
__global__ __launch_bounds__(BLOCK) void k_merge_spans(const uint32_t *ws, uint32_t dwords, uint32_t ubase, uint32_t coff, uint32_t frag0, uint32_t frag0_ys, uint32_t frag0_ye,
                                                       uint32_t fragl, uint32_t fragl_ys, uint32_t fragl_ye, uint32_t *ys, uint32_t *ye)
{
    const uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
    if (k >= dwords) return;
    uint32_t a = ubase + ws[k], b = ubase + ws[k + 1] - 2u;   // word bytes without its EndOfWord
    if (k == frag0) { a = frag0_ys; b = frag0_ye; }
    if (k == fragl) { a = fragl_ys; b = fragl_ye; }
    ys[coff + k] = a; ye[coff + k] = b;
}
__global__ __launch_bounds__(BLOCK) void k_merge_phrases(const uint32_t *pid, const tpos_t *ye, const uint8_t *last, uint32_t m, uint32_t j0, uint32_t goff, uint32_t coff, tpos_t shift,
                                                         const uint32_t *cand_id, tpos_t junction_ye, uint32_t junction_last, int has_junction,
                                                         uint32_t *gpid, tpos_t *gye, uint8_t *glast, uint32_t *occw)
{
    const uint32_t j = j0 + blockIdx.x * BLOCK + threadIdx.x;
    if (j >= m) return;
    const uint32_t g = goff + j - j0;
    const uint32_t id = cand_id[coff + pid[j]];
    gpid[g] = id;
    atomicAdd(&occw[id], 1u);
    if (has_junction && j + 1 == m) { gye[g] = junction_ye; glast[g] = (uint8_t)junction_last; }   // the phrase that ends in the next shard
    else { gye[g] = ye[j] + shift; glast[g] = last[j]; }
}

__global__ __launch_bounds__(BLOCK) void k_merge_extra(const uint32_t *cand_id, const tpos_t *xye, const uint32_t *xlast, uint32_t nx, uint32_t goff, uint32_t *gpid, tpos_t *gye, uint8_t *glast, uint32_t *occw)
{
    const uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
    if (k >= nx) return;
    const uint32_t id = cand_id[k];
    gpid[goff + k] = id; gye[goff + k] = xye[k]; glast[goff + k] = (uint8_t)xlast[k];
    atomicAdd(&occw[id], 1u);
}

// ---- a saved parse (.dict + .parse) as a shard: load_parser pfbwt_io.hpp:211-222, init_from_dict_ranks pfparser.hpp:549-567 ----
__global__ __launch_bounds__(BLOCK) void k_shard_phrases(const uint32_t *parse, const uint32_t *ws, const uint8_t *dict, uint64_t m, uint32_t dwords, int w, uint32_t *pid, unsigned long long *adv, uint8_t *last, uint32_t *bad)
{
    const uint64_t j = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (j >= m) return;
    const uint32_t r = parse[j];
    if (r == 0 || r > dwords) { atomicAdd(bad, 1u); pid[j] = 0; adv[j] = 0; last[j] = 0; return; }
    const uint32_t id = r - 1, s = ws[id], len = ws[id + 1] - s - 1u;            // the dictionary image is in rank order: word id = rank - 1
    if (len <= (uint32_t)w) { atomicAdd(bad, 1u); pid[j] = 0; adv[j] = 0; last[j] = 0; return; }
    pid[j] = id;
    adv[j] = j ? (unsigned long long)(len - (uint32_t)w) : (unsigned long long)len;      // phrase 0 starts at Y[0] (its Dollar); later ones overlap by w
    last[j] = dict[s + len - (uint32_t)w - 1u];                                          // pfparser.hpp:599
}
// a shard view without d_ye / d_last (what travels over xGMI is dictionary + word starts + phrase ids only): phrase j is word
// pid[j], it advances the text by its length minus the w bytes it shares with its predecessor, its `last` byte sits in the word
__global__ __launch_bounds__(BLOCK) void k_view_phrases(const uint32_t *pid, const uint32_t *ws, const uint8_t *dict, uint64_t m, uint32_t dwords, int w, unsigned long long *adv, uint8_t *last, uint32_t *bad)
{
    const uint64_t j = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (j >= m) return;
    const uint32_t id = pid[j];
    if (id >= dwords) { atomicAdd(bad, 1u); adv[j] = 0; last[j] = 0; return; }
    const uint32_t s = ws[id], len = ws[id + 1] - s - 1u;
    if (len <= (uint32_t)w) { atomicAdd(bad, 1u); adv[j] = 0; last[j] = 0; return; }
    adv[j] = j ? (unsigned long long)(len - (uint32_t)w) : (unsigned long long)len;
    last[j] = dict[s + len - (uint32_t)w - 1u];
}
__global__ __launch_bounds__(BLOCK) void k_shard_ends(const unsigned long long *adv_ex, const unsigned long long *adv, uint64_t m, tpos_t *ye)
{
    const uint64_t j = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (j < m) ye[j] = (tpos_t)(adv_ex[j] + adv[j] - 1ULL);                              // Y coordinate of the phrase's last byte
}
int pfp_shard_load(pfp_ctx *c, const uint8_t *dict, uint64_t dsize, const uint32_t *parse, uint64_t m)
{
    if (!c || !dict || !parse || dsize < 3 || m < 1) return PFP_E_ARG;
    if (dsize + 64 >= 0xFFFFFFFFULL || m > 0xFFFFFFFEULL - 64) return PFP_E_TOO_LARGE;
    if (dict[dsize - 1] != EndOfDict || dict[dsize - 2] != EndOfWord) return PFP_E_CORRUPT;
    PFP_HIP(c, hipSetDevice(c->device));
    reset_results(c);
    auto body = [&]() -> int {
        // a shard context stays small: merge_pfp may hold thousands of them
        const size_t saved = c->arena_request;
        if (!saved) c->arena_request = (size_t)(dsize * 26 + m * 48 + ((size_t)4 << 20));
        const int ra = ensure_arena(c, 0);
        c->arena_request = saved;
        if (ra != PFP_OK) return ra;
        c->arena.reset();
        PFP_ALLOC_LO(c, c->d_dict, uint8_t, dsize + 16);
        PFP_ALLOC_LO(c, c->d_pid, uint32_t, m); PFP_ALLOC_LO(c, c->d_ye, tpos_t, m); PFP_ALLOC_LO(c, c->d_last, uint8_t, m);
        PFP_HIP(c, hipMemsetAsync(c->d_dict + dsize, 0, 16, c->stream));
        PFP_TRY(h2d_copy(c, c->d_dict, dict, dsize));
        const size_t mk = c->arena.mark_hi();
        uint32_t *flag, *wid, *d_cnt, *d_parse; unsigned long long *adv, *advx, *d_tot;
        PFP_ALLOC_HI(c, flag, uint32_t, dsize); PFP_ALLOC_HI(c, wid, uint32_t, dsize); PFP_ALLOC_HI(c, d_cnt, uint32_t, 2);
        PFP_LAUNCH(c, K_MISC, dsize * 5, k_eow_flags, nblocks(dsize, BLOCK), (const uint8_t *)c->d_dict, dsize, flag);
        PFP_TRY((device_scan<uint32_t, 0>(c, flag, wid, dsize, d_cnt)));
        uint32_t nw = 0; PFP_TRY(d2h_u32(c, d_cnt, &nw));
        if (nw < 1) return PFP_E_CORRUPT;
        PFP_ALLOC_LO(c, c->d_ws, uint32_t, (size_t)nw + 2);
        PFP_LAUNCH(c, K_MISC, dsize * 5, k_ws_from_flags, nblocks(dsize, BLOCK), (const uint8_t *)c->d_dict, dsize, (const uint32_t *)wid, c->d_ws);
        PFP_ALLOC_HI(c, d_parse, uint32_t, m); PFP_ALLOC_HI(c, adv, unsigned long long, m); PFP_ALLOC_HI(c, advx, unsigned long long, m); PFP_ALLOC_HI(c, d_tot, unsigned long long, 1);
        PFP_TRY(h2d_copy(c, (uint8_t *)d_parse, (const uint8_t *)parse, m * 4));
        PFP_HIP(c, hipMemsetAsync(d_cnt + 1, 0, 4, c->stream));
        PFP_LAUNCH(c, K_MISC, m * 24, k_shard_phrases, nblocks(m, BLOCK), (const uint32_t *)d_parse, (const uint32_t *)c->d_ws, (const uint8_t *)c->d_dict, m, nw, c->w, c->d_pid, adv, c->d_last, d_cnt + 1);
        PFP_TRY((device_scan<unsigned long long, 0>(c, adv, advx, m, d_tot)));
        PFP_LAUNCH(c, K_MISC, m * 24, k_shard_ends, nblocks(m, BLOCK), (const unsigned long long *)advx, (const unsigned long long *)adv, m, c->d_ye);
        uint32_t bad = 0; unsigned long long tot = 0;
        PFP_HIP(c, hipMemcpyAsync(&bad, d_cnt + 1, 4, hipMemcpyDeviceToHost, c->stream));
        PFP_HIP(c, hipMemcpyAsync(&tot, d_tot, 8, hipMemcpyDeviceToHost, c->stream));
        PFP_HIP(c, hipStreamSynchronize(c->stream));
        if (bad || tot < (unsigned long long)c->w + 2) return PFP_E_CORRUPT;           // a rank outside the dictionary, or a word of at most w bytes
        c->arena.release_hi(mk);
        c->dwords = nw; c->dsize = dsize; c->m = m;
        c->n = tot - 1 - (uint64_t)c->w;                                               // Y holds Dollar + X + w Dollars
        c->left_ctx = 0; c->stage = 1;
        return PFP_OK;
    };
    const int rc = body();
    if (rc != PFP_OK) reset_results(c);
    return rc;
}

int pfp_shard_view_get(pfp_ctx *c, pfp_shard_view *v)
{
    if (!c || !v) return PFP_E_ARG;
    if (c->stage < 1 || !c->d_pid || !c->d_last) return PFP_E_STATE;
    v->n = c->n; v->m = c->m; v->dwords = c->dwords; v->dsize = c->dsize;
    v->d_dict = c->d_dict; v->d_ws = c->d_ws; v->d_pid = c->d_pid; v->d_ye = c->d_ye; v->d_last = c->d_last;
    v->left_context = c->left_ctx;
    return PFP_OK;
}

int pfp_device_copy(pfp_ctx *c, void *d_dst, const void *d_src, uint64_t bytes)
{
    if (!c || (!d_dst && bytes) || (!d_src && bytes)) return PFP_E_ARG;
    PFP_HIP(c, hipSetDevice(c->device));
    if (bytes) PFP_HIP(c, hipMemcpyAsync(d_dst, d_src, (size_t)bytes, hipMemcpyDeviceToDevice, c->stream));
    PFP_HIP(c, hipStreamSynchronize(c->stream));
    return PFP_OK;
}

static int merge_shards_impl(pfp_ctx *c, int nshards, const pfp_shard_view *v, pfp_parse_sizes *out);
int pfp_merge_shards(pfp_ctx *c, int nshards, const pfp_shard_view *v, pfp_parse_sizes *out)
{
    if (!c || nshards < 1 || !v) return PFP_E_ARG;
    PFP_HIP(c, hipSetDevice(c->device));
    const int rc = merge_shards_impl(c, nshards, v, out);
    if (rc != PFP_OK) reset_results(c);      // a failed merge leaves an empty context (the shards are untouched)
    return rc;
}
static int merge_shards_impl(pfp_ctx *c, int nshards, const pfp_shard_view *v, pfp_parse_sizes *out)
{
    HostTimer timer;
    const uint32_t w = (uint32_t)c->w;
    uint64_t ntot = 0, mtot = 0, dtot = 0, ctot = 0;
    // interior phrases of shard r = [ia[r], ib[r]): all but the head fragment (r > 0) and the tail fragment (r < N-1); a shard
    // of ONE phrase (a short record or contig without a trigger window) in the middle is both at once and has none
    std::vector<uint64_t> ia((size_t)nshards), ib((size_t)nshards);
    for (int r = 0; r < nshards; ++r) {
        const uint64_t lc = v[r].left_context;
        if ((lc != 0 && lc != w) || (r == 0 && lc != 0)) return PFP_E_ARG;
        if (v[r].m < 1 || v[r].n < lc + 1 || !v[r].d_dict || !v[r].d_ws || !v[r].d_pid || (!v[r].d_ye) != (!v[r].d_last)) return PFP_E_ARG;      // d_ye and d_last: both or neither (derived below)
        ia[r] = r ? 1 : 0; ib[r] = r + 1 < nshards ? v[r].m - 1 : v[r].m;
        if (ib[r] < ia[r]) ib[r] = ia[r];
        ntot += v[r].n - lc; mtot += ib[r] - ia[r]; dtot += v[r].dsize; ctot += v[r].dwords;
    }
    if (ntot + w + 64 >= ((c->flags & PFP_FLAG_U64) ? (1ULL << 40) : 0xFFFFFFFFULL) || dtot + 64 >= 0xFFFFFFFFULL || mtot > 0xFFFFFFFEULL - 64) return PFP_E_TOO_LARGE;
    reset_results(c);
    PFP_TRY(ensure_arena(c, ntot + dtot));
    c->arena.reset();
    PFP_HIP(c, hipStreamSynchronize(c->stream));
    const size_t mk = c->arena.mark_hi();
    // ---- host: the fragment words of every boundary
    struct Frag { uint32_t id0 = 0xFFFFFFFFu, idl = 0xFFFFFFFFu; std::vector<uint8_t> w0, wl; };
    std::vector<Frag> fr((size_t)nshards);
    auto fetch_word = [&](const pfp_shard_view &sv, uint32_t id, std::vector<uint8_t> &dst) -> int {
        uint32_t se[2];
        PFP_HIP(c, hipMemcpy(se, sv.d_ws + id, 8, hipMemcpyDeviceToHost));
        dst.resize(se[1] - se[0] - 1);
        if (!dst.empty()) PFP_HIP(c, hipMemcpy(dst.data(), sv.d_dict + se[0], dst.size(), hipMemcpyDeviceToHost));
        return PFP_OK;
    };
    for (int r = 0; r < nshards; ++r) {
        if (r > 0 || (v[r].m == 1 && nshards > 1)) {      // first phrase: head fragment (shard 0 with one phrase: the start of the first junction phrase)
            PFP_HIP(c, hipMemcpy(&fr[r].id0, v[r].d_pid, 4, hipMemcpyDeviceToHost));
            PFP_TRY(fetch_word(v[r], fr[r].id0, fr[r].w0));
            if (fr[r].w0.size() < 1 + (size_t)w) return PFP_E_CORRUPT;
        }
        if (r + 1 < nshards && v[r].m > 1) {
            PFP_HIP(c, hipMemcpy(&fr[r].idl, v[r].d_pid + (v[r].m - 1), 4, hipMemcpyDeviceToHost));
            PFP_TRY(fetch_word(v[r], fr[r].idl, fr[r].wl));
            if (fr[r].wl.size() < (size_t)w) return PFP_E_CORRUPT;
        }
    }
    // Junction phrases.  PfParser::operator+= (pfparser.hpp:194-263) pops the last phrase of the left operand, strips its w Dollars and
    // goes on appending the first phrase of the right operand without its Dollar (and without the left context, if the shard was
    // parsed with one).  A shard that knew its left context has already cut its head at every trigger; a stand-alone shard could
    // not trigger in its first w windows (pfparser.hpp:347 "pos_ > w"), so those windows are re-tested here exactly as :226-245 does:
    // the hasher starts from w 'A's, a trigger closes the open phrase and the next one starts with the open phrase's real last
    // w characters (:241).  The head of a shard with two or more phrases ends in a trigger window and closes the open phrase; the
    // head of a one-phrase shard (no window of its own triggered) leaves it open for the next operand -- folding the operands one
    // by one, as the reference does, gives the same chain.  All junction phrases are "extra" phrases: pieces[s] = the phrases
    // closed while the head of shard s was appended, in text order; they sit between the interior phrases of shards s-1 and s.
    struct Piece { uint32_t js, je; tpos_t ye; uint8_t last; };               // span in the junction buffer, global end position, last char
    std::vector<uint8_t> junc; std::vector<std::vector<Piece>> pieces((size_t)nshards);
    std::vector<tpos_t> shift((size_t)nshards);
    { uint64_t g = 0; for (int r = 0; r < nshards; ++r) { shift[r] = (tpos_t)(g - v[r].left_context); g += v[r].n - v[r].left_context; } }
    const DivTest dt = make_divtest(c->p);
    const uint64_t kmask = (w == 32) ? 0ULL : ((1ULL << (2 * w)) - 1ULL);
    uint64_t extra = 0;
    if (nshards > 1) {
        std::vector<uint8_t> cur;                                              // the open phrase
        if (v[0].m == 1) cur.assign(fr[0].w0.begin(), fr[0].w0.end() - w);     // Dollar + text of shard 0 (its w Dollars stripped, :211-214)
        else cur.assign(fr[0].wl.begin(), fr[0].wl.end() - w);
        for (int s = 1; s < nshards; ++s) {
            const uint32_t lc = (uint32_t)v[s].left_context;
            const bool one = v[s].m == 1, lastshard = s + 1 == nshards;
            const bool closes = !one || lastshard;                             // the head ends the open phrase (a trigger window, or the end of the text)
            if (fr[s].w0.size() < 1 + (size_t)lc + 1 + ((one && !lastshard) ? (size_t)w : 0)) return PFP_E_CORRUPT;
            const std::vector<uint8_t> head(fr[s].w0.begin() + 1 + lc, fr[s].w0.end() - ((one && !lastshard) ? w : 0));
            const size_t textlen = one && lastshard ? (head.size() >= (size_t)w ? head.size() - w : 0) : head.size();      // head characters that are text (not the final Dollars)
            const tpos_t G = shift[s] + lc;                                    // global index of the head's first character
            auto add_piece = [&](tpos_t ye) -> int {
                if (cur.size() <= (size_t)w) return PFP_E_CORRUPT;
                Piece pc; pc.js = (uint32_t)junc.size(); junc.insert(junc.end(), cur.begin(), cur.end()); pc.je = (uint32_t)junc.size() - 1u;
                pc.ye = ye; pc.last = cur[cur.size() - w - 1];
                pieces[s].push_back(pc);
                return PFP_OK;
            };
            size_t from = 0;                                                   // first head character not yet appended to the open phrase
            if (lc == 0) {
                uint64_t kmer = 0;
                for (uint32_t i = 0; i < w && i < textlen; ++i) {
                    const uint8_t ch = head[i];
                    const uint64_t code = (ch == 'C') ? 1 : (ch == 'G') ? 2 : (ch == 'T' || ch == '-') ? 3 : 0;      // the text is normalised: A, N -> 0
                    kmer = ((kmer << 2) | code) & kmask;
                    if (!divisible(wang_hash(kmer), dt)) continue;
                    if (closes && (size_t)i + 1 == head.size()) continue;      // the head's own last character ends the phrase anyway
                    cur.insert(cur.end(), head.begin() + from, head.begin() + i + 1); from = (size_t)i + 1;
                    PFP_TRY(add_piece((tpos_t)(G + i + 1)));
                    if (cur.size() < (size_t)w) return PFP_E_ARG;              // (the reference's phrase.erase would throw)
                    cur.erase(cur.begin(), cur.end() - w);                     // the next phrase starts with the real last w characters, :241
                }
            }
            cur.insert(cur.end(), head.begin() + from, head.end());
            if (closes) {
                PFP_TRY(add_piece((tpos_t)(G + head.size())));
                if (!lastshard) cur.assign(fr[s].wl.begin(), fr[s].wl.end() - w);      // the next open phrase: this shard's last one without its Dollars
            }
            extra += pieces[s].size();
        }
    }
    if (mtot + extra > 0xFFFFFFFEULL - 64) return PFP_E_TOO_LARGE;
    mtot += extra;
    // ---- device: one buffer with all dictionaries + junction words, candidate spans
    uint8_t *U; uint32_t *cys, *cye, *cand_id;
    const uint64_t call = ctot + extra;                                        // the shards' words + the junction phrases
    PFP_ALLOC_HI(c, U, uint8_t, dtot + junc.size() + 64);
    PFP_ALLOC_HI(c, cys, uint32_t, call); PFP_ALLOC_HI(c, cye, uint32_t, call); PFP_ALLOC_HI(c, cand_id, uint32_t, call);
    std::vector<uint32_t> ubase((size_t)nshards), coff((size_t)nshards);
    { uint64_t ub = 0, co = 0; for (int r = 0; r < nshards; ++r) { ubase[r] = (uint32_t)ub; coff[r] = (uint32_t)co; ub += v[r].dsize; co += v[r].dwords; } }
    for (int r = 0; r < nshards; ++r) PFP_HIP(c, hipMemcpyAsync(U + ubase[r], v[r].d_dict, (size_t)v[r].dsize, hipMemcpyDeviceToDevice, c->stream));
    if (!junc.empty()) PFP_HIP(c, hipMemcpyAsync(U + dtot, junc.data(), junc.size(), hipMemcpyHostToDevice, c->stream));
    const uint32_t jb = (uint32_t)dtot;
    std::vector<uint32_t> xs, xe, xlast; std::vector<tpos_t> xye; std::vector<uint32_t> xfirst((size_t)nshards + 1, 0);
    for (int s = 0; s < nshards; ++s) {
        xfirst[s] = (uint32_t)xs.size();
        for (const Piece &pc : pieces[s]) { xs.push_back(jb + pc.js); xe.push_back(jb + pc.je); xye.push_back(pc.ye); xlast.push_back(pc.last); }
    }
    xfirst[nshards] = (uint32_t)xs.size();
    for (int r = 0; r < nshards; ++r) {
        // fragment words are referenced by no phrase that is kept: they are re-pointed at the first junction phrase, so that the
        // united dictionary holds no orphan
        const bool f0 = r > 0 || (v[r].m == 1 && nshards > 1), fl = r + 1 < nshards && v[r].m > 1;
        const uint32_t fs = extra ? xs[0] : 0u, fe = extra ? xe[0] : 0u;
        PFP_LAUNCH(c, K_MISC, v[r].dwords * 12, k_merge_spans, nblocks(v[r].dwords, BLOCK), v[r].d_ws, (uint32_t)v[r].dwords, ubase[r], coff[r],
                   f0 ? fr[r].id0 : 0xFFFFFFFFu, fs, fe, fl ? fr[r].idl : 0xFFFFFFFFu, fs, fe, cys, cye);
    }
    tpos_t *d_xye = nullptr; uint32_t *d_xlast = nullptr;
    if (extra) {
        PFP_ALLOC_HI(c, d_xye, tpos_t, extra); PFP_ALLOC_HI(c, d_xlast, uint32_t, extra);
        PFP_HIP(c, hipMemcpyAsync(cys + ctot, xs.data(), extra * 4, hipMemcpyHostToDevice, c->stream));
        PFP_HIP(c, hipMemcpyAsync(cye + ctot, xe.data(), extra * 4, hipMemcpyHostToDevice, c->stream));
        PFP_HIP(c, hipMemcpyAsync(d_xye, xye.data(), extra * sizeof(tpos_t), hipMemcpyHostToDevice, c->stream));
        PFP_HIP(c, hipMemcpyAsync(d_xlast, xlast.data(), extra * 4, hipMemcpyHostToDevice, c->stream));
        PFP_HIP(c, hipStreamSynchronize(c->stream));                          // the host vectors go out of use below
    }
    // ---- global distinct words, dictionary
    Spans sp; sp.ye = nullptr; sp.ys32 = cys; sp.ye32 = cye; sp.w = 0;
    uint64_t dwords = 0; uint32_t *rep, *occ_cand, *occw;
    PFP_TRY(dedup_strings(c, U, sp, call, dtot + junc.size(), cand_id, &dwords, &rep, &occ_cand, (uint8_t *)nullptr));
    PFP_TRY(build_dictionary(c, U, sp, rep, dwords));
    // ---- global phrase sequence: junction phrases closed by the head of shard r, then the interior phrases of shard r
    c->n = ntot; c->m = mtot;
    PFP_ALLOC_LO(c, c->d_pid, uint32_t, mtot); PFP_ALLOC_LO(c, c->d_ye, tpos_t, mtot); PFP_ALLOC_LO(c, c->d_last, uint8_t, mtot);
    PFP_ALLOC_HI(c, occw, uint32_t, dwords);
    PFP_HIP(c, hipMemsetAsync(occw, 0, dwords * 4, c->stream));
    // compact views (no d_ye / d_last): phrase ends and last bytes from the shard's own dictionary and phrase ids
    std::vector<const tpos_t *> vye((size_t)nshards); std::vector<const uint8_t *> vlast((size_t)nshards);
    for (int r = 0; r < nshards; ++r) {
        vye[r] = v[r].d_ye; vlast[r] = v[r].d_last;
        if (v[r].d_ye) continue;
        const uint64_t mr = v[r].m;
        tpos_t *ye; uint8_t *la; unsigned long long *adv, *advx, *d_tot; uint32_t *d_bad;
        PFP_ALLOC_HI(c, ye, tpos_t, mr); PFP_ALLOC_HI(c, la, uint8_t, mr);
        const size_t mk2 = c->arena.mark_hi();
        PFP_ALLOC_HI(c, adv, unsigned long long, mr); PFP_ALLOC_HI(c, advx, unsigned long long, mr); PFP_ALLOC_HI(c, d_tot, unsigned long long, 1); PFP_ALLOC_HI(c, d_bad, uint32_t, 1);
        PFP_HIP(c, hipMemsetAsync(d_bad, 0, 4, c->stream));
        PFP_LAUNCH(c, K_MISC, mr * 24, k_view_phrases, nblocks(mr, BLOCK), v[r].d_pid, v[r].d_ws, v[r].d_dict, mr, (uint32_t)v[r].dwords, (int)w, adv, la, d_bad);
        PFP_TRY((device_scan<unsigned long long, 0>(c, adv, advx, mr, d_tot)));
        PFP_LAUNCH(c, K_MISC, mr * 24, k_shard_ends, nblocks(mr, BLOCK), (const unsigned long long *)advx, (const unsigned long long *)adv, mr, ye);
        uint32_t bad = 0; unsigned long long tot = 0;
        PFP_HIP(c, hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, c->stream));
        PFP_HIP(c, hipMemcpyAsync(&tot, d_tot, 8, hipMemcpyDeviceToHost, c->stream));
        PFP_HIP(c, hipStreamSynchronize(c->stream));
        if (bad || tot != v[r].n + 1 + (uint64_t)w) return PFP_E_CORRUPT;      // Y holds Dollar + (context +) text + w Dollars
        c->arena.release_hi(mk2);
        vye[r] = ye; vlast[r] = la;
    }
    {
        uint64_t goff = 0;
        for (int r = 0; r < nshards; ++r) {
            const uint32_t nx = xfirst[r + 1] - xfirst[r];
            if (nx) {
                PFP_LAUNCH(c, K_MISC, nx * 24, k_merge_extra, nblocks(nx, BLOCK), (const uint32_t *)cand_id + ctot + xfirst[r], (const tpos_t *)d_xye + xfirst[r], (const uint32_t *)d_xlast + xfirst[r], nx, (uint32_t)goff,
                           c->d_pid, c->d_ye, c->d_last, occw);
                goff += nx;
            }
            const uint32_t cnt = (uint32_t)(ib[r] - ia[r]);
            if (cnt) {
                PFP_LAUNCH(c, K_MISC, cnt * 24, k_merge_phrases, nblocks(cnt, BLOCK), v[r].d_pid, vye[r], vlast[r], (uint32_t)ib[r], (uint32_t)ia[r], (uint32_t)goff, coff[r], shift[r],
                           (const uint32_t *)cand_id, (tpos_t)0, 0u, 0, c->d_pid, c->d_ye, c->d_last, occw);
                goff += cnt;
            }
        }
        if (goff != mtot) return PFP_E_CORRUPT;
    }
    PFP_TRY(finish_parse(c, occw));
    PFP_HIP(c, hipStreamSynchronize(c->stream));
    c->arena.release_hi(mk);
    c->stage = 1;
    c->lo_after_parse = c->arena.mark_lo();
    c->stage_ms[0] = timer.ms();
    if (out) { out->n = ntot; out->m = mtot; out->dwords = c->dwords; out->dsize = c->dsize; }
    return PFP_OK;
}

// widen / copy helpers for the U-wide getters
static int get_u32_as(pfp_ctx *c, const uint32_t *d, uint64_t cnt, void *dst, bool u64)
{
    if (!dst || !cnt) return PFP_OK;
    if (!u64) { PFP_HIP(c, hipMemcpy(dst, d, cnt * 4, hipMemcpyDeviceToHost)); return PFP_OK; }
    std::vector<uint32_t> tmp((size_t)cnt);
    PFP_HIP(c, hipMemcpy(tmp.data(), d, cnt * 4, hipMemcpyDeviceToHost));
    uint64_t *o = (uint64_t *)dst;
    for (uint64_t i = 0; i < cnt; ++i) o[i] = tmp[(size_t)i];
    return PFP_OK;
}

static int get_u64_as(pfp_ctx *c, const uint64_t *d, uint64_t cnt, void *dst, bool u64)
{
    if (!dst || !cnt) return PFP_OK;
    if (u64) { PFP_HIP(c, hipMemcpy(dst, d, cnt * 8, hipMemcpyDeviceToHost)); return PFP_OK; }
    std::vector<uint64_t> tmp((size_t)cnt);
    PFP_HIP(c, hipMemcpy(tmp.data(), d, cnt * 8, hipMemcpyDeviceToHost));
    uint32_t *o = (uint32_t *)dst;
    for (uint64_t i = 0; i < cnt; ++i) o[i] = (uint32_t)tmp[(size_t)i];   // uint_t = 32 bit wraps like the reference's build
    return PFP_OK;
}

int pfp_parse_get(pfp_ctx *c, uint8_t *dict, void *occ, uint32_t *parse, uint8_t *last, void *sai)
{
    if (!c) return PFP_E_ARG;
    if (c->stage < 1 || !c->d_sdict) return PFP_E_STATE;
    PFP_HIP(c, hipSetDevice(c->device));
    PFP_HIP(c, hipStreamSynchronize(c->stream));
    const bool u64 = (c->flags & PFP_FLAG_U64) != 0;
    if (dict) PFP_HIP(c, hipMemcpy(dict, c->d_sdict, c->dsize, hipMemcpyDeviceToHost));
    PFP_TRY(get_u32_as(c, c->d_occ, c->dwords, occ, u64));
    if (parse) PFP_HIP(c, hipMemcpy(parse, c->d_parse, c->m * 4, hipMemcpyDeviceToHost));
    if (last) PFP_HIP(c, hipMemcpy(last, c->d_last, c->m, hipMemcpyDeviceToHost));
    PFP_TRY(get_u64_as(c, c->d_ye, c->m, sai, u64));   // sai[j] = pos_ at process_phrase = ye[j]  (pfparser.hpp:600)
    return PFP_OK;
}

static int parse_bwt_impl(pfp_ctx *c);
int pfp_parse_bwt(pfp_ctx *c)
{
    if (!c) return PFP_E_ARG;
    if (c->stage < 1 || !c->d_parse) return PFP_E_STATE;
    if (c->m < 2) return PFP_E_ONE_WORD;                        // pfparser.hpp:390-392
    PFP_HIP(c, hipSetDevice(c->device));
    c->arena.release_lo(c->lo_after_parse);
    ArenaGuard g(c);
    const int rc = g.done(parse_bwt_impl(c));
    if (rc != PFP_OK) { c->stage = 1; c->d_bwlast = nullptr; c->d_ilist = nullptr; c->d_bwsai = nullptr; c->d_bwl_il = nullptr; c->nrows = 0; }
    return rc;
}
static int parse_bwt_impl(pfp_ctx *c)
{
    HostTimer timer;
    const uint64_t m = c->m, N = m + 1;
    const size_t mk = c->arena.mark_hi();
    PFP_LAUNCH(c, K_MISC, 4, k_set_u32, 1, c->d_parse, m, 0u);  // :407-410 (d_parse has m+1 slots)
    uint32_t *SAP, *W, *rowid, *W2, *rowid2, *rk;
    PFP_ALLOC_LO(c, c->d_bwlast, uint8_t, N);
    PFP_ALLOC_LO(c, c->d_ilist, uint32_t, N);
    const bool sai = (c->flags & PFP_FLAG_SAI) != 0;
    if (sai) PFP_ALLOC_LO(c, c->d_bwsai, tpos_t, N); else c->d_bwsai = nullptr;
    PFP_ALLOC_LO(c, c->d_bwl_il, uint8_t, N);
    uint32_t *d_bad; PFP_ALLOC_HI(c, d_bad, uint32_t, 1);
    PFP_HIP(c, hipMemsetAsync(d_bad, 0, 4, c->stream));
    const int pack = N <= (1ULL << BWL_SHIFT) ? 1 : 0;
    PFP_ALLOC_HI(c, SAP, uint32_t, N); PFP_ALLOC_HI(c, rk, uint32_t, N);
    PFP_ALLOC_HI(c, W, uint32_t, N); PFP_ALLOC_HI(c, rowid, uint32_t, N);
    PFP_ALLOC_HI(c, W2, uint32_t, N); PFP_ALLOC_HI(c, rowid2, uint32_t, N);
    int rounds = 0;
    PFP_TRY(sort_int_suffixes(c, c->d_parse, N, c->dwords, SAP, rk, &rounds, 0, false));   // sacak_int, :425 (recsort.h; the ranks are not asked for)
    {
        uint4 *rec; PFP_ALLOC_HI(c, rec, uint4, m);
        PFP_LAUNCH(c, K_PBWT_ROWS, m * 29, k_pbwt_pack, nblocks(m, BLOCK), (const uint32_t *)c->d_parse, (const uint8_t *)c->d_last, sai ? (const tpos_t *)c->d_ye : (const tpos_t *)nullptr, m, rec);
        PFP_LAUNCH(c, K_PBWT_ROWS, N * (4 + 16 + 17), k_pbwt_rows, nblocks(N, BLOCK), (const uint32_t *)SAP, (const uint4 *)rec, m, c->d_bwlast, c->d_bwsai, W, rowid, pack, d_bad);
    }
    // ilist: rows grouped by word, ascending inside a word (:452-462) = stable sort of row ids by word
    BitRange wr = {0, bits_for(c->dwords)};
    uint32_t *sw, *sr;
    PFP_TRY(radix_sort_pairs<uint32_t>(c, W, rowid, W2, rowid2, N, &wr, 1, &sw, &sr));
    uint32_t bad = 0;
    if (pack) PFP_TRY(d2h_u32(c, d_bad, &bad));
    if (pack && !bad) PFP_LAUNCH(c, K_PBWT_ROWS, N * 9, k_ilist_split, nblocks(N, BLOCK), (const uint32_t *)sr, N, c->d_ilist, c->d_bwl_il);
    else PFP_LAUNCH(c, K_PBWT_ROWS, N * 10, k_ilist_gather, nblocks(N, BLOCK), (const uint32_t *)sr, N, pack ? (1u << BWL_SHIFT) - 1u : 0xFFFFFFFFu, (const uint8_t *)c->d_bwlast, c->d_ilist, c->d_bwl_il);
    PFP_HIP(c, hipStreamSynchronize(c->stream));
    c->arena.release_hi(mk);
    c->nrows = N; c->stage = 2;
    c->lo_after_pbwt = c->arena.mark_lo();
    c->stage_ms[1] = timer.ms();
    return PFP_OK;
}

int pfp_parse_bwt_get(pfp_ctx *c, uint8_t *bwlast, void *ilist, void *bwsai)
{
    if (!c) return PFP_E_ARG;
    if (c->stage < 2) return PFP_E_STATE;
    PFP_HIP(c, hipSetDevice(c->device));
    const bool u64 = (c->flags & PFP_FLAG_U64) != 0;
    if (bwlast) PFP_HIP(c, hipMemcpy(bwlast, c->d_bwlast, c->nrows, hipMemcpyDeviceToHost));
    PFP_TRY(get_u32_as(c, c->d_ilist, c->nrows, ilist, u64));
    if (bwsai) { if (!c->d_bwsai) return PFP_E_STATE; PFP_TRY(get_u64_as(c, c->d_bwsai, c->nrows, bwsai, u64)); }
    return PFP_OK;
}

// ---- stage 2 --------------------------------------------------------------------------------------
static int upload_u32_from(pfp_ctx *c, const void *src, uint64_t cnt, bool u64, uint32_t *d)
{
    if (!u64) { PFP_HIP(c, hipMemcpy(d, src, cnt * 4, hipMemcpyHostToDevice)); return PFP_OK; }
    std::vector<uint32_t> tmp((size_t)cnt);
    const uint64_t *s = (const uint64_t *)src;
    for (uint64_t i = 0; i < cnt; ++i) { if (s[i] > 0xFFFFFFFFULL) return PFP_E_TOO_LARGE; tmp[(size_t)i] = (uint32_t)s[i]; }
    PFP_HIP(c, hipMemcpy(d, tmp.data(), cnt * 4, hipMemcpyHostToDevice));
    return PFP_OK;
}

// consistency of a loaded file set (the emission gathers ilist[F[rank] + r], bwsai[q], bwlast[q] with these values as
// indices: a truncated or mismatched set must become PFP_E_CORRUPT here, not an out-of-bounds device read there)
__global__ __launch_bounds__(BLOCK) void k_load_check(const uint32_t *ilist, uint64_t nrows, const uint32_t *occ, uint64_t dwords, unsigned long long *out /*[0] sum of occ, [1] ilist entries >= nrows*/)
{
    __shared__ unsigned long long red[4];
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    unsigned long long tot;
    (void)block_excl_sum((unsigned long long)(i < dwords ? occ[i] : 0u), red, &tot);
    if (threadIdx.x == 0 && tot) atomicAdd(&out[0], tot);
    (void)block_excl_sum((unsigned long long)((i < nrows && ilist[i] >= nrows) ? 1u : 0u), red, &tot);
    if (threadIdx.x == 0 && tot) atomicAdd(&out[1], tot);
}

static int bwt_load_impl(pfp_ctx *c, const uint8_t *dict, uint64_t dsize, const void *occ, uint64_t dwords,
                         const uint8_t *bwlast, const void *ilist, const void *bwsai, uint64_t nrows, uint64_t n_hint);
int pfp_bwt_load(pfp_ctx *c, const uint8_t *dict, uint64_t dsize, const void *occ, uint64_t dwords,
                 const uint8_t *bwlast, const void *ilist, const void *bwsai, uint64_t nrows, uint64_t n_hint)
{
    if (!c || !dict || !occ || !bwlast || !ilist || dsize < 2 || dwords < 1 || nrows < 2) return PFP_E_ARG;
    if (dsize + 64 >= 0xFFFFFFFFULL || nrows + 64 >= 0xFFFFFFFFULL) return PFP_E_TOO_LARGE;
    if (dict[dsize - 1] != EndOfDict || dict[dsize - 2] != EndOfWord) return PFP_E_CORRUPT;       // pfbwt_io.hpp:71-82
    if (n_hint && nrows > n_hint + 1) return PFP_E_CORRUPT;                                       // more phrases than text positions
    PFP_HIP(c, hipSetDevice(c->device));
    reset_results(c);
    const int rc = bwt_load_impl(c, dict, dsize, occ, dwords, bwlast, ilist, bwsai, nrows, n_hint);
    if (rc != PFP_OK) reset_results(c);
    return rc;
}
static int bwt_load_impl(pfp_ctx *c, const uint8_t *dict, uint64_t dsize, const void *occ, uint64_t dwords,
                         const uint8_t *bwlast, const void *ilist, const void *bwsai, uint64_t nrows, uint64_t n_hint)
{
    // n (the .n file, src/pfbwt-f.cpp:282-285) sizes the outputs; without it assume n <= 4 * dsize
    PFP_TRY(ensure_arena(c, (n_hint ? n_hint : 4 * dsize) + dsize + nrows));
    c->arena.reset();
    const bool u64 = (c->flags & PFP_FLAG_U64) != 0;
    c->dsize = dsize; c->dwords = dwords; c->nrows = nrows; c->m = nrows - 1;
    PFP_ALLOC_LO(c, c->d_dict, uint8_t, dsize + 16);
    PFP_ALLOC_LO(c, c->d_wordid, uint32_t, dsize);
    PFP_ALLOC_LO(c, c->d_ws, uint32_t, dwords + 2);
    PFP_ALLOC_LO(c, c->d_occ, uint32_t, dwords);
    PFP_ALLOC_LO(c, c->d_bwlast, uint8_t, nrows);
    PFP_ALLOC_LO(c, c->d_ilist, uint32_t, nrows);
    PFP_HIP(c, hipMemcpy(c->d_dict, dict, dsize, hipMemcpyHostToDevice));
    PFP_HIP(c, hipMemcpy(c->d_bwlast, bwlast, nrows, hipMemcpyHostToDevice));
    PFP_TRY(upload_u32_from(c, occ, dwords, u64, c->d_occ));
    PFP_TRY(upload_u32_from(c, ilist, nrows, u64, c->d_ilist));
    if (bwsai) {
        PFP_ALLOC_LO(c, c->d_bwsai, tpos_t, nrows);
        if (u64) PFP_HIP(c, hipMemcpy(c->d_bwsai, bwsai, nrows * 8, hipMemcpyHostToDevice));
        else {
            std::vector<uint64_t> tmp((size_t)nrows);
            for (uint64_t i = 0; i < nrows; ++i) tmp[(size_t)i] = ((const uint32_t *)bwsai)[i];
            PFP_HIP(c, hipMemcpy(c->d_bwsai, tmp.data(), nrows * 8, hipMemcpyHostToDevice));
        }
    }
    // word index of every dictionary offset = number of EndOfWord bytes before it (dict_idx.rank, pfbwt.hpp:83-85)
    const size_t mk = c->arena.mark_hi();
    uint32_t *flag, *d_cnt;
    PFP_ALLOC_HI(c, flag, uint32_t, dsize); PFP_ALLOC_HI(c, d_cnt, uint32_t, 1);
    PFP_LAUNCH(c, K_MISC, dsize * 5, k_eow_flags, nblocks(dsize, BLOCK), (const uint8_t *)c->d_dict, dsize, flag);
    PFP_TRY((device_scan<uint32_t, 0>(c, flag, c->d_wordid, dsize, d_cnt)));
    uint32_t nw = 0; PFP_TRY(d2h_u32(c, d_cnt, &nw));
    if (nw != dwords) return PFP_E_CORRUPT;
    PFP_LAUNCH(c, K_MISC, dsize * 5, k_ws_from_flags, nblocks(dsize, BLOCK), (const uint8_t *)c->d_dict, dsize, (const uint32_t *)c->d_wordid, c->d_ws);
    {   // sum(occ) + 1 == rows of the parse BWT (pfparser.hpp:452-462), every ilist entry is a row
        unsigned long long *d_chk, chk[2];
        PFP_ALLOC_HI(c, d_chk, unsigned long long, 2);
        PFP_HIP(c, hipMemsetAsync(d_chk, 0, 16, c->stream));
        PFP_LAUNCH(c, K_MISC, (nrows + dwords) * 4, k_load_check, nblocks(nrows > dwords ? nrows : dwords, BLOCK), (const uint32_t *)c->d_ilist, nrows, (const uint32_t *)c->d_occ, dwords, d_chk);
        PFP_HIP(c, hipMemcpyAsync(chk, d_chk, 16, hipMemcpyDeviceToHost, c->stream));
        PFP_HIP(c, hipStreamSynchronize(c->stream));
        if (chk[0] + 1 != nrows || chk[1] != 0) return PFP_E_CORRUPT;
    }
    PFP_HIP(c, hipStreamSynchronize(c->stream));
    c->arena.release_hi(mk);
    c->d_wrank = nullptr; c->gsa_valid = false; c->stage = 2; c->n = n_hint;     // n known: the emission checks that it produces exactly n + 1 rows
    c->lo_after_pbwt = c->arena.mark_lo();
    return PFP_OK;
}

extern "C++" {
__global__ __launch_bounds__(BLOCK) void k_flag_special(const uint8_t *s_fl, uint64_t dsize, uint32_t *flag)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < dsize) flag[i] = slot_is_special(s_fl[i]) ? 1u : 0u;
}
template <typename EBT> __global__ __launch_bounds__(BLOCK) void k_gather_counts(const EBT *cnt, const uint32_t *list, uint64_t n, EBT *out)
{
    const uint64_t j = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (j < n) out[j] = cnt[list[j]];
}

// Emission of the rows [s0, s1) of this slice (all rows when nslices == 1), in windows of at most `chunk_rows` rows so
// that the per-window scratch (parse rows of the enumerated rows, run counts) stays bounded for texts of tens of
// Gbases.  Rows of a group of equal suffixes that straddles a window boundary are enumerated for both windows.
// ea.special != 0 (no full SA wanted): run-aware emission -- k_fill writes every row as a run of its slot's preceding
// byte, k_emit walks only the rows of the special slots (tot2 of them), samples look their parse rows up.
template <typename SAT, typename EBT> static int emit_and_sample(pfp_ctx *c, EmitArgs ea, bool want_sa, bool want_rssa, int slice, int nslices, uint64_t tot2)
{
    const uint64_t total = ea.nout;
    const uint64_t s0 = total / (uint64_t)nslices * (uint64_t)slice + (total % (uint64_t)nslices) * (uint64_t)slice / (uint64_t)nslices;
    const uint64_t s1 = slice + 1 == nslices ? total : total / (uint64_t)nslices * (uint64_t)(slice + 1) + (total % (uint64_t)nslices) * (uint64_t)(slice + 1) / (uint64_t)nslices;
    const uint64_t lead = s0 ? 1 : 0;                     // the row in front of the slice (run detection needs its BWT byte)
    const uint64_t nrows = s1 - s0;
    c->slice_begin = s0; c->slice_rows = nrows;
    const uint64_t chunk_rows = c->tun.emit_chunk_rows ? c->tun.emit_chunk_rows : (3ULL << 30);
    const uint64_t nchunks = (nrows + chunk_rows - 1) / chunk_rows;
    const bool windowed = nslices > 1 || nchunks > 1;
    const bool runaware = ea.special != 0;
    // the byte of output row o lives at (bwtbuf - (s0 - lead)) + o, and that address is congruent to o modulo 16 (k_fill stores 16 aligned rows at a time)
    uint8_t *bwtraw; PFP_ALLOC_LO(c, bwtraw, uint8_t, nrows + lead + 48);
    uint8_t *bwtbuf = bwtraw + ((s0 - lead) & 15);
    c->d_bwt = bwtbuf + lead;
    const bool keep_sa = want_sa;                          // a full SA array for this slice lives in the arena
    SAT *sabuf = nullptr;
    if (keep_sa) PFP_ALLOC_LO(c, sabuf, SAT, nrows + lead);
    c->d_sa = sabuf ? sabuf + lead : nullptr;
    c->d_ssa = c->d_esa = nullptr;
    unsigned long long *d_b; PFP_ALLOC_HI(c, d_b, unsigned long long, 6);
    // windows: rows [cs - cl, ce) are written, [e0, e1) (all rows) resp. [q0, q1) (special rows) are enumerated for them
    struct Win { uint64_t cs, ce, cl, e0, e1, q0, q1; };
    std::vector<Win> wins((size_t)nchunks);
    for (uint64_t ch = 0; ch < nchunks; ++ch) {
        Win &wn = wins[(size_t)ch];
        wn.cs = s0 + ch * chunk_rows; wn.ce = (wn.cs + chunk_rows < s1) ? wn.cs + chunk_rows : s1; wn.cl = wn.cs ? 1 : 0;
        wn.e0 = 0; wn.e1 = total; wn.q0 = 0; wn.q1 = tot2;
    }
    if (windowed) {
        unsigned long long *d_bounds; PFP_ALLOC_HI(c, d_bounds, unsigned long long, 4 * nchunks);
        for (uint64_t ch = 0; ch < nchunks; ++ch)
            PFP_LAUNCH(c, K_MISC, 64, (k_slice_bounds<EBT>), 1, ea, wins[(size_t)ch].cs - wins[(size_t)ch].cl, wins[(size_t)ch].ce, d_bounds + 4 * ch);
        std::vector<unsigned long long> hb(4 * (size_t)nchunks);
        PFP_HIP(c, hipMemcpyAsync(hb.data(), d_bounds, hb.size() * 8, hipMemcpyDeviceToHost, c->stream));
        PFP_HIP(c, hipStreamSynchronize(c->stream));
        for (uint64_t ch = 0; ch < nchunks; ++ch) { Win &wn = wins[(size_t)ch]; wn.e0 = hb[4 * ch]; wn.e1 = hb[4 * ch + 1]; wn.q0 = hb[4 * ch + 2]; wn.q1 = hb[4 * ch + 3]; }
    }
    uint64_t maxq = 0, maxrows = 0;
    for (const Win &wn : wins) { if (wn.q1 - wn.q0 > maxq) maxq = wn.q1 - wn.q0; if (wn.ce - wn.cs > maxrows) maxrows = wn.ce - wn.cs; }
    // list of the rows of many-member groups (sorted per window instead of ranked row by row)
    uint64_t *bk0 = nullptr, *bk1 = nullptr; uint32_t *bv0 = nullptr, *bv1 = nullptr, *btg = nullptr;
    if (ea.big_total) {
        if (ea.big_total >= 0xFFFFFFF0ULL) return PFP_E_TOO_LARGE;
        PFP_ALLOC_HI(c, bk0, uint64_t, ea.big_total); PFP_ALLOC_HI(c, bk1, uint64_t, ea.big_total);
        PFP_ALLOC_HI(c, bv0, uint32_t, ea.big_total); PFP_ALLOC_HI(c, bv1, uint32_t, ea.big_total); PFP_ALLOC_HI(c, btg, uint32_t, ea.big_total);
    }
    ea.big_keys = bk0; ea.big_vals = bv0; ea.big_count = d_b + 4; ea.big_cap = ea.big_total;
    // group-stationary route of the special rows (k_emit_groups); what it leaves behind is marked for k_emit
    ea.gleft = ea.tile_left = nullptr; ea.group_rows_cap = c->tun.emit_group_rows; ea.rank_members_max = c->tun.big_group_members != -2 ? 0xFFFFFFFFu : c->tun.emit_group_rows >= (uint32_t)EG_BUF ? BIG_GROUP_MEMBERS : 3u;      // (tests with small batches: groups of more than three members through the LDS sort)
    const uint64_t max_etiles = maxq / EMIT_TILE + 3;
    if (runaware && ea.group_rows_cap && tot2 && ea.cinfo) {
        PFP_ALLOC_HI(c, ea.gleft, uint8_t, (size_t)ea.ecount + 1); PFP_ALLOC_HI(c, ea.tile_left, uint8_t, max_etiles);
        PFP_HIP(c, hipMemsetAsync(ea.gleft, 0, (size_t)ea.ecount + 1, c->stream));      // which groups are left is a property of the build, not of a window
    }
    ea.lglist = nullptr; ea.lgcount = nullptr; ea.lgcap = 0; ea.qpasses = (bits_for(c->nrows ? c->nrows : 0xFFFFFFFFULL) + 7) / 8;
    if (ea.gleft) {
        const uint32_t cap = ea.group_rows_cap < (uint32_t)EG_BUF ? ea.group_rows_cap : (uint32_t)EG_BUF;
        ea.lgcap = maxq / cap + 2;
        PFP_ALLOC_HI(c, ea.lglist, uint32_t, 2 * ea.lgcap); PFP_ALLOC_HI(c, ea.lgcount, unsigned long long, 2);
    }
    ea.gstat = nullptr;
    if (ea.gleft && c->tun.verbose) { PFP_ALLOC_HI(c, ea.gstat, unsigned long long, 12); PFP_HIP(c, hipMemsetAsync(ea.gstat, 0, 96, c->stream)); }
    struct GStat { pfp_ctx *c; EmitArgs &ea; ~GStat() {
        if (!ea.gstat) return;
        unsigned long long h[12];
        if (hipMemcpy(h, ea.gstat, 96, hipMemcpyDeviceToHost) != hipSuccess) return;
        fprintf(stderr, "[pfbwt_hip] special rows left to the row-wise kernel (counted once per window that enumerates them): whole-word member %llu, sort route %llu, too many rows %llu, too many slots %llu; by group rows (<1K <4K <16K <64K <256K <1M <4M more):",
                h[0], h[1], h[2], h[3]);
        for (int b = 0; b < 8; ++b) fprintf(stderr, " %llu", h[4 + b]);
        fprintf(stderr, "\n");
    } } gstat_print{c, ea};
    const BitRange big_ranges[2] = {{0, bits_for(c->nrows)}, {32, 32 + bits_for(ea.dsize)}};
    const uint32_t fill_subs_env = c->tun.fill_subs;     // super-tiles (4 x 4096 rows) per workgroup
    const uint32_t fill_subs = fill_subs_env < 1u ? 1u : fill_subs_env > FILL_MAX_SUBS ? FILL_MAX_SUBS : fill_subs_env;
    // emits the rows whose output position lies in [cs - cl, ce); bwt_at / sa_at point at that first position; q_at receives
    // the parse row of every row written (all rows) resp. of every special row enumerated (run-aware)
    auto emit_window = [&](const Win &wn, uint8_t *bwt_at, SAT *sa_at, uint32_t *q_at, bool fill) -> int {
        ea.w0 = wn.cs - wn.cl; ea.w1 = wn.ce;
        const uint64_t rows = ea.w1 - ea.w0;
        if (ea.big_total) PFP_HIP(c, hipMemsetAsync(ea.big_count, 0, 16, c->stream));
        if (runaware) {
            ea.e0 = wn.q0; ea.e1 = wn.q1; ea.q0 = wn.q0;
            if (fill) {
                const uint64_t super = (uint64_t)FILL_GROUPS * FILL_SUB;
                const uint64_t nsub = (ea.w1 - 1) / super - ea.w0 / super + 1;
                PFP_LAUNCH(c, K_FILL, rows, (k_fill<EBT>), nblocks(nsub, fill_subs), ea, bwt_at, fill_subs);
            }
            if (ea.e1 > ea.e0) {
                const unsigned ge = (unsigned)((ea.e1 - 1) / EMIT_TILE - ea.e0 / EMIT_TILE + 1);
                if (ea.gleft) {
                    PFP_HIP(c, hipMemsetAsync(ea.tile_left, 0, (size_t)ge, c->stream));
                    PFP_HIP(c, hipMemsetAsync(ea.lgcount, 0, 16, c->stream));
                    PFP_LAUNCH(c, K_EMIT, (ea.e1 - ea.e0) * (1 + 4 + (q_at ? 4 : 0)), (k_emit_groups<EBT>), ge, ea, bwt_at, q_at);
                    PFP_LAUNCH(c, K_EMIT_LARGE, 0, (k_emit_groups_large<EBT, EG1_BUF>), ge < 1024u ? ge : 1024u, ea, bwt_at, q_at);      // the groups too long for a batch, one per workgroup turn
                    PFP_LAUNCH(c, K_EMIT_LARGE, 0, (k_emit_groups_large<EBT, EG2_BUF>), ge < 512u ? ge : 512u, ea, bwt_at, q_at);
                    PFP_LAUNCH(c, K_EMIT_BIG, 0, (k_emit<SAT, EBT>), ge, ea, bwt_at, (SAT *)nullptr, q_at);      // the groups left over (workgroups of other tiles return at once)
                } else
                PFP_LAUNCH(c, K_EMIT, (ea.e1 - ea.e0) * (1 + 4 + (q_at ? 4 : 0)), (k_emit<SAT, EBT>), ge, ea, bwt_at, (SAT *)nullptr, q_at);
            }
        } else {
            ea.e0 = wn.e0; ea.e1 = wn.e1; ea.q0 = 0;
            PFP_LAUNCH(c, K_EMIT, rows * (1 + 4 + (sa_at ? 8 + sizeof(SAT) : 0) + (q_at ? 4 : 0)),   // per row: BWT byte out, ilist entry in, (bwsai gather + SA out | parse row out)
                       (k_emit<SAT, EBT>), (unsigned)((ea.e1 - 1) / EMIT_TILE - ea.e0 / EMIT_TILE + 1), ea, bwt_at, sa_at, q_at);
        }
        if (ea.big_total) {
            unsigned long long hb[2];
            PFP_HIP(c, hipMemcpyAsync(hb, ea.big_count, 16, hipMemcpyDeviceToHost, c->stream));
            PFP_HIP(c, hipStreamSynchronize(c->stream));
            if (hb[1] || hb[0] > ea.big_total) return PFP_E_CORRUPT;
            const uint64_t nb = hb[0];
            if (nb) {
                uint64_t *sk; uint32_t *sv;
                PFP_TRY(radix_sort_pairs<uint64_t>(c, bk0, bv0, bk1, bv1, nb, big_ranges, 2, &sk, &sv));
                PFP_LAUNCH(c, K_EMIT_BIG, nb * 12, k_big_heads, nblocks(nb, BLOCK), (const uint64_t *)sk, nb, btg);
                PFP_TRY((device_scan<uint32_t, 1>(c, btg, btg, nb, nullptr)));
                PFP_LAUNCH(c, K_EMIT_BIG, nb * (30 + sizeof(SAT)), (k_big_place<SAT, EBT>), nblocks(nb, BLOCK), ea, (const uint64_t *)sk, (const uint32_t *)sv, (const uint32_t *)btg, nb, bwt_at, sa_at, q_at);
            }
        }
        return PFP_OK;
    };
    auto bwt_of = [&](const Win &wn) -> uint8_t * { return bwtbuf + (wn.cs - wn.cl - (s0 - lead)); };   // position cs - cl
    // pfp_bwt_build_stream: the rows of a finished window start their way to the host while the next window is emitted.  Window k + 1
    // WRITES row cs - 1, the last row of window k, once more (k_fill puts a slot's placeholder byte there before k_emit_groups / k_emit
    // restore the true one; run detection needs that row): a copy of window k that included it could deliver the placeholder.
    // So every window sends its rows shifted by one -- [cs - cl, ce - 1), the last window up to ce -- and no row is in flight while a
    // later window's kernels can still store to it.  (The row in front of a slice belongs to the neighbouring slice's buffers.)
    std::vector<hipEvent_t> wev;
    auto stream_out = [&](const Win &wn) -> int {
        if (!c->h_bwt && !c->h_sa) return PFP_OK;
        const uint64_t r0 = wn.cs - wn.cl < s0 ? s0 : wn.cs - wn.cl, r1 = wn.ce == s1 ? wn.ce : wn.ce - 1;
        if (r1 <= r0) return PFP_OK;
        hipEvent_t e; PFP_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming)); wev.push_back(e);
        PFP_HIP(c, hipEventRecord(e, c->stream));
        PFP_HIP(c, hipStreamWaitEvent(c->fa.copy, e, 0));
        const uint64_t rows = r1 - r0;
        if (c->h_bwt) PFP_HIP(c, hipMemcpyAsync(c->h_bwt + (r0 - s0), bwtbuf + (r0 - (s0 - lead)), (size_t)rows, hipMemcpyDeviceToHost, c->fa.copy));
        if (c->h_sa && sabuf) PFP_HIP(c, hipMemcpyAsync((char *)c->h_sa + (r0 - s0) * sizeof(SAT), sabuf + (r0 - (s0 - lead)), (size_t)rows * sizeof(SAT), hipMemcpyDeviceToHost, c->fa.copy));
        return PFP_OK;
    };
    struct EvGuard { std::vector<hipEvent_t> &v; ~EvGuard() { for (auto e : v) (void)hipEventDestroy(e); } } evguard{wev};
    const uint64_t qcap = runaware ? maxq + 1 : maxrows + 1;     // parse rows kept per window
    bool bwt_done = false;
    if (want_rssa && !keep_sa) {
        // Samples only: ONE pass per window -- emit the BWT bytes and the parse rows q needed for sampling into scratch, find
        // the run starts, compute SA values for the 2r sampled rows only (k_sample_rows, k_sample_values), forget the q's.  r is not known in
        // advance, so the sample arrays get a capacity from the free workspace; if r exceeds it the exact two-pass
        // route below is taken.
        const size_t lo_mark = c->arena.mark_lo(), hi_mark = c->arena.mark_hi();
        const uint64_t maxtiles = nblocks(maxrows, RUN_TILE);
        uint32_t *tilecnt, *tilebase, *d_cnt, *qtmp; uint16_t *rmask;
        PFP_ALLOC_HI(c, tilecnt, uint32_t, maxtiles); PFP_ALLOC_HI(c, tilebase, uint32_t, maxtiles); PFP_ALLOC_HI(c, d_cnt, uint32_t, 1);
        PFP_ALLOC_HI(c, rmask, uint16_t, maxtiles * BLOCK);
        PFP_ALLOC_HI(c, qtmp, uint32_t, qcap);
        const size_t freeb = c->arena.hi > c->arena.lo + ((size_t)256 << 20) ? c->arena.hi - c->arena.lo - ((size_t)256 << 20) : 0;
        uint64_t cap = freeb / (4 * sizeof(SAT));
        if (cap > nrows) cap = nrows;
        if (c->tun.sample_cap < cap) cap = c->tun.sample_cap;   // tests: force the fallback
        SAT *samp = nullptr;
        if (cap) { samp = (SAT *)c->arena.reserve_lo(sizeof(SAT) * (4 * cap + 4)); if (!samp) return PFP_E_NOMEM; }      // address space for the worst case; committed window by window
        SAT *ssa = samp, *esa = samp ? samp + 2 * cap : nullptr;
        SAT *esa_w = esa ? esa + 2 * lead : nullptr;      // slices > 0: the first run start of the slice closes a run of the previous slice
        uint64_t run_base = 0; bool overflow = cap == 0;
        ea.qspec = qtmp;
        for (uint64_t ch = 0; ch < nchunks; ++ch) {
            const Win &wn = wins[(size_t)ch];
            const uint64_t rows = wn.ce - wn.cs;
            uint8_t *bw = bwt_of(wn) + wn.cl;                                // first row of the window
            PFP_TRY(emit_window(wn, bw - wn.cl, (SAT *)nullptr, qtmp, true));
            PFP_TRY(stream_out(wn));
            const uint64_t ntiles = nblocks(rows, RUN_TILE);
            PFP_LAUNCH(c, K_RUNS, rows, k_run_tile_count, ntiles, (const uint8_t *)bw, rows, (int)wn.cl, tilecnt, rmask);
            PFP_TRY((device_scan<uint32_t, 0>(c, tilecnt, tilebase, ntiles, d_cnt)));
            uint32_t rc = 0; PFP_TRY(d2h_u32(c, d_cnt, &rc));
            if (!overflow && run_base + rc > cap) overflow = true;
            if (!overflow && !(c->arena.commit_range(ssa + 2 * run_base, sizeof(SAT) * (2 * (size_t)rc + 4)) && c->arena.commit_range(esa + 2 * run_base, sizeof(SAT) * (2 * (size_t)rc + 8)))) overflow = true;
            if (!overflow) {
                const bool last = wn.ce == total;
                PFP_LAUNCH(c, K_SAMPLES, rows / 8 + (uint64_t)rc * 4 * sizeof(SAT), (k_sample_rows<SAT>), nblocks(ntiles, SR_TILES), (const uint16_t *)rmask, rows, (uint64_t)ntiles, (const uint32_t *)tilebase, wn.cs, run_base, total,
                           last ? run_base + rc + 1 : (uint64_t)0, ssa, esa_w);
                PFP_LAUNCH(c, K_SAMPLES, (uint64_t)rc * (2 * 60 + 4 * sizeof(SAT)), (k_sample_values<SAT, EBT>), nblocks((uint64_t)rc + 1, BLOCK), ea, (const SAT *)nullptr, (const uint32_t *)qtmp, wn.cs - wn.cl, (uint64_t)rc, run_base,
                           (int)last, last ? (uint64_t)(run_base + rc - 1) : (uint64_t)0, ssa, esa_w);
            }
            run_base += rc;
        }
        c->runs = run_base; c->esa_pairs = run_base - (s0 == 0 ? 1 : 0) + (s1 == total ? 1 : 0);
        c->arena.release_hi(hi_mark);
        if (!overflow) {
            c->d_ssa = ssa; c->d_esa = esa;
            c->arena.release_lo(c->arena.offset_of(esa) + sizeof(SAT) * (2 * (size_t)run_base + 8));      // what lies behind the run ends that were written is free again
            return PFP_OK;
        }
        c->arena.release_lo(lo_mark);      // fall through: BWT bytes are complete, samples are redone with exact sizes
        bwt_done = true;
    } else {
    // pass 1: BWT bytes (and SA values if a full SA is kept)
    for (uint64_t ch = 0; ch < nchunks; ++ch) {
        const Win &wn = wins[(size_t)ch];
        PFP_TRY(emit_window(wn, bwt_of(wn), sabuf ? sabuf + (wn.cs - wn.cl - (s0 - lead)) : (SAT *)nullptr, (uint32_t *)nullptr, true));
        PFP_TRY(stream_out(wn));
    }
    bwt_done = true;
    // runs (src/pfbwt-f.cpp:304-305): runs that start in this slice
    {
        unsigned long long *d_runs = d_b + 2;
        PFP_HIP(c, hipMemsetAsync(d_runs, 0, 8, c->stream));
        PFP_LAUNCH(c, K_RUNS, nrows, k_run_count, nblocks(nrows, 16 * BLOCK), (const uint8_t *)c->d_bwt, nrows, (int)lead, d_runs);
        unsigned long long r = 0;
        PFP_HIP(c, hipMemcpyAsync(&r, d_runs, 8, hipMemcpyDeviceToHost, c->stream));
        PFP_HIP(c, hipStreamSynchronize(c->stream));
        c->runs = r; c->esa_pairs = r - (s0 == 0 ? 1 : 0) + (s1 == total ? 1 : 0);
    }
    }
    if (want_rssa) {   // .ssa / .esa samples (pfbwt-f.cpp:306-315, 325-328) of the runs that start in this slice
        const uint64_t r = c->runs;
        if (sabuf) {      // with a full SA the samples need nothing of the per-slot arrays any more (~60 B per dictionary byte): on a
            PFP_HIP(c, hipStreamSynchronize(c->stream));      // non-repetitive genome (S-3G: r = 0.74 n) they and the samples do not fit together
            c->arena.release_hi(c->emit_scratch_mark);
        }
        SAT *ssa, *esa;
        PFP_ALLOC_LO(c, ssa, SAT, 2 * r + 2); PFP_ALLOC_LO(c, esa, SAT, 2 * r + 4);
        c->d_ssa = ssa; c->d_esa = esa;
        SAT *esa_w = esa + 2 * lead;
        const uint64_t maxtiles = nblocks(maxrows, RUN_TILE);
        uint32_t *tilecnt, *tilebase, *d_cnt, *qtmp = nullptr; uint16_t *rmask;
        PFP_ALLOC_HI(c, tilecnt, uint32_t, maxtiles); PFP_ALLOC_HI(c, tilebase, uint32_t, maxtiles); PFP_ALLOC_HI(c, d_cnt, uint32_t, 1);
        PFP_ALLOC_HI(c, rmask, uint16_t, maxtiles * BLOCK);
        if (!sabuf) PFP_ALLOC_HI(c, qtmp, uint32_t, qcap);
        ea.qspec = qtmp;
        uint64_t run_base = 0;
        for (uint64_t ch = 0; ch < nchunks; ++ch) {
            const Win &wn = wins[(size_t)ch];
            const uint64_t rows = wn.ce - wn.cs;
            uint8_t *bw = bwt_of(wn) + wn.cl;                                // first row of the chunk
            if (!sabuf) PFP_TRY(emit_window(wn, bw - wn.cl, (SAT *)nullptr, qtmp, !bwt_done));   // pass 2 of this window: the same rows again, now with their q
            const uint64_t ntiles = nblocks(rows, RUN_TILE);
            PFP_LAUNCH(c, K_RUNS, rows, k_run_tile_count, ntiles, (const uint8_t *)bw, rows, (int)wn.cl, tilecnt, rmask);
            PFP_TRY((device_scan<uint32_t, 0>(c, tilecnt, tilebase, ntiles, d_cnt)));
            uint32_t rc = 0; PFP_TRY(d2h_u32(c, d_cnt, &rc));
            if (run_base + rc > r) return PFP_E_CORRUPT;
            const bool last = wn.ce == total;
            PFP_LAUNCH(c, K_SAMPLES, rows / 8 + (uint64_t)rc * 4 * sizeof(SAT), (k_sample_rows<SAT>), nblocks(ntiles, SR_TILES), (const uint16_t *)rmask, rows, (uint64_t)ntiles, (const uint32_t *)tilebase, wn.cs, run_base, total, last ? r + 1 : (uint64_t)0, ssa, esa_w);
            PFP_LAUNCH(c, K_SAMPLES, (uint64_t)rc * (2 * 60 + 4 * sizeof(SAT)), (k_sample_values<SAT, EBT>), nblocks((uint64_t)rc + 1, BLOCK), ea, sabuf ? (const SAT *)(sabuf + (wn.cs - wn.cl - (s0 - lead))) : (const SAT *)nullptr,
                       (const uint32_t *)qtmp, wn.cs - wn.cl, (uint64_t)rc, run_base, (int)last, last ? (uint64_t)(r - 1) : (uint64_t)0, ssa, esa_w);
            run_base += rc;
        }
        if (run_base != r) return PFP_E_CORRUPT;
    }
    return PFP_OK;
}

// everything of stage 2 that depends on the width of the row counter (EBT = uint32_t while n + 1 < 2^32)
template <typename EBT> static int emit_stage(pfp_ctx *c, EmitArgs ea, int want_sa, int want_rssa, int slice, int nslices)
{
    const uint64_t dsize = c->dsize;
    const bool no_runaware = c->tun.no_runaware != 0;      // tests / measurements: every row enumerated, as with a full SA
    const bool runaware = !want_sa && !no_runaware;
    EBT *cnt, *EB, *d_tot, *cnt2 = nullptr; unsigned long long *d_hard;
    PFP_ALLOC_HI(c, cnt, EBT, dsize); PFP_ALLOC_HI(c, EB, EBT, dsize); PFP_ALLOC_HI(c, d_hard, unsigned long long, 2); PFP_ALLOC_HI(c, d_tot, EBT, 2);
    PFP_HIP(c, hipMemsetAsync(d_hard, 0, 16, c->stream));
    ea.EB = EB;
    uint32_t *s_g0, *gk, *gqf = nullptr, *gql = nullptr; uint8_t *gfl, *gnu;
    PFP_ALLOC_HI(c, s_g0, uint32_t, dsize); PFP_ALLOC_HI(c, gk, uint32_t, dsize); PFP_ALLOC_HI(c, gfl, uint8_t, dsize); PFP_ALLOC_HI(c, gnu, uint8_t, dsize);
    PFP_HIP(c, hipMemsetAsync(gfl, 0, dsize, c->stream));
    PFP_HIP(c, hipMemsetAsync(gnu, 0, dsize, c->stream));
    if (runaware) {
        PFP_ALLOC_HI(c, cnt2, EBT, dsize); PFP_ALLOC_HI(c, gqf, uint32_t, dsize); PFP_ALLOC_HI(c, gql, uint32_t, dsize);
        PFP_HIP(c, hipMemsetAsync(gqf, 0xFF, dsize * 4, c->stream));
        PFP_HIP(c, hipMemsetAsync(gql, 0, dsize * 4, c->stream));
    }
    ea.s_g0 = s_g0; ea.gk = gk; ea.cnt = cnt; ea.gqf = gqf; ea.gql = gql;
    {   // per dictionary offset, for k_emit_slots only: ONE 16-byte record (suffix lengths must fit 26 bits), else word id | preceding
        // byte + class head and a second gather of the word record.  The array lives only for that kernel: what is allocated behind
        // it (the 16-byte per-slot records of the row kernels) takes its place -- same stream, so the reuse is ordered
        const size_t mkp = c->arena.mark_hi();
        if (ea.use_prec) {
            uint4 *prec; PFP_ALLOC_HI(c, prec, uint4, dsize);
            PFP_LAUNCH(c, K_EMIT_COUNT, dsize * 29, k_pack_prec, nblocks(dsize, BLOCK), ea.D, ea.wordid, ea.winfo, dsize, ea.dwords, prec, ea.use_e0 ? ea.bwsai_il : (const tpos_t *)nullptr);
            ea.prec = prec;
        } else {
            uint2 *posinfo; PFP_ALLOC_HI(c, posinfo, uint2, dsize);
            PFP_LAUNCH(c, K_EMIT_COUNT, dsize * 17, k_pack_posinfo, nblocks(dsize, BLOCK), ea.D, ea.wordid, dsize, posinfo);
            ea.posinfo = posinfo;
        }
        PFP_LAUNCH(c, K_EMIT_COUNT, dsize * (30 + sizeof(EBT)), (k_emit_slots<EBT>), nblocks(dsize, BLOCK), ea, cnt, d_hard, (uint32_t *)ea.s_sl, (uint32_t *)ea.s_fb, (uint8_t *)ea.s_fl, (uint8_t *)ea.s_pc, s_g0, gk, gfl, gnu);
        c->arena.release_hi(mkp);
        ea.prec = nullptr; ea.posinfo = nullptr;
    }
    uint4 *sinfo; PFP_ALLOC_HI(c, sinfo, uint4, dsize); ea.sinfo = sinfo;
    const long big_members = c->tun.big_group_members == -2 ? (long)BIG_GROUP_MEMBERS : c->tun.big_group_members;   // < 0: never
    PFP_TRY((device_scan<EBT, 0>(c, cnt, EB, dsize, d_tot)));
    // run-aware: a group of more rows than k_emit_groups_large holds in LDS takes the sort route too (S-32G: 1.4 M rows in groups of 16-64 K
    // rows and ~100 members were ranked row by row in memory by k_emit: 9 ms)
    const uint32_t big_rows = !(runaware && big_members >= 0 && c->tun.emit_group_rows) ? 0u : c->tun.emit_group_rows >= (uint32_t)EG_BUF ? (uint32_t)EG2_BUF : 4u * c->tun.emit_group_rows;      // (tests: small batches -> small limit)
    PFP_LAUNCH(c, K_EMIT_COUNT, dsize * 14, (k_big_mark<EBT>), nblocks(dsize, BLOCK), (const EBT *)cnt, (const uint32_t *)s_g0, (const uint32_t *)gk, (const uint8_t *)gfl, (const uint8_t *)gnu, (const uint32_t *)ea.s_fb, ea.ilist, dsize,
               big_members >= 0 ? (uint32_t)big_members : 0xFFFFFFFFu, runaware ? 1 : 0, (uint8_t *)ea.s_fl, sinfo, cnt2, gqf, gql, d_hard + 1, (const EBT *)EB, (const EBT *)d_tot, big_rows, (big_rows && c->tun.big_group_members == -2) ? 1 : 0);
    EBT tot = 0; unsigned long long hardrows = 0, hh[2] = {0, 0};
    PFP_HIP(c, hipMemcpyAsync(hh, d_hard, 16, hipMemcpyDeviceToHost, c->stream));
    PFP_HIP(c, hipMemcpyAsync(&tot, d_tot, sizeof(EBT), hipMemcpyDeviceToHost, c->stream));
    PFP_HIP(c, hipStreamSynchronize(c->stream));
    hardrows = hh[0]; ea.big_total = hh[1];
    {   // slot under every EMIT_TILE-th output row (k_emit, k_fill, k_sample_values, k_slice_bounds start their searches there)
        const uint64_t ntiles = ((uint64_t)tot + EMIT_TILE - 1) / EMIT_TILE;
        uint32_t *tile_slot; PFP_ALLOC_HI(c, tile_slot, uint32_t, ntiles + 1);
        PFP_LAUNCH(c, K_EMIT_COUNT, dsize * 2 * sizeof(EBT) + ntiles * 4, (k_tile_slots<EBT>), nblocks(dsize, BLOCK), (const EBT *)cnt, (const EBT *)EB, dsize, ntiles, tile_slot);
        ea.tile_slot = tile_slot;
    }
    // enumeration order of k_emit: every row, or (run-aware) the rows of the special slots through a compacted list of them
    ea.ENB = EB; ea.etile_slot = ea.tile_slot; ea.elist = nullptr; ea.cpos = nullptr; ea.ecount = (uint32_t)dsize; ea.special = 0; ea.q0 = 0; ea.qspec = nullptr; ea.cinfo = nullptr; ea.cgb = nullptr; ea.town = nullptr;
    uint64_t tot2 = 0;
    if (runaware) {
        uint32_t *flag, *cpos, *spl, *d_cnt;
        PFP_ALLOC_HI(c, flag, uint32_t, dsize); PFP_ALLOC_HI(c, cpos, uint32_t, dsize + 1); PFP_ALLOC_HI(c, d_cnt, uint32_t, 1);
        PFP_LAUNCH(c, K_EMIT_COUNT, dsize * 5, k_flag_special, nblocks(dsize, BLOCK), (const uint8_t *)ea.s_fl, dsize, flag);
        PFP_TRY((device_scan<uint32_t, 0>(c, flag, cpos, dsize, d_cnt)));
        uint32_t nsp = 0; PFP_TRY(d2h_u32(c, d_cnt, &nsp));
        PFP_ALLOC_HI(c, spl, uint32_t, (size_t)nsp + 1);
        if (nsp) PFP_LAUNCH(c, K_COMPACT, dsize * 12, k_compact_scatter, nblocks(dsize, BLOCK), (const uint32_t *)nullptr, (const uint32_t *)flag, (const uint32_t *)cpos, dsize, spl);
        EBT *cntc, *ENBc;
        PFP_ALLOC_HI(c, cntc, EBT, (size_t)nsp + 1); PFP_ALLOC_HI(c, ENBc, EBT, (size_t)nsp + 1);
        if (nsp) {
            PFP_LAUNCH(c, K_EMIT_COUNT, (uint64_t)nsp * (4 + 2 * sizeof(EBT)), (k_gather_counts<EBT>), nblocks(nsp, BLOCK), (const EBT *)cnt, (const uint32_t *)spl, (uint64_t)nsp, cntc);
            PFP_TRY((device_scan<EBT, 0>(c, cntc, ENBc, nsp, ENBc + nsp)));
            EBT t2 = 0;
            PFP_HIP(c, hipMemcpyAsync(&t2, ENBc + nsp, sizeof(EBT), hipMemcpyDeviceToHost, c->stream));
            PFP_HIP(c, hipStreamSynchronize(c->stream));
            tot2 = (uint64_t)t2;
            const uint64_t ntiles2 = (tot2 + EMIT_TILE - 1) / EMIT_TILE;
            uint32_t *et; PFP_ALLOC_HI(c, et, uint32_t, ntiles2 + 1);
            PFP_LAUNCH(c, K_EMIT_COUNT, (uint64_t)nsp * 2 * sizeof(EBT) + ntiles2 * 4, (k_tile_slots<EBT>), nblocks(nsp, BLOCK), (const EBT *)cntc, (const EBT *)ENBc, (uint64_t)nsp, ntiles2, et);
            ea.etile_slot = et;
        } else PFP_HIP(c, hipMemsetAsync(ENBc, 0, sizeof(EBT), c->stream));
        ea.ENB = ENBc; ea.elist = spl; ea.cpos = cpos; ea.ecount = nsp; ea.special = 1;
        ea.cinfo = nullptr; ea.cgb = nullptr; ea.town = nullptr;
        if (nsp && c->tun.emit_group_rows) {      // per special slot / per enumeration tile: what k_emit_groups would otherwise chase through three arrays per batch
            const uint64_t ntiles2 = (tot2 + EMIT_TILE - 1) / EMIT_TILE;
            uint4 *cinfo; unsigned long long *cgb; uint32_t *town;
            PFP_ALLOC_HI(c, cinfo, uint4, nsp); PFP_ALLOC_HI(c, cgb, unsigned long long, nsp); PFP_ALLOC_HI(c, town, uint32_t, ntiles2 + 1);
            PFP_LAUNCH(c, K_EMIT_COUNT, (uint64_t)nsp * 50, (k_special_pack<EBT>), nblocks(nsp, BLOCK), ea, nsp, cinfo, cgb);
            PFP_LAUNCH(c, K_EMIT_COUNT, (ntiles2 + 1) * 40, (k_tile_own<EBT>), nblocks(ntiles2 + 1, BLOCK), (const EBT *)ENBc, (const uint4 *)cinfo, (const uint32_t *)ea.etile_slot, (uint64_t)nsp, ntiles2, town);
            ea.cinfo = cinfo; ea.cgb = cgb; ea.town = town;
        }
    }
    if (c->tun.verbose) {
        unsigned long long *d_hist, hist[64]; PFP_ALLOC_HI(c, d_hist, unsigned long long, 64);
        PFP_HIP(c, hipMemsetAsync(d_hist, 0, 512, c->stream));
        PFP_LAUNCH(c, K_MISC, dsize * 20, (k_group_stats<EBT>), nblocks(dsize, BLOCK), (const EBT *)cnt, (const EBT *)EB, (const uint32_t *)s_g0, (const uint32_t *)gk, (const uint8_t *)ea.s_fl, dsize, (uint64_t)tot, d_hist);
        PFP_HIP(c, hipMemcpyAsync(hist, d_hist, 512, hipMemcpyDeviceToHost, c->stream));
        PFP_HIP(c, hipStreamSynchronize(c->stream));
        fprintf(stderr, "[pfbwt_hip] rows by group members (rows) x group rows (cols: <1K <4K <16K <64K <256K <1M <4M more); hard %llu, sort-route %llu of %llu; run-aware %d: %llu rows of %u special slots enumerated\n",
                hh[0], hh[1], (unsigned long long)tot, (int)runaware, (unsigned long long)tot2, (unsigned)ea.ecount);
        static const char *kn[8] = {"1", "2-3", "4-7", "8-15", "16-31", "32-63", "64-127", "128+"};
        for (int a = 0; a < 8; ++a) { fprintf(stderr, "[pfbwt_hip]  k %-7s", kn[a]); for (int b = 0; b < 8; ++b) fprintf(stderr, " %13llu", hist[a * 8 + b]); fprintf(stderr, "\n"); }
    }
    const uint64_t nout = tot;
    if (nout < 2) return PFP_E_CORRUPT;
    if (c->n && nout != c->n + 1) return PFP_E_CORRUPT;         // emission must produce exactly n+1 rows
    if (!c->n) c->n = nout - 1;
    ea.nout = nout; ea.n = c->n;
    c->nout = nout; c->hard = hardrows; c->easy = nout - hardrows;
    if (c->flags & PFP_FLAG_U64) return emit_and_sample<uint64_t, EBT>(c, ea, want_sa != 0, want_rssa != 0, slice, nslices, tot2);
    if (nout > 0xFFFFFFFFULL) return PFP_E_TOO_LARGE;           // 32-bit uint_t cannot hold the SA values (pfparser.hpp:326-331)
    return emit_and_sample<uint32_t, EBT>(c, ea, want_sa != 0, want_rssa != 0, slice, nslices, tot2);
}
} // extern "C++"

static int bwt_build_body(pfp_ctx *c, int want_sa, int want_rssa, int slice, int nslices, pfp_bwt_sizes *out);
static int bwt_build_impl(pfp_ctx *c, int want_sa, int want_rssa, int slice, int nslices, pfp_bwt_sizes *out)
{
    if (!c || nslices < 1 || slice < 0 || slice >= nslices) return PFP_E_ARG;
    if (c->stage < 2) return PFP_E_STATE;
    if ((want_sa || want_rssa) && !c->d_bwsai) return PFP_E_STATE;
    PFP_HIP(c, hipSetDevice(c->device));
    c->arena.release_lo(c->lo_after_pbwt);
    if (!c->gsa_valid) {   // gsacak, pfbwt.hpp:211 (--pfbwt-only: the loaded dictionary has not been sorted yet)
        ArenaGuard gs(c);
        const int rs = gs.done(sort_dict_suffixes(c));
        if (rs != PFP_OK) { c->gsa_valid = false; return rs; }
        c->lo_after_pbwt = c->arena.mark_lo();
    }
    ArenaGuard g(c);
    const int rc = g.done(bwt_build_body(c, want_sa, want_rssa, slice, nslices, out));
    if (rc != PFP_OK) { c->stage = 2; c->d_bwt = nullptr; c->d_sa = c->d_ssa = c->d_esa = nullptr; }   // e.g. PFP_E_NOMEM with want_sa: retry without, or in slices
    return rc;
}
static int bwt_build_body(pfp_ctx *c, int want_sa, int want_rssa, int slice, int nslices, pfp_bwt_sizes *out)
{
    HostTimer timer;
    const size_t mk = c->arena.mark_hi();
    c->emit_scratch_mark = mk;
    const uint64_t dsize = c->dsize, dwords = c->dwords;
    uint32_t *F, *s_sl, *s_fb; uint8_t *s_fl, *s_pc; uint4 *winfo;
    if (dwords > WID_MASK) return PFP_E_TOO_LARGE;
    PFP_ALLOC_HI(c, F, uint32_t, dwords + 1);
    PFP_ALLOC_HI(c, s_sl, uint32_t, dsize); PFP_ALLOC_HI(c, s_fb, uint32_t, dsize);
    PFP_ALLOC_HI(c, s_fl, uint8_t, dsize); PFP_ALLOC_HI(c, s_pc, uint8_t, dsize);
    // F[r] = 1 + sum_{r' < r} occ[r']  (ilist[0] is the EOS row; pfbwt.hpp:259-268)
    PFP_TRY((device_scan<uint32_t, 0>(c, c->d_occ, F, dwords, nullptr)));
    PFP_LAUNCH(c, K_MISC, dwords * 8, k_u32_add_store, nblocks(dwords, BLOCK), (const uint32_t *)F, dwords, 1u, F);
    EmitArgs ea;
    ea.D = c->d_dict; ea.dsize = dsize; ea.dwords = (uint32_t)dwords; ea.w = c->w;
    ea.SA = c->d_gsa; ea.srank = c->d_srank; ea.ws = c->d_ws; ea.wrank = c->d_wrank;
    ea.occ = c->d_occ; ea.F = F; ea.ilist = c->d_ilist; ea.bwsai = c->d_bwsai; ea.bwlast = c->d_bwlast; ea.bwl_il = c->d_bwl_il;
    PFP_ALLOC_HI(c, winfo, uint4, dwords);
    PFP_LAUNCH(c, K_MISC, dwords * 32, k_pack_winfo, nblocks(dwords, BLOCK), (const uint32_t *)c->d_ws, (const uint32_t *)c->d_wrank, (const uint32_t *)c->d_occ, (const uint32_t *)F, dwords, winfo);
    ea.winfo = winfo;
    // per dictionary offset: one 16-byte record for k_emit_slots (suffix lengths must fit 26 bits), else word id | preceding byte + class head
    uint32_t maxlen = 0;
    {
        uint32_t *d_ml; PFP_ALLOC_HI(c, d_ml, uint32_t, 1);
        PFP_HIP(c, hipMemsetAsync(d_ml, 0, 4, c->stream));
        PFP_LAUNCH(c, K_MISC, dwords * 4, k_max_word_length, nblocks(dwords, BLOCK), (const uint32_t *)c->d_ws, dwords, d_ml);
        PFP_TRY(d2h_u32(c, d_ml, &maxlen));
    }
    ea.bwsai_il = nullptr; ea.use_e0 = 0;
    if (want_sa && c->d_bwsai && c->nrows) {      // bwsai in ilist order: one gather per parse row here instead of a second dependent gather per OUTPUT row
        tpos_t *E; PFP_ALLOC_HI(c, E, tpos_t, c->nrows);
        PFP_LAUNCH(c, K_MISC, c->nrows * 20, k_bwsai_by_ilist, nblocks(c->nrows, BLOCK), (const uint32_t *)c->d_ilist, (const tpos_t *)c->d_bwsai, c->nrows, E);
        ea.bwsai_il = E;
        ea.use_e0 = (c->n != 0 && c->n + (uint64_t)c->w + 2 < 0xFFFFFFFFULL && !c->tun.no_slot_records) ? 1 : 0;      // positions fit the 32 bits of prec.x / s_g0
    }
    ea.posinfo = nullptr; ea.prec = nullptr; ea.wordid = c->d_wordid; ea.use_prec = (maxlen < (1u << PREC_SL_BITS) && !c->tun.no_slot_records) ? 1 : 0;
    ea.EB = nullptr; ea.s_sl = s_sl; ea.s_fb = s_fb; ea.s_fl = s_fl; ea.s_pc = s_pc; ea.nout = 0; ea.n = 0; ea.e0 = ea.e1 = ea.w0 = ea.w1 = 0;
    c->have_sa = want_sa != 0; c->have_rssa = want_rssa != 0;
    // 64-bit row counters when the text may have 2^32 - 1 positions or more (n unknown after pfp_bwt_load without a hint)
    const bool force_wide = c->tun.force_wide_rows != 0;
    const bool wide = force_wide || c->n == 0 || c->n + 2 >= 0xFFFFFFFFULL;
    int rc = wide ? emit_stage<uint64_t>(c, ea, want_sa, want_rssa, slice, nslices) : emit_stage<uint32_t>(c, ea, want_sa, want_rssa, slice, nslices);
    if (rc != PFP_OK) return rc;
    PFP_HIP(c, hipStreamSynchronize(c->stream));
    c->arena.release_hi(mk);
    c->stage = 3;
    c->stage_ms[2] = timer.ms();
    if (out) { out->nout = c->nout; out->r = c->runs; out->easy_cases = c->easy; out->hard_cases = c->hard; }
    return PFP_OK;
}

int pfp_bwt_build(pfp_ctx *c, int want_sa, int want_rssa, pfp_bwt_sizes *out) { return bwt_build_impl(c, want_sa, want_rssa, 0, 1, out); }
int pfp_bwt_build_stream(pfp_ctx *c, int want_sa, int want_rssa, uint8_t *host_bwt, void *host_sa, pfp_bwt_sizes *out)
{
    if (!c || !host_bwt || (want_sa && !host_sa)) return PFP_E_ARG;
    PFP_HIP(c, hipSetDevice(c->device));
    PFP_TRY(ensure_copy_stream(c));
    c->h_bwt = host_bwt; c->h_sa = want_sa ? host_sa : nullptr;
    const int rc = bwt_build_impl(c, want_sa, want_rssa, 0, 1, out);
    c->h_bwt = nullptr; c->h_sa = nullptr;
    const hipError_t e = hipStreamSynchronize(c->fa.copy);               // the last windows' rows have arrived
    if (rc == PFP_OK && e != hipSuccess) { c->hip_err = (int)e; return PFP_E_HIP; }
    return rc;
}
int pfp_text_length(pfp_ctx *c, uint64_t *n) { if (!c || !n) return PFP_E_ARG; *n = c->n; return PFP_OK; }
int pfp_bwt_build_slice(pfp_ctx *c, int want_sa, int want_rssa, int slice, int nslices, pfp_bwt_sizes *out, uint64_t *slice_begin, uint64_t *slice_rows, uint64_t *esa_pairs)
{
    int rc = bwt_build_impl(c, want_sa, want_rssa, slice, nslices, out);
    if (rc != PFP_OK) return rc;
    if (slice_begin) *slice_begin = c->slice_begin;
    if (slice_rows) *slice_rows = c->slice_rows;
    if (esa_pairs) *esa_pairs = c->esa_pairs;
    return PFP_OK;
}

int pfp_bwt_get(pfp_ctx *c, uint8_t *bwt, void *sa, void *ssa, void *esa)
{
    if (!c) return PFP_E_ARG;
    if (c->stage < 3) return PFP_E_STATE;
    PFP_HIP(c, hipSetDevice(c->device));
    const size_t U = (c->flags & PFP_FLAG_U64) ? 8 : 4;
    if (bwt) PFP_HIP(c, hipMemcpy(bwt, c->d_bwt, c->slice_rows, hipMemcpyDeviceToHost));
    if (sa) { if (!c->d_sa) return PFP_E_STATE; PFP_HIP(c, hipMemcpy(sa, c->d_sa, c->slice_rows * U, hipMemcpyDeviceToHost)); }
    if (ssa) { if (!c->d_ssa) return PFP_E_STATE; PFP_HIP(c, hipMemcpy(ssa, c->d_ssa, c->runs * 2 * U, hipMemcpyDeviceToHost)); }
    if (esa) { if (!c->d_esa) return PFP_E_STATE; PFP_HIP(c, hipMemcpy(esa, c->d_esa, c->esa_pairs * 2 * U, hipMemcpyDeviceToHost)); }
    return PFP_OK;
}

int pfp_bwt_device_ptrs(pfp_ctx *c, const void **d_bwt, const void **d_sa, const void **d_ssa, const void **d_esa)
{
    if (!c) return PFP_E_ARG;
    if (c->stage < 3) return PFP_E_STATE;
    if (d_bwt) *d_bwt = c->d_bwt;
    if (d_sa) *d_sa = c->d_sa;
    if (d_ssa) *d_ssa = c->d_ssa;
    if (d_esa) *d_esa = c->d_esa;
    return PFP_OK;
}

// ---- marker-array post-pass (include/marker_array.hpp:138-174, src/mps_to_ma.cpp) ----------------------
extern "C++" {
template <typename SAT> static int marker_array_impl(pfp_ctx *c, const uint64_t *mps, uint64_t mps_words, const SAT *d_sa, uint64_t nrows, uint64_t *out_words)
{
    // host: the records of the .mps stream; every distinct marker list gets one id (the reference compares lists by content)
    std::vector<uint64_t> istart, iend, lvals; std::vector<uint32_t> ilist, loff(1, 0u);
    std::map<std::vector<uint64_t>, uint32_t> ids;
    for (uint64_t i = 0; i < mps_words;) {
        uint64_t j = i;
        while (j < mps_words && mps[j] != ~0ULL) ++j;
        if (j == mps_words || j - i < 2) return PFP_E_CORRUPT;                 // a record without its keys or its delimiter
        if (!istart.empty() && (mps[i] <= iend.back() || mps[i + 1] < mps[i])) return PFP_E_CORRUPT;   // intervals ascend and do not overlap (rle_window_array.hpp:31-34)
        std::vector<uint64_t> lst(mps + i + 2, mps + j);
        auto it = ids.find(lst);
        uint32_t id;
        if (it != ids.end()) id = it->second;
        else { id = (uint32_t)ids.size(); ids.emplace(lst, id); lvals.insert(lvals.end(), lst.begin(), lst.end()); loff.push_back((uint32_t)lvals.size()); }
        // a record with an empty list answers at() like no record at all
        if (!lst.empty()) { istart.push_back(mps[i]); iend.push_back(mps[i + 1]); ilist.push_back(id); }
        i = j + 1;
    }
    if (istart.size() >= 0xFFFFFFF0ULL || lvals.size() >= 0xFFFFFFF0ULL || nrows >= 0xFFFFFFF0ULL) return PFP_E_TOO_LARGE;      // run heads are counted and placed with 32-bit values
    const uint32_t nint = (uint32_t)istart.size();
    const size_t mk = c->arena.mark_hi();
    uint64_t *d_is, *d_ie, *d_lv; uint32_t *d_il, *d_lo, *rowlist, *head, *pos, *d_cnt;
    PFP_ALLOC_HI(c, d_is, uint64_t, nint); PFP_ALLOC_HI(c, d_ie, uint64_t, nint); PFP_ALLOC_HI(c, d_il, uint32_t, nint);
    PFP_ALLOC_HI(c, d_lo, uint32_t, loff.size()); PFP_ALLOC_HI(c, d_lv, uint64_t, lvals.size());
    PFP_ALLOC_HI(c, rowlist, uint32_t, nrows); PFP_ALLOC_HI(c, head, uint32_t, nrows); PFP_ALLOC_HI(c, pos, uint32_t, nrows); PFP_ALLOC_HI(c, d_cnt, uint32_t, 1);
    if (nint) {
        PFP_HIP(c, hipMemcpyAsync(d_is, istart.data(), (size_t)nint * 8, hipMemcpyHostToDevice, c->stream));
        PFP_HIP(c, hipMemcpyAsync(d_ie, iend.data(), (size_t)nint * 8, hipMemcpyHostToDevice, c->stream));
        PFP_HIP(c, hipMemcpyAsync(d_il, ilist.data(), (size_t)nint * 4, hipMemcpyHostToDevice, c->stream));
    }
    PFP_HIP(c, hipMemcpyAsync(d_lo, loff.data(), loff.size() * 4, hipMemcpyHostToDevice, c->stream));
    if (!lvals.empty()) PFP_HIP(c, hipMemcpyAsync(d_lv, lvals.data(), lvals.size() * 8, hipMemcpyHostToDevice, c->stream));
    const unsigned gr = nblocks(nrows, BLOCK);
    PFP_LAUNCH(c, K_MISC, nrows * (sizeof(SAT) + 4 + 40), (k_ma_lookup<SAT>), gr, d_sa, nrows, (const uint64_t *)d_is, (const uint64_t *)d_ie, (const uint32_t *)d_il, nint, rowlist);
    PFP_LAUNCH(c, K_MISC, nrows * 8, k_ma_heads, gr, (const uint32_t *)rowlist, nrows, head);
    PFP_TRY((device_scan<uint32_t, 0>(c, head, pos, nrows, d_cnt)));
    uint32_t nh = 0; PFP_TRY(d2h_u32(c, d_cnt, &nh));          // also waits for the host vectors' uploads
    uint64_t *hrow; uint32_t *hlist; unsigned long long *len, *off, *d_tot;
    PFP_ALLOC_HI(c, hrow, uint64_t, nh); PFP_ALLOC_HI(c, hlist, uint32_t, nh); PFP_ALLOC_HI(c, len, unsigned long long, nh); PFP_ALLOC_HI(c, off, unsigned long long, nh); PFP_ALLOC_HI(c, d_tot, unsigned long long, 1);
    PFP_LAUNCH(c, K_MISC, nrows * 12, k_ma_collect, gr, (const uint32_t *)rowlist, (const uint32_t *)head, (const uint32_t *)pos, nrows, hrow, hlist);
    PFP_LAUNCH(c, K_MISC, (uint64_t)nh * 16, k_ma_lengths, nblocks(nh, BLOCK), (const uint32_t *)hlist, (const uint32_t *)d_lo, (uint64_t)nh, len);
    PFP_TRY((device_scan<unsigned long long, 0>(c, len, off, nh, d_tot)));
    unsigned long long tot = 0;
    PFP_HIP(c, hipMemcpyAsync(&tot, d_tot, 8, hipMemcpyDeviceToHost, c->stream));
    PFP_HIP(c, hipStreamSynchronize(c->stream));
    // a result of an earlier call on the same build (several .mps streams against one suffix array) gives its space back first
    if (c->ma_lo_mark != (size_t)-1 && c->arena.lo == c->ma_lo_end) c->arena.release_lo(c->ma_lo_mark);
    c->d_ma = nullptr; c->ma_words = tot; c->ma_lo_mark = (size_t)-1;
    if (tot) {
        const size_t mark = c->arena.mark_lo();
        uint64_t *d_out = (uint64_t *)c->arena.alloc_lo(tot * 8);               // result: low end, survives the release of the scratch
        if (!d_out) return PFP_E_NOMEM;
        c->ma_lo_mark = mark; c->ma_lo_end = c->arena.mark_lo();
        PFP_LAUNCH(c, K_MISC, tot * 8, k_ma_write, nblocks(nh, BLOCK), (const uint64_t *)hrow, (const uint32_t *)hlist, (const unsigned long long *)off, (const uint32_t *)d_lo, (const uint64_t *)d_lv, (uint64_t)nh, nrows, d_out);
        PFP_HIP(c, hipStreamSynchronize(c->stream));
        c->d_ma = d_out;
    }
    c->arena.release_hi(mk);
    if (out_words) *out_words = tot;
    return PFP_OK;
}
} // extern "C++"

int pfp_marker_array(pfp_ctx *c, const uint64_t *mps, uint64_t mps_words, const void *sa_host, uint64_t nrows, uint64_t *out_words)
{
    if (!c || (!mps && mps_words)) return PFP_E_ARG;
    PFP_HIP(c, hipSetDevice(c->device));
    const bool u64 = (c->flags & PFP_FLAG_U64) != 0;
    ArenaGuard g(c);
    if (!sa_host) {      // fused: the suffix array the last pfp_bwt_build(want_sa = 1) left on the device (whole output, not a slice)
        if (c->stage < 3 || !c->d_sa || c->slice_rows != c->nout) return PFP_E_STATE;
        return g.done(u64 ? marker_array_impl<uint64_t>(c, mps, mps_words, (const uint64_t *)c->d_sa, c->nout, out_words)
                          : marker_array_impl<uint32_t>(c, mps, mps_words, (const uint32_t *)c->d_sa, c->nout, out_words));
    }
    if (!nrows) return PFP_E_ARG;      // stand-alone (src/mps_to_ma.cpp): the suffix array comes from a file or pipe
    reset_results(c);
    const size_t U = u64 ? 8 : 4;
    int rc = ensure_arena(c, nrows);
    if (rc != PFP_OK) return rc;
    c->arena.reset();
    ArenaGuard g2(c);
    auto body = [&]() -> int {
        void *d_sa = c->arena.alloc_hi(nrows * U);
        if (!d_sa) return PFP_E_NOMEM;
        PFP_TRY(h2d_copy(c, (uint8_t *)d_sa, (const uint8_t *)sa_host, nrows * U));
        return u64 ? marker_array_impl<uint64_t>(c, mps, mps_words, (const uint64_t *)d_sa, nrows, out_words) : marker_array_impl<uint32_t>(c, mps, mps_words, (const uint32_t *)d_sa, nrows, out_words);
    };
    rc = g2.done(body());
    if (rc != PFP_OK) { c->d_ma = nullptr; c->ma_words = 0; c->ma_lo_mark = (size_t)-1; }
    return rc;
}
int pfp_marker_array_get(pfp_ctx *c, uint64_t *dst)
{
    if (!c || (!dst && c->ma_words)) return PFP_E_ARG;
    if (c->ma_words && !c->d_ma) return PFP_E_STATE;
    PFP_HIP(c, hipSetDevice(c->device));
    if (c->ma_words) PFP_HIP(c, hipMemcpy(dst, c->d_ma, c->ma_words * 8, hipMemcpyDeviceToHost));
    return PFP_OK;
}

// ---- development aid: position-weighted checksum of a device buffer (sum over bytes of (byte + 1) * mix(global position),
// two independent mixes, modulo 2^64): the checksums of the pieces of a buffer add up to the checksum of the whole, so
// outputs that live sliced over several builds / GPUs can be compared with a single-context output without moving them
__global__ __launch_bounds__(BLOCK) void k_checksum(const uint8_t *p, uint64_t bytes, uint64_t offset, unsigned long long *out)
{
    __shared__ unsigned long long red[4];
    const uint64_t i0 = ((uint64_t)blockIdx.x * BLOCK + threadIdx.x) * 16;
    unsigned long long a = 0, b = 0;
    for (int k = 0; k < 16 && i0 + k < bytes; ++k) {
        const uint64_t pos = offset + i0 + k, v = (uint64_t)p[i0 + k] + 1;
        a += v * mix64(pos * 0x9E3779B97F4A7C15ULL + 1); b += v * mix64(pos * 0xD6E8FEB86659FD93ULL + 7);
    }
    unsigned long long ta, tb;
    (void)block_excl_sum(a, red, &ta); (void)block_excl_sum(b, red, &tb);
    if (threadIdx.x == 0) { atomicAdd(&out[0], ta); atomicAdd(&out[1], tb); }
}
int pfp_debug_checksum(pfp_ctx *c, const void *d_buf, uint64_t bytes, uint64_t global_offset, uint64_t out[2])
{
    if (!c || (!d_buf && bytes) || !out) return PFP_E_ARG;
    PFP_HIP(c, hipSetDevice(c->device));
    unsigned long long *d_out; PFP_HIP(c, hipMalloc((void **)&d_out, 16));
    PFP_HIP(c, hipMemsetAsync(d_out, 0, 16, c->stream));
    const uint64_t per = (uint64_t)1 << 34;            // pieces of 16 GiB keep the grid below 2^31 workgroups
    for (uint64_t off = 0; off < bytes; off += per) {
        const uint64_t nb = bytes - off < per ? bytes - off : per;
        PFP_LAUNCH(c, K_MISC, nb, k_checksum, nblocks(nb, 16 * BLOCK), (const uint8_t *)d_buf + off, nb, global_offset + off, d_out);
    }
    unsigned long long h[2];
    PFP_HIP(c, hipMemcpyAsync(h, d_out, 16, hipMemcpyDeviceToHost, c->stream));
    PFP_HIP(c, hipStreamSynchronize(c->stream));
    PFP_HIP(c, hipFree(d_out));
    out[0] = h[0]; out[1] = h[1];
    return PFP_OK;
}

// sum of the 64-bit words of a device buffer and sum of word * (index + 1), modulo 2^64 (bench.py compares the outputs that
// reached host memory with the device-resident ones this way; bytes behind the last whole word count as a zero-padded word)
__global__ __launch_bounds__(BLOCK) void k_wordsum(const uint8_t *p, uint64_t bytes, unsigned long long *out)
{
    __shared__ unsigned long long red[4];
    const uint64_t nw = (bytes + 7) / 8;
    unsigned long long a = 0, b = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < nw; i += (uint64_t)gridDim.x * BLOCK) {
        unsigned long long v = 0;
        if (8 * i + 8 <= bytes) v = ld8(p + 8 * i); else for (uint64_t k = 0; 8 * i + k < bytes; ++k) v |= (unsigned long long)p[8 * i + k] << (8 * k);
        a += v; b += v * (i + 1);
    }
    unsigned long long ta, tb;
    (void)block_excl_sum(a, red, &ta); (void)block_excl_sum(b, red, &tb);
    if (threadIdx.x == 0) { atomicAdd(&out[0], ta); atomicAdd(&out[1], tb); }
}
int pfp_debug_wordsum(pfp_ctx *c, const void *d_buf, uint64_t bytes, uint64_t out[2])
{
    if (!c || (!d_buf && bytes) || !out) return PFP_E_ARG;
    PFP_HIP(c, hipSetDevice(c->device));
    unsigned long long *d_out; PFP_HIP(c, hipMalloc((void **)&d_out, 16));
    PFP_HIP(c, hipMemsetAsync(d_out, 0, 16, c->stream));
    if (bytes) PFP_LAUNCH(c, K_MISC, bytes, k_wordsum, 4096, (const uint8_t *)d_buf, bytes, d_out);
    unsigned long long h[2];
    PFP_HIP(c, hipMemcpyAsync(h, d_out, 16, hipMemcpyDeviceToHost, c->stream));
    PFP_HIP(c, hipStreamSynchronize(c->stream));
    PFP_HIP(c, hipFree(d_out));
    out[0] = h[0]; out[1] = h[1];
    return PFP_OK;
}
// Full-size order check that does not lean on the engine's own sorting: .esa[k] and .ssa[k + 1] are ADJACENT
// rows of the BWT matrix (a run ends, the next one starts), so the suffix at esa[k].sa must be lexicographically smaller than the
// suffix at ssa[k + 1].sa -- compared directly on the text that is still resident (its w Dollars stand for the terminator, which is
// smaller than every base).  One thread per pair, eight bytes per step.  out: pairs checked, order violations, pairs whose rows
// are not adjacent, longest common prefix seen, sum of the common prefixes.
extern "C++" {
template <typename SAT> __global__ __launch_bounds__(BLOCK) void k_check_sample_order(const uint8_t *X, uint64_t n, const SAT *ssa, const SAT *esa, uint64_t r, unsigned long long *out)
{
    const uint64_t k = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (k + 1 >= r) return;
    const uint64_t ra = esa[2 * k], a = esa[2 * k + 1], rb = ssa[2 * k + 2], b = ssa[2 * k + 3];
    if (ra + 1 != rb) atomicAdd(&out[2], 1ULL);
    uint64_t h = 0; bool ok = false;
    for (;;) {
        const uint64_t pa = a + h, pb = b + h;
        if (pa > n || pb > n) break;                                        // (cannot happen: the Dollar at n ends every comparison)
        const uint64_t va = ld8(X + pa), vb = ld8(X + pb);                  // the buffer holds w Dollars and slack behind the text
        if (va != vb) {
            const int sh = (__ffsll((long long)(va ^ vb)) - 1) & ~7;
            const uint64_t q = (uint64_t)sh >> 3;
            if (pa + q > n || pb + q > n) break;                            // the difference lies behind the terminator of one of them: not an order
            ok = ((va >> sh) & 0xFF) < ((vb >> sh) & 0xFF); h += q;
            break;
        }
        h += 8;
    }
    if (!ok) atomicAdd(&out[1], 1ULL);
    atomicAdd(&out[0], 1ULL); atomicMax(&out[3], (unsigned long long)h); atomicAdd(&out[4], (unsigned long long)h);
}
}
int pfp_debug_check_sample_order(pfp_ctx *c, uint64_t out[5])
{
    if (!c || !out) return PFP_E_ARG;
    if (c->stage < 3 || !c->d_ssa || !c->d_esa || !c->tb || c->tb_n != c->n || c->slice_rows != c->nout) return PFP_E_STATE;      // needs the text and the whole output's samples
    PFP_HIP(c, hipSetDevice(c->device));
    unsigned long long *d_out; PFP_HIP(c, hipMalloc((void **)&d_out, 40));
    PFP_HIP(c, hipMemsetAsync(d_out, 0, 40, c->stream));
    const uint64_t r = c->runs;
    if (r > 1) {
        if (c->flags & PFP_FLAG_U64) PFP_LAUNCH(c, K_MISC, r * 64, (k_check_sample_order<uint64_t>), nblocks(r, BLOCK), (const uint8_t *)c->tb + 16, c->n, (const uint64_t *)c->d_ssa, (const uint64_t *)c->d_esa, r, d_out);
        else PFP_LAUNCH(c, K_MISC, r * 64, (k_check_sample_order<uint32_t>), nblocks(r, BLOCK), (const uint8_t *)c->tb + 16, c->n, (const uint32_t *)c->d_ssa, (const uint32_t *)c->d_esa, r, d_out);
    }
    unsigned long long h[5];
    PFP_HIP(c, hipMemcpyAsync(h, d_out, 40, hipMemcpyDeviceToHost, c->stream));
    PFP_HIP(c, hipStreamSynchronize(c->stream));
    PFP_HIP(c, hipFree(d_out));
    for (int i = 0; i < 5; ++i) out[i] = h[i];
    return PFP_OK;
}
// Properties of a full suffix array at any size, on the device (the host-side version, tools/big_check.py, needs minutes and
// 10 bytes of host memory per base): every value of [0, n] occurs exactly once (bitmap + atomicOr), row 0 holds n, BWT[row] is the
// text byte in front of SA[row] (0x00 for the one row whose suffix is the whole text).  out: rows checked, values out of range,
// values seen twice, rows with a wrong BWT byte, 0x00 bytes in the BWT.
extern "C++" {
template <typename SAT> __global__ __launch_bounds__(BLOCK) void k_check_sa(const uint8_t *X, uint64_t n, const SAT *sa, const uint8_t *bwt, uint64_t rows, uint32_t *seen, unsigned long long *out)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= rows) return;
    const uint64_t v = sa[i];
    if (v > n) { atomicAdd(&out[1], 1ULL); return; }
    const uint32_t bit = 1u << (v & 31);
    if (atomicOr(&seen[v >> 5], bit) & bit) atomicAdd(&out[2], 1ULL);
    const uint8_t want = v == 0 ? (uint8_t)0 : X[v - 1];
    if (bwt[i] != want || (i == 0 && v != n)) atomicAdd(&out[3], 1ULL);
    if (bwt[i] == 0) atomicAdd(&out[4], 1ULL);
}
}
int pfp_debug_check_sa(pfp_ctx *c, uint64_t out[5])
{
    if (!c || !out) return PFP_E_ARG;
    if (c->stage < 3 || !c->d_sa || !c->have_sa || !c->d_bwt || !c->tb || c->tb_n != c->n || c->slice_rows != c->nout) return PFP_E_STATE;      // needs the text and the whole SA
    PFP_HIP(c, hipSetDevice(c->device));
    const uint64_t rows = c->nout, words = (c->n + 32) / 32 + 1;
    unsigned long long *d_out; uint32_t *seen;
    PFP_HIP(c, hipMalloc((void **)&d_out, 40)); PFP_HIP(c, hipMalloc((void **)&seen, words * 4));
    PFP_HIP(c, hipMemsetAsync(d_out, 0, 40, c->stream)); PFP_HIP(c, hipMemsetAsync(seen, 0, words * 4, c->stream));
    if (c->flags & PFP_FLAG_U64) PFP_LAUNCH(c, K_MISC, rows * 10, (k_check_sa<uint64_t>), nblocks(rows, BLOCK), (const uint8_t *)c->tb + 16, c->n, (const uint64_t *)c->d_sa, (const uint8_t *)c->d_bwt, rows, seen, d_out);
    else PFP_LAUNCH(c, K_MISC, rows * 6, (k_check_sa<uint32_t>), nblocks(rows, BLOCK), (const uint8_t *)c->tb + 16, c->n, (const uint32_t *)c->d_sa, (const uint8_t *)c->d_bwt, rows, seen, d_out);
    unsigned long long h[5];
    PFP_HIP(c, hipMemcpyAsync(h, d_out, 40, hipMemcpyDeviceToHost, c->stream));
    PFP_HIP(c, hipStreamSynchronize(c->stream));
    PFP_HIP(c, hipFree(d_out)); PFP_HIP(c, hipFree(seen));
    out[0] = rows; for (int i = 1; i < 5; ++i) out[i] = h[i];
    return PFP_OK;
}
// The run samples of a -s -r build against its own full outputs, on the device (S-3G: 2.3 G runs = 74 GB of samples, not something
// a test moves to the host): run k starts at row ssa[k].row, a position where the BWT byte changes (or row 0), and ends at
// esa[k].row = ssa[k + 1].row - 1 (the last one at the last row); the values are the SA entries of those rows (src/pfbwt-f.cpp:304-315,
// 325-328).  r itself is the engine's count of byte changes, so r distinct change positions are all of them.
// out: runs checked, runs with a wrong row, runs with a wrong value.
extern "C++" {
template <typename SAT> __global__ __launch_bounds__(BLOCK) void k_check_samples(const SAT *ssa, const SAT *esa, uint64_t r, const SAT *sa, const uint8_t *bwt, uint64_t rows, unsigned long long *out)
{
    const uint64_t k = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (k >= r) return;
    const uint64_t rs = ssa[2 * k], vs = ssa[2 * k + 1], re = esa[2 * k], ve = esa[2 * k + 1];
    bool row_ok = rs < rows && re < rows && rs <= re;
    if (row_ok) {
        row_ok = (k == 0 ? rs == 0 : bwt[rs] != bwt[rs - 1]) && (k + 1 == r ? re + 1 == rows : (uint64_t)ssa[2 * k + 2] == re + 1) && bwt[rs] == bwt[re];
    }
    if (!row_ok) { atomicAdd(&out[1], 1ULL); return; }
    if ((uint64_t)sa[rs] != vs || (uint64_t)sa[re] != ve) atomicAdd(&out[2], 1ULL);
}
}
int pfp_debug_check_samples(pfp_ctx *c, uint64_t out[3])
{
    if (!c || !out) return PFP_E_ARG;
    if (c->stage < 3 || !c->d_sa || !c->have_sa || !c->have_rssa || !c->d_ssa || !c->d_esa || c->slice_rows != c->nout || c->esa_pairs != c->runs) return PFP_E_STATE;
    PFP_HIP(c, hipSetDevice(c->device));
    unsigned long long *d_out; PFP_HIP(c, hipMalloc((void **)&d_out, 24));
    PFP_HIP(c, hipMemsetAsync(d_out, 0, 24, c->stream));
    const uint64_t r = c->runs;
    if (r) {
        if (c->flags & PFP_FLAG_U64) PFP_LAUNCH(c, K_MISC, r * 50, (k_check_samples<uint64_t>), nblocks(r, BLOCK), (const uint64_t *)c->d_ssa, (const uint64_t *)c->d_esa, r, (const uint64_t *)c->d_sa, (const uint8_t *)c->d_bwt, c->nout, d_out);
        else PFP_LAUNCH(c, K_MISC, r * 30, (k_check_samples<uint32_t>), nblocks(r, BLOCK), (const uint32_t *)c->d_ssa, (const uint32_t *)c->d_esa, r, (const uint32_t *)c->d_sa, (const uint8_t *)c->d_bwt, c->nout, d_out);
    }
    unsigned long long h[3];
    PFP_HIP(c, hipMemcpyAsync(h, d_out, 24, hipMemcpyDeviceToHost, c->stream));
    PFP_HIP(c, hipStreamSynchronize(c->stream));
    PFP_HIP(c, hipFree(d_out));
    out[0] = r; out[1] = h[1]; out[2] = h[2];
    return PFP_OK;
}
// page-locked host memory for the callers of pfp_bwt_build_stream / pfp_parse_feed_fasta (they need not link the HIP runtime)
int pfp_host_register(void *p, uint64_t bytes) { if (!p || !bytes) return PFP_E_ARG; if (hipHostRegister(p, (size_t)bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return PFP_E_HIP; } return PFP_OK; }
int pfp_host_unregister(void *p) { if (!p) return PFP_E_ARG; if (hipHostUnregister(p) != hipSuccess) { (void)hipGetLastError(); return PFP_E_HIP; } return PFP_OK; }

// ---- development aid: time the pair sort on pseudo-random keys (no product path calls this) ----------
__global__ __launch_bounds__(BLOCK) void k_debug_fill(uint64_t *keys, uint32_t *vals, uint64_t n, int bits, uint64_t seed)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    uint64_t z = (i + seed) * 0x9E3779B97F4A7C15ULL; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; z ^= z >> 31;
    keys[i] = bits >= 64 ? z : (z & ((1ULL << bits) - 1ULL)); vals[i] = (uint32_t)i;
}
__global__ __launch_bounds__(BLOCK) void k_debug_check_sorted(const uint64_t *keys, uint64_t n, uint32_t *bad)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i + 1 < n && keys[i] > keys[i + 1]) atomicAdd(bad, 1u);
}
int pfp_debug_sort(pfp_ctx *c, uint64_t n, int bits, int reps, double *ms_out, uint32_t *unsorted_pairs)
{
    if (!c || n < 2 || bits < 1 || bits > 64) return PFP_E_ARG;
    PFP_HIP(c, hipSetDevice(c->device));
    PFP_TRY(ensure_arena(c, n));
    c->arena.reset(); c->stage = 0;
    uint64_t *k0, *k1; uint32_t *v0, *v1, *d_bad;
    PFP_ALLOC_HI(c, k0, uint64_t, n); PFP_ALLOC_HI(c, k1, uint64_t, n); PFP_ALLOC_HI(c, v0, uint32_t, n); PFP_ALLOC_HI(c, v1, uint32_t, n); PFP_ALLOC_HI(c, d_bad, uint32_t, 1);
    BitRange br = {0, bits};
    double best = 1e30;
    uint64_t *sk = k0; uint32_t *sv = v0;
    for (int r = 0; r < reps; ++r) {
        PFP_LAUNCH(c, K_MISC, n * 12, k_debug_fill, nblocks(n, BLOCK), k0, v0, n, bits, (uint64_t)r * 7919);
        PFP_HIP(c, hipStreamSynchronize(c->stream));
        HostTimer t;
        int rc = radix_sort_pairs<uint64_t>(c, k0, v0, k1, v1, n, &br, 1, &sk, &sv);
        if (rc != PFP_OK) return rc;
        PFP_HIP(c, hipStreamSynchronize(c->stream));
        const double ms = t.ms();
        if (ms < best) best = ms;
    }
    PFP_HIP(c, hipMemsetAsync(d_bad, 0, 4, c->stream));
    PFP_LAUNCH(c, K_MISC, n * 8, k_debug_check_sorted, nblocks(n, BLOCK), (const uint64_t *)sk, n, d_bad);
    uint32_t bad = 0; PFP_TRY(d2h_u32(c, d_bad, &bad));
    if (ms_out) *ms_out = best;
    if (unsorted_pairs) *unsorted_pairs = bad;
    c->arena.reset();
    return PFP_OK;
}

// ---- gsa/gsacak.h:76-103 drop-ins ------------------------------------------------------------------
static int sacak_int_impl(const uint32_t *s, void *SA, uint64_t n, uint64_t k, bool u64)
{
    if (!s || !SA || n == 0) return -1;
    if (n + 64 >= 0xFFFFFFFFULL) return -1;
    int st = 0;
    pfp_ctx *c = pfp_create(10, 100, 0, 0, (uint64_t)(80 * n + (32ULL << 20)), &st);
    if (!c) return -1;
    int rounds = -1;
    do {
        if (ensure_arena(c, n) != PFP_OK) break;
        uint32_t *dS = (uint32_t *)c->arena.alloc_lo(n * 4), *dSA = (uint32_t *)c->arena.alloc_lo(n * 4), *dR = (uint32_t *)c->arena.alloc_lo(n * 4);
        if (!dS || !dSA || !dR) break;
        if (hipMemcpy(dS, s, n * 4, hipMemcpyHostToDevice) != hipSuccess) break;
        int r = 0;
        if (sort_int_suffixes(c, dS, n, k ? k - 1 : 0, dSA, dR, &r) != PFP_OK) break;
        if (hipStreamSynchronize(c->stream) != hipSuccess) break;
        if (!u64) { if (hipMemcpy(SA, dSA, n * 4, hipMemcpyDeviceToHost) != hipSuccess) break; }
        else {
            std::vector<uint32_t> tmp((size_t)n);
            if (hipMemcpy(tmp.data(), dSA, n * 4, hipMemcpyDeviceToHost) != hipSuccess) break;
            for (uint64_t i = 0; i < n; ++i) ((uint64_t *)SA)[i] = tmp[(size_t)i];
        }
        rounds = r;
    } while (0);
    pfp_destroy(c);
    return rounds;
}
extern "C++" {
// int gsacak(unsigned char *s, uint_t *SA, int_t *LCP, int_t *DA, uint_t n), gsa/gsacak.h:86-96 -- the call of
// include/pfbwt.hpp:211.  s = strings over any byte alphabet (the parser's dictionaries: '-', A, C, G, N, T and Dollar = 2), each
// followed by the separator 1, s[n-1] = 0.  Suffixes are compared up to their separator; suffixes that are byte-identical up to it
// are ordered by position (gsacak.c:877-912) and LCP stops at the separator (:64).
// LCP[i] of SA[i-1], SA[i]: bytes are compared eight at a time; inside runs of one byte (a 10 Mbp run of N is one phrase whose
// suffixes are neighbours in SA) the shorter of the two runs is skipped at once (M: run lengths, see k_ss_runend_marks)
template <typename LT> __global__ __launch_bounds__(BLOCK) void k_gsa_lcp(const uint8_t *D, const uint32_t *SA, const uint32_t *M, uint64_t n, LT *lcp)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    if (i == 0) { lcp[0] = 0; return; }
    const uint64_t x = SA[i - 1], y = SA[i];
    uint64_t h = 0;
    for (;;) {
        const uint64_t px = x + h, py = y + h;
        if (px >= n || py >= n) break;
        const uint8_t cx = D[px];
        if (cx != D[py] || cx <= EndOfWord) break;
        const uint32_t ix = (uint32_t)(n - 1 - px), iy = (uint32_t)(n - 1 - py);
        const uint64_t rx = ix - M[ix] + 1u, ry = iy - M[iy] + 1u;          // lengths of the runs of cx that start at px / py
        if (rx >= 16 && ry >= 16) { h += rx < ry ? rx : ry; continue; }
        const uint64_t a = ld8(D + px), b = ld8(D + py);                     // (the device buffer is padded)
        // first byte that differs or is a separator / terminator: bytes are < 0x80, so (v - 0x02..02) has its high bit set exactly for v <= 1
        const uint64_t stop = (a ^ b) | (((a - 0x0202020202020202ULL) & ~a & 0x8080808080808080ULL));
        if (stop) { h += (uint64_t)(__ffsll((long long)stop) - 1) >> 3; break; }
        h += 8;
    }
    lcp[i] = (LT)h;
}
template <typename LT> __global__ __launch_bounds__(BLOCK) void k_gsa_da(const uint32_t *SA, const uint32_t *wordid, uint64_t n, LT *da)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) da[i] = (LT)wordid[SA[i]];
}
static int gsacak_impl(const uint8_t *s, void *SA, void *LCP, void *DA, uint64_t n, bool u64)
{
    if (!s || !SA || n < 2) return -1;
    if (n + 64 >= 0xFFFFFFFFULL) return -1;
    // the alphabet: bytes 0 (terminator: only at the very end) and 1 (separator) keep their roles; every other byte that occurs gets
    // the next code in byte order.  The parser's own dictionaries ('-' A C G N T and Dollar) take the tuned 16-characters-per-key
    // kernel; any other byte set (another caller of gsa/gsacak.h:86-96) takes the generic one with as many characters as fit 51 bits
    bool seen[256] = {false};
    for (uint64_t i = 0; i < n; ++i) seen[s[i]] = true;
    if (s[n - 1] != 0) return -1;
    for (uint64_t i = 0; i + 1 < n; ++i) if (s[i] == 0) return -1;      // no unique terminator
    bool dict_alphabet = true;
    for (int b = 3; b < 256; ++b) if (seen[b] && !(b == '-' || b == 'A' || b == 'C' || b == 'G' || b == 'N' || b == 'T')) dict_alphabet = false;
    uint8_t code[256]; uint32_t sigma = 2;
    for (int b = 0; b < 256; ++b) code[b] = (uint8_t)(b <= 1 ? b : 0);
    for (int b = 2; b < 256; ++b) if (seen[b]) code[b] = (uint8_t)sigma++;
    uint32_t chars = 0; { unsigned __int128 v = 1; while (chars < 16 && v * sigma < ((unsigned __int128)1 << 51)) { v *= sigma; ++chars; } }
    int st = 0;
    pfp_ctx *c = pfp_create(10, 100, u64 ? PFP_FLAG_U64 : 0u, 0, (uint64_t)(72 * n + (64ULL << 20)), &st);
    if (!c) return -1;
    int rounds = -1;
    auto body = [&]() -> int {
        PFP_TRY(ensure_arena(c, n));
        c->arena.reset();
        c->dsize = n;
        PFP_ALLOC_LO(c, c->d_dict, uint8_t, n + 16);
        PFP_HIP(c, hipMemsetAsync(c->d_dict + n, 0, 16, c->stream));
        PFP_TRY(h2d_copy(c, c->d_dict, s, n));
        // sort_dict_suffixes with a counted round number
        uint64_t *k0, *k1; uint32_t *v0, *v1;
        PFP_ALLOC_LO(c, c->d_gsa, uint32_t, n); PFP_ALLOC_LO(c, c->d_grank, uint2, n);
        const size_t mk = c->arena.mark_hi();
        PFP_ALLOC_HI(c, k0, uint64_t, n); PFP_ALLOC_HI(c, k1, uint64_t, n); PFP_ALLOC_HI(c, v0, uint32_t, n); PFP_ALLOC_HI(c, v1, uint32_t, n);
        if (dict_alphabet) {
            PFP_LAUNCH(c, K_SS_INIT_KEYS, n * 13, k_dict_init_keys, nblocks(n, DK_TILE), (const uint8_t *)c->d_dict, n, k0, v0);
            BitRange full = {0, DK_KEY_BITS};
            PFP_TRY(suffix_sort_doubling<true>(c, n, k0, v0, k1, v1, &full, 1, DK_CHARS, c->d_dict, c->d_gsa, (uint32_t *)nullptr, c->d_grank, &rounds));
        } else {
            uint8_t *d_code; PFP_ALLOC_HI(c, d_code, uint8_t, 256);
            PFP_HIP(c, hipMemcpyAsync(d_code, code, 256, hipMemcpyHostToDevice, c->stream));
            PFP_HIP(c, hipStreamSynchronize(c->stream));
            PFP_LAUNCH(c, K_SS_INIT_KEYS, n * 13, k_dict_init_keys_any, nblocks(n, BLOCK), (const uint8_t *)c->d_dict, n, (const uint8_t *)d_code, sigma, chars, k0, v0);
            BitRange full = {0, 51};
            PFP_TRY(suffix_sort_doubling<true>(c, n, k0, v0, k1, v1, &full, 1, chars, c->d_dict, c->d_gsa, (uint32_t *)nullptr, c->d_grank, &rounds, sigma));
        }
        c->arena.release_hi(mk);
        PFP_HIP(c, hipStreamSynchronize(c->stream));
        if (!u64) PFP_HIP(c, hipMemcpy(SA, c->d_gsa, n * 4, hipMemcpyDeviceToHost));
        else PFP_TRY(get_u32_as(c, c->d_gsa, n, SA, true));
        if (LCP) {
            uint32_t *M; PFP_ALLOC_HI(c, M, uint32_t, n);
            PFP_LAUNCH(c, K_MISC, n * 5, k_ss_runend_marks, nblocks(n, BLOCK), (const uint8_t *)c->d_dict, n, M);
            PFP_TRY((device_scan<uint32_t, 1>(c, M, M, n, nullptr)));
            if (u64) { int64_t *d; PFP_ALLOC_HI(c, d, int64_t, n); PFP_LAUNCH(c, K_MISC, n * 40, (k_gsa_lcp<int64_t>), nblocks(n, BLOCK), (const uint8_t *)c->d_dict, (const uint32_t *)c->d_gsa, (const uint32_t *)M, n, d);
                       PFP_HIP(c, hipStreamSynchronize(c->stream)); PFP_HIP(c, hipMemcpy(LCP, d, n * 8, hipMemcpyDeviceToHost)); }
            else { int32_t *d; PFP_ALLOC_HI(c, d, int32_t, n); PFP_LAUNCH(c, K_MISC, n * 36, (k_gsa_lcp<int32_t>), nblocks(n, BLOCK), (const uint8_t *)c->d_dict, (const uint32_t *)c->d_gsa, (const uint32_t *)M, n, d);
                   PFP_HIP(c, hipStreamSynchronize(c->stream)); PFP_HIP(c, hipMemcpy(LCP, d, n * 4, hipMemcpyDeviceToHost)); }
            c->arena.release_hi(mk);
        }
        if (DA) {   // document of every suffix = number of separators in front of it
            uint32_t *flag, *wid; PFP_ALLOC_HI(c, flag, uint32_t, n); PFP_ALLOC_HI(c, wid, uint32_t, n);
            PFP_LAUNCH(c, K_MISC, n * 5, k_eow_flags, nblocks(n, BLOCK), (const uint8_t *)c->d_dict, n, flag);
            PFP_TRY((device_scan<uint32_t, 0>(c, flag, wid, n, nullptr)));
            if (u64) { int64_t *d; PFP_ALLOC_HI(c, d, int64_t, n); PFP_LAUNCH(c, K_MISC, n * 16, (k_gsa_da<int64_t>), nblocks(n, BLOCK), (const uint32_t *)c->d_gsa, (const uint32_t *)wid, n, d);
                       PFP_HIP(c, hipStreamSynchronize(c->stream)); PFP_HIP(c, hipMemcpy(DA, d, n * 8, hipMemcpyDeviceToHost)); }
            else { int32_t *d; PFP_ALLOC_HI(c, d, int32_t, n); PFP_LAUNCH(c, K_MISC, n * 12, (k_gsa_da<int32_t>), nblocks(n, BLOCK), (const uint32_t *)c->d_gsa, (const uint32_t *)wid, n, d);
                   PFP_HIP(c, hipStreamSynchronize(c->stream)); PFP_HIP(c, hipMemcpy(DA, d, n * 4, hipMemcpyDeviceToHost)); }
            c->arena.release_hi(mk);
        }
        return PFP_OK;
    };
    const int rc = body();
    pfp_destroy(c);
    return rc == PFP_OK ? rounds : -1;
}
} // extern "C++"
int pfp_gsacak_u32(const uint8_t *s, uint32_t *SA, int32_t *LCP, int32_t *DA, uint32_t n) { return gsacak_impl(s, SA, LCP, DA, n, false); }
int pfp_gsacak_u64(const uint8_t *s, uint64_t *SA, int64_t *LCP, int64_t *DA, uint64_t n) { return gsacak_impl(s, SA, LCP, DA, n, true); }

int pfp_sacak_int_u32(const uint32_t *s, uint32_t *SA, uint32_t n, uint32_t k) { return sacak_int_impl(s, SA, n, k, false); }
int pfp_sacak_int_u64(const uint32_t *s, uint64_t *SA, uint64_t n, uint64_t k) { return sacak_int_impl(s, SA, n, k, true); }

} // extern "C"

#include "sharded.h"
