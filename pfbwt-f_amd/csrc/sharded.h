// pfbwt-f_amd/csrc/sharded.h -- the sharded build for a C / C++ caller: N devices of one node driven from N host threads of ONE
// process, the dictionaries exchanged by RCCL called directly (ncclCommInitAll + one ncclAllGather per rank; no torch, no launcher).
//
// Reference: the only parallelism of the reference is of this shape -- src/merge_pfp.cpp:131-152 gives every std::thread its own
// PfParser over a contiguous slice of the inputs, folds the per-thread parsers with PfParser::operator+= (include/pfparser.hpp:194-263)
// and goes on single-threaded.  Here thread r owns GPU r: it parses its run of whole sequences (with the w 'A's that end the run in
// front of it as left context, pfparser.hpp:335-337), packs {dictionary, word starts, phrase ids} -- 4 bytes per phrase + the
// dictionary --, all ranks exchange the packs in ONE all-gather over xGMI, every rank merges them (pfp_merge_shards: the union
// dictionary, the seams re-hashed as :226-245 does), sorts dictionary and parse (redundant, identical work), and
// emits ITS slice of the output rows (pfp_bwt_build_slice): .bwt / .sa / run samples stay distributed over the GPUs' HBM, in slice
// order they are the reference's files.  The Python recipe pfbwt-f_amd/python/pfbwt_dist.py is the same protocol between processes.
//
// RCCL is looked up at run time (dlopen of librccl.so) the first time distinct devices have to talk: a single-GPU user of the library
// never loads it.  Contexts that share one device (a rehearsal of the N-rank protocol on one card: the tests of a one-GPU box)
// exchange their packs by device-to-device copies instead -- same buffers, same layout, no collective library involved.
#pragma once
#include <condition_variable>
#include <mutex>
#include <thread>
#ifndef PFBWT_EMU_HIP_RUNTIME_H
#include <dlfcn.h>
#include <rccl/rccl.h>
#endif

struct pfp_sharded {
    int ndev = 0, w = 10; uint64_t p = 100; unsigned flags = 0;
    std::vector<int> dev;
    std::vector<pfp_ctx *> ctx;
    bool distinct = true;                       // every rank has its own device: RCCL; otherwise device-to-device copies
#ifndef PFBWT_EMU_HIP_RUNTIME_H
    std::vector<ncclComm_t> comm;
    void *rccl = nullptr;
    ncclResult_t (*p_init_all)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*p_all_gather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*p_destroy)(ncclComm_t) = nullptr;
    const char *(*p_errstr)(ncclResult_t) = nullptr;
#endif
    // per-rank send / receive buffers of the exchange (device memory of the rank's GPU; kept between builds, grown on demand)
    std::vector<uint8_t *> sendbuf, recvbuf; std::vector<size_t> sendcap, recvcap;
    char err[256] = {0};
};

namespace pfp {

struct ShardMeta { uint64_t n, m, dwords, dsize, bytes, left_context; };
inline size_t sh_align(size_t x) { return (x + 255) & ~(size_t)255; }
inline void sh_sizes(const pfp_shard_view &v, size_t b[3]) { b[0] = (size_t)v.dsize; b[1] = ((size_t)v.dwords + 1) * 4; b[2] = (size_t)v.m * 4; }

static int sharded_reopen(pfp_sharded *s)
{
    for (int r = 0; r < s->ndev; ++r) {
        PFP_TRY(pfp_reset(s->ctx[(size_t)r]));
        if (r > 0) PFP_TRY(pfp_parse_feed_left_context(s->ctx[(size_t)r]));
    }
    return PFP_OK;
}

// rank r: finalize the shard, pack {dictionary, word starts, phrase ids} into its send buffer
static int sharded_phase_pack(pfp_sharded *s, int r, ShardMeta *meta)
{
    pfp_ctx *c = s->ctx[(size_t)r];
    pfp_parse_sizes sz;
    PFP_TRY(pfp_parse_finalize_shard(c, &sz));
    pfp_shard_view v; PFP_TRY(pfp_shard_view_get(c, &v));
    size_t b[3]; sh_sizes(v, b);
    const size_t tot = sh_align(b[0]) + sh_align(b[1]) + sh_align(b[2]);
    PFP_HIP(c, hipSetDevice(c->device));
    if (s->sendcap[(size_t)r] < tot) {
        if (s->sendbuf[(size_t)r]) PFP_HIP(c, hipFree(s->sendbuf[(size_t)r]));
        s->sendbuf[(size_t)r] = nullptr; s->sendcap[(size_t)r] = 0;
        if (hipMalloc((void **)&s->sendbuf[(size_t)r], tot) != hipSuccess) { (void)hipGetLastError(); return PFP_E_NOMEM; }
        s->sendcap[(size_t)r] = tot;
    }
    uint8_t *q = s->sendbuf[(size_t)r];
    PFP_TRY(pfp_device_copy(c, q, v.d_dict, b[0])); q += sh_align(b[0]);
    PFP_TRY(pfp_device_copy(c, q, v.d_ws, b[1])); q += sh_align(b[1]);
    PFP_TRY(pfp_device_copy(c, q, v.d_pid, b[2]));
    PFP_HIP(c, hipStreamSynchronize(c->stream));
    *meta = {v.n, v.m, v.dwords, v.dsize, (uint64_t)tot, v.left_context};
    return PFP_OK;
}

// rank r: receive every rank's pack (all-gather, padded to the largest), merge, sort, emit slice r
static int sharded_phase_exchange(pfp_sharded *s, int r, const ShardMeta *meta, size_t maxb)
{
    pfp_ctx *c = s->ctx[(size_t)r];
    const int N = s->ndev;
    PFP_HIP(c, hipSetDevice(c->device));
    if (s->sendcap[(size_t)r] < maxb) {      // the collective sends maxb bytes from every rank
        uint8_t *nb = nullptr;
        if (hipMalloc((void **)&nb, maxb) != hipSuccess) { (void)hipGetLastError(); return PFP_E_NOMEM; }
        if (hipMemcpy(nb, s->sendbuf[(size_t)r], (size_t)meta[r].bytes, hipMemcpyDeviceToDevice) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(nb); return PFP_E_HIP; }
        (void)hipFree(s->sendbuf[(size_t)r]); s->sendbuf[(size_t)r] = nb; s->sendcap[(size_t)r] = maxb;
    }
    if (s->recvcap[(size_t)r] < maxb * (size_t)N) {
        if (s->recvbuf[(size_t)r]) PFP_HIP(c, hipFree(s->recvbuf[(size_t)r]));
        s->recvbuf[(size_t)r] = nullptr; s->recvcap[(size_t)r] = 0;
        if (hipMalloc((void **)&s->recvbuf[(size_t)r], maxb * (size_t)N) != hipSuccess) { (void)hipGetLastError(); return PFP_E_NOMEM; }
        s->recvcap[(size_t)r] = maxb * (size_t)N;
    }
    return PFP_OK;
}
static int sharded_phase_gather(pfp_sharded *s, int r, const ShardMeta *meta, size_t maxb)
{
    pfp_ctx *c = s->ctx[(size_t)r];
    const int N = s->ndev;
    PFP_HIP(c, hipSetDevice(c->device));
#ifndef PFBWT_EMU_HIP_RUNTIME_H
    if (s->distinct) {
        const ncclResult_t e = s->p_all_gather(s->sendbuf[(size_t)r], s->recvbuf[(size_t)r], maxb, ncclUint8, s->comm[(size_t)r], c->stream);
        if (e != ncclSuccess) {      // several ranks may fail at once: one writer of the shared message
            static std::mutex err_mu; std::lock_guard<std::mutex> g(err_mu);
            if (!s->err[0]) snprintf(s->err, sizeof s->err, "rank %d: ncclAllGather: %s", r, s->p_errstr ? s->p_errstr(e) : "error");
            return PFP_E_HIP;
        }
        PFP_HIP(c, hipStreamSynchronize(c->stream));
        return PFP_OK;
    }
#endif
    // contexts on one device: every pack is copied straight out of its owner's send buffer (all packs are complete: the caller
    // passed the barrier behind the pack phase)
    for (int q = 0; q < N; ++q)
        PFP_HIP(c, hipMemcpyAsync(s->recvbuf[(size_t)r] + (size_t)q * maxb, s->sendbuf[(size_t)q], (size_t)meta[q].bytes, hipMemcpyDeviceToDevice, c->stream));
    PFP_HIP(c, hipStreamSynchronize(c->stream));
    return PFP_OK;
}
static int sharded_phase_build(pfp_sharded *s, int r, const ShardMeta *meta, size_t maxb, int want_sa, int want_rssa, pfp_parse_sizes *psz, pfp_bwt_sizes *bsz,
                               uint64_t *slice_begin, uint64_t *slice_rows, uint64_t *esa_pairs)
{
    pfp_ctx *c = s->ctx[(size_t)r];
    const int N = s->ndev;
    std::vector<pfp_shard_view> views((size_t)N);
    for (int q = 0; q < N; ++q) {
        pfp_shard_view &v = views[(size_t)q];
        memset(&v, 0, sizeof v);
        v.n = meta[q].n; v.m = meta[q].m; v.dwords = meta[q].dwords; v.dsize = meta[q].dsize; v.left_context = meta[q].left_context;
        size_t b[3]; sh_sizes(v, b);
        const uint8_t *base = s->recvbuf[(size_t)r] + (size_t)q * maxb;
        v.d_dict = base; v.d_ws = (const uint32_t *)(base + sh_align(b[0])); v.d_pid = (const uint32_t *)(base + sh_align(b[0]) + sh_align(b[1]));
        v.d_ye = nullptr; v.d_last = nullptr;      // compact views: the merge derives phrase ends and last bytes
    }
    pfp_parse_sizes sz;
    PFP_TRY(pfp_merge_shards(c, N, views.data(), &sz));
    if (psz) *psz = sz;
    PFP_TRY(pfp_parse_bwt(c));
    uint64_t sb = 0, sr = 0, ep = 0; pfp_bwt_sizes bs;
    PFP_TRY(pfp_bwt_build_slice(c, want_sa, want_rssa, r, N, &bs, &sb, &sr, &ep));
    if (bsz) bsz[r] = bs;
    if (slice_begin) slice_begin[r] = sb;
    if (slice_rows) slice_rows[r] = sr;
    if (esa_pairs) esa_pairs[r] = ep;
    return PFP_OK;
}

} // namespace pfp

extern "C" {

pfp_sharded *pfp_sharded_create(int w, uint64_t p, unsigned flags, int ndev, const int *devices, uint64_t workspace_bytes, int *status)
{
    using namespace pfp;
    int st = PFP_OK;
    pfp_sharded *s = nullptr;
    if (ndev < 1 || ndev > 64) st = PFP_E_ARG;
    else {
        s = new pfp_sharded();
        s->ndev = ndev; s->w = w; s->p = p; s->flags = flags;
        for (int r = 0; r < ndev; ++r) s->dev.push_back(devices ? devices[r] : r);
        for (int a = 0; a < ndev; ++a) for (int b = a + 1; b < ndev; ++b) if (s->dev[(size_t)a] == s->dev[(size_t)b]) s->distinct = false;
        s->sendbuf.assign((size_t)ndev, nullptr); s->recvbuf.assign((size_t)ndev, nullptr); s->sendcap.assign((size_t)ndev, 0); s->recvcap.assign((size_t)ndev, 0);
        for (int r = 0; r < ndev && st == PFP_OK; ++r) {
            pfp_ctx *c = pfp_create(w, p, flags, s->dev[(size_t)r], workspace_bytes, &st);
            if (c) s->ctx.push_back(c);
        }
#ifndef PFBWT_EMU_HIP_RUNTIME_H
        if (st == PFP_OK && s->distinct) {
            s->rccl = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
            if (!s->rccl) s->rccl = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
            if (s->rccl) {
                s->p_init_all = (decltype(s->p_init_all))dlsym(s->rccl, "ncclCommInitAll");
                s->p_all_gather = (decltype(s->p_all_gather))dlsym(s->rccl, "ncclAllGather");
                s->p_destroy = (decltype(s->p_destroy))dlsym(s->rccl, "ncclCommDestroy");
                s->p_errstr = (decltype(s->p_errstr))dlsym(s->rccl, "ncclGetErrorString");
            }
            if (!s->rccl || !s->p_init_all || !s->p_all_gather || !s->p_destroy) {
                const char *why = dlerror();      // one call: dlerror() clears the message it returns
                fprintf(stderr, "[pfbwt_hip] pfp_sharded_create: librccl.so not available (%s)\n", why ? why : "symbols missing"); st = PFP_E_HIP;
            }
            else {
                s->comm.assign((size_t)ndev, nullptr);
                const ncclResult_t e = s->p_init_all(s->comm.data(), ndev, s->dev.data());
                if (e != ncclSuccess) { fprintf(stderr, "[pfbwt_hip] ncclCommInitAll: %s\n", s->p_errstr ? s->p_errstr(e) : "error"); s->comm.clear(); st = PFP_E_HIP; }
            }
        }
#endif
        if (st == PFP_OK) st = sharded_reopen(s);
        if (st != PFP_OK) { pfp_sharded_destroy(s); s = nullptr; }
    }
    if (status) *status = st;
    return s;
}

void pfp_sharded_destroy(pfp_sharded *s)
{
    if (!s) return;
#ifndef PFBWT_EMU_HIP_RUNTIME_H
    for (auto cm : s->comm) if (cm && s->p_destroy) (void)s->p_destroy(cm);
#endif
    for (size_t r = 0; r < s->ctx.size(); ++r) {
        (void)hipSetDevice(s->ctx[r]->device);
        (void)hipStreamSynchronize(s->ctx[r]->stream);
        if (r < s->sendbuf.size() && s->sendbuf[r]) (void)hipFree(s->sendbuf[r]);
        if (r < s->recvbuf.size() && s->recvbuf[r]) (void)hipFree(s->recvbuf[r]);
    }
    for (auto c : s->ctx) pfp_destroy(c);
#ifndef PFBWT_EMU_HIP_RUNTIME_H
    if (s->rccl) dlclose(s->rccl);
#endif
    delete s;
}

int pfp_sharded_ranks(pfp_sharded *s) { return s ? s->ndev : 0; }
pfp_ctx *pfp_sharded_ctx(pfp_sharded *s, int rank) { return (s && rank >= 0 && rank < s->ndev) ? s->ctx[(size_t)rank] : nullptr; }
const char *pfp_sharded_error(pfp_sharded *s) { return s ? s->err : ""; }
int pfp_sharded_reset(pfp_sharded *s) { return s ? pfp::sharded_reopen(s) : PFP_E_ARG; }

int pfp_sharded_build(pfp_sharded *s, int want_sa, int want_rssa, pfp_parse_sizes *psz, pfp_bwt_sizes *bsz, uint64_t *slice_begin, uint64_t *slice_rows, uint64_t *esa_pairs)
{
    using namespace pfp;
    if (!s) return PFP_E_ARG;
    s->err[0] = 0;      // the message belongs to this build, not to an earlier one that failed
    const int N = s->ndev;
    std::vector<ShardMeta> meta((size_t)N);
    std::vector<int> rc((size_t)N, PFP_OK);
    std::vector<pfp_parse_sizes> ps((size_t)N);
    size_t maxb = 0;
    // Known limit: a rank whose ncclAllGather call itself fails (phase 2) leaves the others waiting inside the collective -- the
    // barriers only cover failures in front of it (parse, pack, allocation); a watchdog with ncclCommAbort is not built.
    // the phases of one rank; between them every rank waits for all the others (the packs must be complete before they are
    // read, the receive buffers allocated before a collective writes to them) and learns whether one of them failed -- a rank
    // that cannot parse its shard must not leave the others waiting inside the all-gather
    auto any_failed = [&]() { for (int q = 0; q < N; ++q) if (rc[(size_t)q] != PFP_OK) return true; return false; };
    auto phase = [&](int r, int ph) {
        if (rc[(size_t)r] != PFP_OK) return;
        switch (ph) {
        case 0: rc[(size_t)r] = sharded_phase_pack(s, r, &meta[(size_t)r]); break;
        case 1: rc[(size_t)r] = sharded_phase_exchange(s, r, meta.data(), maxb); break;
        case 2: rc[(size_t)r] = sharded_phase_gather(s, r, meta.data(), maxb); break;
        default: rc[(size_t)r] = sharded_phase_build(s, r, meta.data(), maxb, want_sa, want_rssa, &ps[(size_t)r], bsz, slice_begin, slice_rows, esa_pairs); break;
        }
    };
    auto between = [&](int ph) { if (ph == 0) { maxb = 0; for (int q = 0; q < N; ++q) if (rc[(size_t)q] == PFP_OK && (size_t)meta[(size_t)q].bytes > maxb) maxb = (size_t)meta[(size_t)q].bytes; } };
#ifdef PFBWT_EMU_HIP_RUNTIME_H
    // the CPU interpreter of the tests runs one kernel at a time: the ranks take every phase in turn
    for (int ph = 0; ph < 4 && !any_failed(); ++ph) { for (int r = 0; r < N; ++r) phase(r, ph); between(ph); }
#else
    std::mutex mu; std::condition_variable cv; int arrived = 0, generation = 0; bool stop = false;
    auto barrier = [&](int ph) -> bool {      // returns false when the build is given up
        std::unique_lock<std::mutex> lk(mu);
        const int gen = generation;
        if (++arrived == N) { between(ph); if (any_failed()) stop = true; arrived = 0; ++generation; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != gen; });
        return !stop;
    };
    std::vector<std::thread> th;
    for (int r = 0; r < N; ++r)
        th.emplace_back([&, r] { for (int ph = 0; ph < 4; ++ph) { phase(r, ph); if (ph < 3 && !barrier(ph)) return; } });
    for (auto &t : th) t.join();
#endif
    int first = PFP_OK;
    for (int q = 0; q < N; ++q) if (rc[(size_t)q] != PFP_OK) { first = rc[(size_t)q]; if (!s->err[0]) snprintf(s->err, sizeof s->err, "rank %d: %s", q, pfp_strerror(first)); break; }
    if (first == PFP_OK && psz) *psz = ps[0];
    return first;
}

} // extern "C"
