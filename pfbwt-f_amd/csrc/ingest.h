// pfbwt-f_amd/csrc/ingest.h -- the reading side of PfParser::add_fasta (include/pfparser.hpp:300-307: gzopen / kseq_init /
// kseq_read, include/kseq.h:178-228) for the engine: a file, gzip stream or stdin becomes blocks of raw bytes in page-locked
// memory that cross PCIe while the next blocks are still being read, and are stripped on the device (csrc/fasta.h).
//   * plain regular file: READERS threads pread() 64 MiB blocks straight into a ring of pinned buffers (no second copy)
//   * gzip / stdin: one thread inflates / reads into the same ring (zlib is the bound there, as it is for kseq)
//   * the consumer (caller's thread) uploads block k + 1 before it waits for the few bytes of totals of block k, so the
//     copy engine never idles while the host learns how far the text grew
//   * FASTQ ('@' first, or a '+' line): record by record on the host, like kseq (quality lines skipped), through pfp_parse_feed
// Host code only; included by pfbwt_hip.hip.
#pragma once
#include <condition_variable>
#include <fcntl.h>
#include <mutex>
#include <cstdio>
#include <sched.h>
#include <string>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <zlib.h>

namespace pfp {

constexpr size_t ING_BLOCK = (size_t)64 << 20;
// CPUs this process can really use: the affinity mask, capped by the cgroup's CPU quota (a one-GPU job of the pool sees 256 CPUs and
// may use 16 of them: 64 writer threads on such a box throttle each other)
static int usable_cpus()
{
    cpu_set_t cs; CPU_ZERO(&cs);
    int n = sched_getaffinity(0, sizeof cs, &cs) == 0 ? CPU_COUNT(&cs) : (int)std::thread::hardware_concurrency();
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {      // cgroup v2: "<quota> <period>" or "max <period>"
        char q[32]; long period = 0;
        if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") && period > 0) { const long cpus = (atol(q) + period - 1) / period; if (cpus >= 1 && cpus < n) n = (int)cpus; }
        fclose(f);
    }
    return n < 1 ? 1 : n;
}
constexpr int ING_RING = 16;      // the ring is a multiple of the readers (4, 8 or 16): a slot is always filled by the same thread
// readers of a plain file: one per CPU the process may run on, at most 16 (a pread from the page cache is a copy at ~5 GB/s per thread: eight
// threads gave 35 GB/s on the driver's box where the link does 57 -- VERDICT r3)
inline int ingest_readers(const pfp_ctx *c)
{
    int want = c->tun.ingest_readers;
    if (want <= 0) want = usable_cpus();
    return want >= 16 ? 16 : want >= 8 ? 8 : 4;
}

struct IngestStats { uint64_t raw_bytes = 0, records = 0; double read_wait_ms = 0, total_ms = 0; int mode = 0; };

// ---- FASTQ / fallback: kseq's record rules on the host --------------------------------------------------------------
struct HostRecordReader {
    gzFile fp = nullptr; const uint8_t *pre = nullptr; size_t pre_len = 0, pre_pos = 0;      // bytes already read by the caller come first
    std::vector<char> buf; int pos = 0, len = 0, pending = 0;
    int getc_()
    {
        if (pre_pos < pre_len) return pre[pre_pos++];
        if (pos == len) { if (!fp) return -1; len = gzread(fp, buf.data(), (unsigned)buf.size()); pos = 0; if (len <= 0) { len = 0; return -1; } }
        return (unsigned char)buf[pos++];
    }
    bool next(std::string &name, std::string &seq)
    {
        int ch;
        if (!pending) { while ((ch = getc_()) >= 0 && ch != '>' && ch != '@') {} if (ch < 0) return false; }
        pending = 0; name.clear(); seq.clear();
        bool in_name = true;
        while ((ch = getc_()) >= 0 && ch != '\n') { if (in_name) { if (ch == ' ' || ch == '\t' || ch == '\r') in_name = false; else name.push_back((char)ch); } }
        bool line_start = true;
        while ((ch = getc_()) >= 0) {
            if (line_start && (ch == '>' || ch == '@')) { pending = ch; return true; }
            if (line_start && ch == '+') break;
            if (ch == '\n') { line_start = true; continue; }
            line_start = false;
            if (ch != '\r') seq.push_back((char)ch);
        }
        if (ch == '+') {   // kseq.h:209-221: skip the '+' line, then as many quality characters as bases
            while ((ch = getc_()) >= 0 && ch != '\n') {}
            size_t q = 0;
            while (q < seq.size() && (ch = getc_()) >= 0) if (ch != '\n' && ch != '\r') ++q;
        }
        return true;
    }
};

// ---- ring of page-locked blocks filled by reader threads ---------------------------------------------------------------
struct BlockRing {
    uint8_t *buf[ING_RING] = {}; size_t len[ING_RING] = {};
    uint64_t filled[ING_RING] = {};          // block index + 1 that the slot holds (0: nothing yet)
    std::mutex mu; std::condition_variable cv;
    uint64_t consumed = 0;                   // blocks the consumer is done with (block b may be read into its slot once b < consumed + ING_RING)
    uint64_t nblocks = ~0ULL;                // known for regular files; set by the reader at end of stream otherwise
    bool fail = false, stop = false;
};

// names of --print-docs (pfparser.hpp:321-325): the header's first word; a header line may be cut by a block boundary
static void ingest_names(pfp_ctx *c, const uint8_t *blk, uint64_t len, uint64_t blk_off, bool *name_open)
{
    auto take = [&](uint64_t from) {
        std::string &nm = c->doc_names.back();
        uint64_t i = from;
        for (; i < len; ++i) { const uint8_t ch = blk[i]; if (ch == ' ' || ch == '\t' || ch == '\r' || ch == '\n') break; nm.push_back((char)ch); }
        *name_open = i == len;
    };
    if (*name_open) take(0);
    for (size_t k = 0; k < c->fa.rec_raw.size(); ++k) {
        c->doc_names.emplace_back(); c->doc_starts.push_back(c->fa.rec_pos[k]);
        take(c->fa.rec_raw[k] - blk_off + 1);
    }
}

// FASTQ (or any input the device reader does not take): records one by one through pfp_parse_feed's staging ring
static int ingest_records(pfp_ctx *c, HostRecordReader &rd, bool want_docs, IngestStats *st)
{
    std::string name, seq;
    while (rd.next(name, seq)) {
        if (want_docs) { c->doc_names.push_back(name); c->doc_starts.push_back(c->n); }
        PFP_TRY(feed_common(c, seq.data(), seq.size(), 1, hipMemcpyHostToDevice));
        st->records++;
    }
    st->mode = 3;
    return PFP_OK;
}

static int ingest_file(pfp_ctx *c, const char *path, unsigned flags, IngestStats *st)
{
    HostTimer timer;
    const size_t BLK = (c->tun.ingest_block_bytes && c->tun.ingest_block_bytes < ING_BLOCK) ? (size_t)c->tun.ingest_block_bytes : ING_BLOCK;      // tests: many blocks on small files
    const bool want_docs = (flags & PFP_FASTA_RECORDS) != 0;
    const bool is_stdin = !strcmp(path, "-");
    c->doc_names.clear(); c->doc_starts.clear();
    int fd = is_stdin ? 0 : open(path, O_RDONLY);
    if (fd < 0) return PFP_E_IO;
    struct stat sb; const bool regular = fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode);
    bool gz = false;
    if (regular) { unsigned char mg[2] = {0, 0}; if (pread(fd, mg, 2, 0) == 2 && mg[0] == 0x1f && mg[1] == 0x8b) gz = true; }
    const bool parallel = regular && !gz;                                    // preads at independent offsets
    gzFile gzf = nullptr;
    if (!parallel) { gzf = gzdopen(fd, "r"); if (!gzf) { if (!is_stdin) close(fd); return PFP_E_IO; } gzbuffer(gzf, 1 << 20); }
    if (regular && !gz && !c->text.live() && !c->text_hint) c->text_hint = (uint64_t)sb.st_size;
    auto &f = c->fa;
    // page-locked ring (kept by the context: a second file re-uses it); a slot's buffer is allocated by the reader that first fills
    // it -- the readers allocate in parallel, a cold process does not wait for 1 GiB of page-locked memory before the first pread
    BlockRing ring;
    for (int k = 0; k < ING_RING; ++k) ring.buf[k] = c->ing_buf[k];
    if (parallel) ring.nblocks = ((uint64_t)sb.st_size + BLK - 1) / BLK;
    auto fill = [&](uint64_t b, int slot) -> bool {      // reader side: block b into its slot; false at a read error
        size_t got = 0;
        if (!ring.buf[slot]) {
            (void)hipSetDevice(c->device);
            if (hipHostMalloc((void **)&c->ing_buf[slot], ING_BLOCK, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return false; }
            ring.buf[slot] = c->ing_buf[slot];
        }
        if (parallel) {
            const uint64_t o = b * BLK, want = (uint64_t)sb.st_size - o < BLK ? (uint64_t)sb.st_size - o : BLK;
            while (got < want) { const ssize_t r = pread(fd, ring.buf[slot] + got, (size_t)(want - got), (off_t)(o + got)); if (r < 0) return false; if (r == 0) break; got += (size_t)r; }
        } else {
            while (got < BLK) { const int r = gzread(gzf, ring.buf[slot] + got, (unsigned)(BLK - got < ((size_t)1 << 30) ? BLK - got : ((size_t)1 << 30))); if (r < 0) return false; if (r == 0) break; got += (size_t)r; }
        }
        ring.len[slot] = got;
        return true;
    };
    const int nreaders = parallel ? ingest_readers(c) : 1;
    auto reader = [&](int tid) {
        for (uint64_t b = (uint64_t)tid;; b += (uint64_t)nreaders) {
            const int slot = (int)(b % ING_RING);
            {
                std::unique_lock<std::mutex> lk(ring.mu);
                ring.cv.wait(lk, [&] { return ring.stop || ring.fail || b >= ring.nblocks || b < ring.consumed + ING_RING; });
                if (ring.stop || ring.fail || b >= ring.nblocks) return;
            }
            const bool ok = fill(b, slot);
            std::lock_guard<std::mutex> lk(ring.mu);
            if (!ok) ring.fail = true;
            else { ring.filled[slot] = b + 1; if (!parallel && ring.len[slot] < BLK) ring.nblocks = b + 1; }      // a short block ends a stream
            ring.cv.notify_all();
            if (!ok) return;
        }
    };
    std::vector<std::thread> threads;
    for (int t = 0; t < nreaders; ++t) threads.emplace_back(reader, t);
    auto shutdown = [&]() { { std::lock_guard<std::mutex> lk(ring.mu); ring.stop = true; } ring.cv.notify_all(); for (auto &t : threads) t.join(); threads.clear(); };
    // consumer
    int rc = PFP_OK; bool name_open = false, first = true, fastq = false;
    uint64_t b = 0; bool issued_next = false;
    auto wait_block = [&](uint64_t blk, bool block) -> int {      // 1 ready, 0 not yet / end of stream, -1 error
        std::unique_lock<std::mutex> lk(ring.mu);
        const int slot = (int)(blk % ING_RING);
        if (block) { HostTimer tw; ring.cv.wait(lk, [&] { return ring.fail || blk >= ring.nblocks || ring.filled[slot] == blk + 1; }); st->read_wait_ms += tw.ms(); }
        if (ring.fail) return -1;
        return ring.filled[slot] == blk + 1 ? 1 : 0;
    };
    auto blk_range = [&](uint64_t blk, const uint8_t **p, uint64_t *l, uint64_t *skip) {      // the part of block blk that counts (behind the stream's preamble)
        const int slot = (int)(blk % ING_RING);
        *p = ring.buf[slot]; *l = ring.len[slot]; *skip = 0;
    };
    while (rc == PFP_OK) {
        const int ready = issued_next ? 1 : wait_block(b, true);
        if (ready < 0) { rc = PFP_E_IO; break; }
        if (ready == 0) break;                                                  // end of stream
        const uint8_t *p; uint64_t l, skip; blk_range(b, &p, &l, &skip);
        st->raw_bytes += l;
        uint64_t off = 0;
        if (!issued_next) {
            off = fa_skip_preamble(c, p, l);
            if (first && off < l && p[off] == '@') { fastq = true; break; }     // FASTQ: on the host, record by record
            if (off < l) first = false;
            if (off < l) { rc = fasta_buffers(c, BLK); if (rc == PFP_OK && f.rawcap < BLK) rc = PFP_E_ARG; if (rc == PFP_OK) rc = fa_issue(c, p + off, l - off, (int)(b & 1)); if (rc != PFP_OK) break; }
        }
        const uint64_t my_off = issued_next ? c->ing_next_off : off;
        issued_next = false;
        if (my_off < l) {
            // the next block's upload goes out before this block's totals are waited for (if the readers already have it)
            if (wait_block(b + 1, false) == 1 && f.started) {
                const uint8_t *p2; uint64_t l2, s2; blk_range(b + 1, &p2, &l2, &s2);
                if (l2) { rc = fa_issue(c, p2, l2, (int)((b + 1) & 1)); if (rc != PFP_OK) break; issued_next = true; c->ing_next_off = 0; }
            }
            f.rec_raw.clear(); f.rec_pos.clear();
            rc = fa_process(c, l - my_off, (int)(b & 1), want_docs, b * BLK + my_off, &st->records);
            if (rc != PFP_OK) break;
            if (want_docs) ingest_names(c, p, l, b * BLK, &name_open);
        } else if (want_docs && name_open) ingest_names(c, p, l, b * BLK, &name_open);
        { std::lock_guard<std::mutex> lk(ring.mu); ring.consumed = b + 1; }
        ring.cv.notify_all();
        ++b;
    }
    if (fastq) {
        // hand the stream to the host record reader: the block already read comes first, the rest through zlib (regular plain
        // files are re-opened by name; the reader threads are stopped first)
        shutdown();
        HostRecordReader rd; rd.buf.resize(1 << 16);
        std::vector<uint8_t> head;
        if (parallel) { rd.fp = gzopen(path, "r"); if (!rd.fp) rc = PFP_E_IO; }
        else {      // the blocks the reader thread got to before it was stopped, in order; the stream goes on behind the last one if that was a full block
            bool more = true;
            for (uint64_t blk = b;; ++blk) { const int slot = (int)(blk % ING_RING); if (ring.filled[slot] != blk + 1) break; head.insert(head.end(), ring.buf[slot], ring.buf[slot] + ring.len[slot]); more = ring.len[slot] == BLK; }
            rd.pre = head.data(); rd.pre_len = head.size(); rd.fp = more ? gzf : nullptr;
        }
        c->fa.started = false; c->fa.state = 2; c->fa.records = 0;
        if (rc == PFP_OK) rc = ingest_records(c, rd, want_docs, st);
        if (parallel && rd.fp) gzclose(rd.fp);
    } else {
        shutdown();
        if (rc == PFP_OK) rc = fa_finish_stream(c);
        st->mode = parallel ? 1 : 2;
    }
    if (gzf) gzclose(gzf); else if (!is_stdin) close(fd);
    if (rc != PFP_OK) { if (f.copy_ready) (void)hipStreamSynchronize(f.copy); (void)hipStreamSynchronize(c->stream); }      // no upload of a ring block is left in flight behind an error return
    st->total_ms = timer.ms();
    return rc;
}

// ---- the way out: a device-resident result goes to a file descriptor through the same ring of page-locked blocks -- the DMA
// transfer of block k + 1 overlaps the write of block k; a regular file is written by several threads with pwrite at the blocks'
// own offsets, a pipe (`-c bwt`: stdout) by one thread in order.  Stands in for the per-row fwrite of out_fn,
// src/pfbwt-f.cpp:298-328 (two to four calls per base there).
static int write_device_to_fd(pfp_ctx *c, const void *d_src, uint64_t bytes, int fd)
{
    if (!bytes) return PFP_OK;
    PFP_TRY(ensure_copy_stream(c));
    constexpr int NB = ING_RING;
    const off_t base = lseek(fd, 0, SEEK_CUR);
    // pwrite at computed offsets only where offsets mean something: a regular file that was not opened for appending (Linux ignores
    // pwrite's offset on an O_APPEND descriptor: `pfbwt-f -c bwt ... >> out` would get its blocks in completion order; the reference
    // writes sequentially with fwrite, src/pfbwt-f.cpp:298-328) -- everything else gets ONE in-order writer
    struct stat st; const int fl = fcntl(fd, F_GETFL);
    const bool seekable = base != (off_t)-1 && fl != -1 && !(fl & O_APPEND) && fstat(fd, &st) == 0 && S_ISREG(st.st_mode);
    const int nw = seekable ? 4 : 1;      // (measured on a memory-resident file system: 4 writers 6 GB/s, 8 writers 4.3 GB/s -- page allocation, not the copy, bounds a pwrite)
    for (int k = 0; k < NB; ++k) if (!c->ing_buf[k]) PFP_HIP(c, hipHostMalloc((void **)&c->ing_buf[k], ING_BLOCK, hipHostMallocDefault));
    hipEvent_t ev[NB];
    for (int k = 0; k < NB; ++k) PFP_HIP(c, hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
    const uint64_t nblk = (bytes + ING_BLOCK - 1) / ING_BLOCK;
    std::mutex mu; std::condition_variable cv;
    uint64_t issued = 0, written[NB]; bool fail = false;          // written[s]: blocks of slot s that are on the file (a slot is re-used when its previous block is)
    for (auto &x : written) x = 0;
    auto writer = [&](int tid) {
        for (uint64_t b = (uint64_t)tid; b < nblk; b += (uint64_t)nw) {
            const int slot = (int)(b % NB);
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return fail || issued > b; }); if (fail) return; }
            bool ok = hipEventSynchronize(ev[slot]) == hipSuccess;
            const uint64_t off = b * ING_BLOCK, len = bytes - off < ING_BLOCK ? bytes - off : ING_BLOCK;
            for (uint64_t done = 0; ok && done < len;) {
                const ssize_t r = seekable ? pwrite(fd, c->ing_buf[slot] + done, (size_t)(len - done), base + (off_t)(off + done)) : write(fd, c->ing_buf[slot] + done, (size_t)(len - done));
                if (r <= 0) ok = false; else done += (uint64_t)r;
            }
            std::lock_guard<std::mutex> lk(mu);
            if (!ok) fail = true;
            written[slot] += 1;
            cv.notify_all();
            if (!ok) return;
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < nw; ++t) th.emplace_back(writer, t);
    int rc = PFP_OK;
    for (uint64_t b = 0; b < nblk && rc == PFP_OK; ++b) {
        const int slot = (int)(b % NB);
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return fail || written[slot] >= b / NB; }); if (fail) { rc = PFP_E_IO; break; } }
        const uint64_t off = b * ING_BLOCK, len = bytes - off < ING_BLOCK ? bytes - off : ING_BLOCK;
        if (hipMemcpyAsync(c->ing_buf[slot], (const uint8_t *)d_src + off, (size_t)len, hipMemcpyDeviceToHost, c->fa.copy) != hipSuccess || hipEventRecord(ev[slot], c->fa.copy) != hipSuccess) { (void)hipGetLastError(); rc = PFP_E_HIP; }
        std::lock_guard<std::mutex> lk(mu);
        if (rc != PFP_OK) fail = true; else issued = b + 1;
        cv.notify_all();
    }
    { std::lock_guard<std::mutex> lk(mu); if (rc != PFP_OK) fail = true; } cv.notify_all();
    for (auto &t : th) t.join();
    (void)hipStreamSynchronize(c->fa.copy);
    for (int k = 0; k < NB; ++k) (void)hipEventDestroy(ev[k]);
    if (rc == PFP_OK && fail) rc = PFP_E_IO;
    if (rc == PFP_OK && seekable && lseek(fd, base + (off_t)bytes, SEEK_SET) == (off_t)-1) rc = PFP_E_IO;
    return rc;
}

} // namespace pfp
