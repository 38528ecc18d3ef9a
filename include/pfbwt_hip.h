/*
 * include/pfbwt_hip.h -- C ABI of libpfbwt_hip.so, the MI355X (gfx950) engine for the hot path of
 * alshai/pfbwt-f: prefix-free parse, BWT of the parse, dictionary suffix sort and BWT/SA emission.
 *
 * Plain pointers and sizes only (no C++/torch types), so the reference's host code can bind it:
 * INTEGRATION.md shows the replacement bodies for include/pfparser.hpp and include/pfbwt.hpp.
 * Every entry point names the reference interface it stands in for (file:line under the
 * reference tree).  All entry points return PFP_OK (0) or a negative pfp_status; none calls exit().
 *
 * uint_t width: the reference fixes `uint_t` at compile time (-DM64, gsa/gsacak.h:44-58).  Here it
 * is a per-context flag: arrays documented as "U-wide" hold uint32_t without PFP_FLAG_U64 and
 * uint64_t with it.
 *
 * Threading: one pfp_ctx per host thread (the reference gives each std::thread its own PfParser,
 * src/merge_pfp.cpp:97-104).  A context owns one HIP stream and one device workspace.
 */
#ifndef PFBWT_HIP_H
#define PFBWT_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pfp_ctx pfp_ctx;

typedef enum pfp_status {
    PFP_OK = 0,
    PFP_E_ARG = -1,          /* bad argument (w > 32: pfparser.hpp:371-376; p == 0; NULL) */
    PFP_E_INVALID_CHAR = -2, /* hash.hpp:31 "error, invalid character"; see pfp_error_detail */
    PFP_E_TOO_LARGE = -3,    /* input needs 64-bit device indices (pfparser.hpp:326-331, 393-404) */
    PFP_E_NOMEM = -4,        /* device workspace exhausted; pfp_workspace_needed() says how much */
    PFP_E_HIP = -5,          /* HIP runtime error; pfp_error_detail gives the hipError_t */
    PFP_E_ONE_WORD = -6,     /* pfparser.hpp:390-392 "only one dict word total" */
    PFP_E_STATE = -7,        /* call order violated (e.g. pfp_parse_bwt before pfp_parse_finalize) */
    PFP_E_CORRUPT = -8,      /* loaded parse files are inconsistent (pfbwt.hpp:139 "something went wrong!") */
    PFP_E_IO = -9            /* pfp_parse_feed_fasta_file: the file cannot be opened or read (pfparser.hpp:302-304 "failed to open file!") */
} pfp_status;

/* pfp_create flags */
#define PFP_FLAG_U64           1u /* uint_t = uint64_t (pfbwt-f64), else uint32_t (pfbwt-f) */
#define PFP_FLAG_NON_ACGT_TO_A 2u /* PfParserParams::non_acgt_to_a, pfparser.hpp:342-344 */
#define PFP_FLAG_SAI           4u /* PfParserParams::get_sai: keep sai/bwsai (needed for -s / -r) */

/* sizes reported after the parse; names as in SURVEY.md section 8 */
typedef struct pfp_parse_sizes {
    uint64_t n;      /* PfParser::get_n(): text length incl. the w 'A's after each sequence */
    uint64_t m;      /* get_parse_size(): phrases in the parse */
    uint64_t dwords; /* distinct phrases */
    uint64_t dsize;  /* bytes of the .dict image */
} pfp_parse_sizes;

typedef struct pfp_bwt_sizes {
    uint64_t nout;   /* n + 1 outputs */
    uint64_t r;      /* number of BWT runs (src/pfbwt-f.cpp:304-305) */
    uint64_t easy_cases, hard_cases; /* pfbwt.hpp:188 statistics (single-word / multi-word groups) */
} pfp_bwt_sizes;

/* ---- context --------------------------------------------------------------------------------- */
/* PfParser(PfParserParams) pfparser.hpp:82-84 + PrefixFreeBWT ctor pfbwt.hpp:64-81 (w only).
 * device = HIP device ordinal; workspace_bytes = device arena size, 0 = sized on demand. */
pfp_ctx *pfp_create(int w, uint64_t p, unsigned flags, int device, uint64_t workspace_bytes, int *status);
void pfp_destroy(pfp_ctx *ctx);
const char *pfp_strerror(int status);
/* for PFP_E_INVALID_CHAR: text position and byte; for PFP_E_HIP: *ch = hipError_t */
int pfp_error_detail(pfp_ctx *ctx, uint64_t *pos, int *ch);
/* bytes of device workspace the last PFP_E_NOMEM call would have needed (estimate) */
uint64_t pfp_workspace_needed(pfp_ctx *ctx);
/* drop the fed text and every result, keep the context and its workspace.  A stage that fails leaves the
 * context as it was before the call (workspace marks restored): after a failed pfp_parse_finalize the fed text
 * is still there -- retry, append more, or pfp_reset; after PFP_E_NOMEM from pfp_bwt_build(want_sa = 1) a retry
 * with want_sa = 0 or pfp_bwt_build_slice starts from the same state. */
int pfp_reset(pfp_ctx *ctx);

/* ---- stage 1: parse -------------------------------------------------------------------------- */
/* PfParser::add_fasta inner loop, pfparser.hpp:335-352: append raw sequence bytes (host memory).
 * end_of_seq != 0 closes the record: the w 'A's of :335-337 are appended.  Case folding, the
 * optional non-ACGT->A mapping and the validity check of hash.hpp:30-31 happen on the device. */
int pfp_parse_feed(pfp_ctx *ctx, const uint8_t *bases, uint64_t len, int end_of_seq);
/* `count` records of `len` bytes each, record k at bases + k*stride (host memory): the same as `count` calls of
 * pfp_parse_feed(.., len, 1).  Ingest (include/kseq.h:228 reads 16 KiB at a time): page-locked
 * (hipHostMalloc / hipHostRegister) memory is moved by one strided DMA transfer; pageable memory goes through the
 * context's two 32 MiB pinned staging buffers, the host copy into one overlapping the transfer of the other.
 * Both feed calls return when the caller's buffer may be reused. */
int pfp_parse_feed_batch(pfp_ctx *ctx, const uint8_t *bases, uint64_t count, uint64_t len, uint64_t stride);
/* same, but the bytes are already in device memory (one record, pad appended by the library) */
int pfp_parse_feed_device(pfp_ctx *ctx, const void *d_bases, uint64_t len, int end_of_seq);
/* `count` records of `len` bytes each, record k at d_bases + k*stride (device memory): the same as `count` calls of
 * pfp_parse_feed_device(.., len, 1), done as one strided copy (a collection of equal-length haplotypes) */
int pfp_parse_feed_device_batch(pfp_ctx *ctx, const void *d_bases, uint64_t count, uint64_t len, uint64_t stride);
/* The same rows WITHOUT the copy: the caller's buffer becomes the text of this parse -- it must be the first and only feed, and the
 * rows must stay valid and unchanged until pfp_parse_finalize returns.  The trigger scan of pfp_parse_finalize reads the rows where
 * they are and writes the normalised text (with the w 'A's behind every row) into the context's own buffer, which the later stages
 * use: one pass over the input instead of a copy pass and a scan pass (S-32G: 11 ms of 300).  Any other call that appends to or hands
 * out the text first materialises the rows like pfp_parse_feed_device_batch would have.  PFP_E_STATE: the context already holds text. */
int pfp_parse_feed_device_view(pfp_ctx *ctx, const void *d_bases, uint64_t count, uint64_t len, uint64_t stride);
/* FASTA ingest on the device: RAW file bytes -- header lines, newlines and all -- in any chunking (host memory;
 * page-locked memory is read by DMA in place).  Stands in for kseq_read as PfParser::add_fasta drives it, include/kseq.h:178-228,
 * include/pfparser.hpp:300-337: bytes in front of the first '>' / '@' are skipped, a line that starts with '>' or '@' is a header
 * line, the other lines are concatenated without their line ends ('\r' dropped), every record -- empty ones too -- is followed
 * by the w 'A's.  flags: PFP_FASTA_FINAL closes the stream (the last record's pad; the next call starts a new file),
 * PFP_FASTA_RECORDS collects, for the records that START in this call, the offset of their header's first byte in `raw` and the
 * text position (get_n() coordinates) of their first base -- the (name, start) pairs of --print-docs, pfparser.hpp:321-325 -- to be
 * fetched with pfp_parse_fasta_records; *nrec (nullable) = their number.  The raw bytes cross PCIe in chunks of up to 256 MiB;
 * stripping chunk k overlaps the upload of chunk k + 1; returns when the caller's buffer may be reused.
 * PFP_E_ARG with pfp_error_detail ch == '+': a line starts with '+' (a FASTQ quality section, kseq.h:209-221) -- not handled on
 * the device, the context must be reset and the input read by a host-side reader (pfp_parse_feed per record). */
#define PFP_FASTA_FINAL   1u
#define PFP_FASTA_RECORDS 2u
int pfp_parse_feed_fasta(pfp_ctx *ctx, const uint8_t *raw, uint64_t len, unsigned flags, uint64_t *nrec);
int pfp_parse_fasta_records(pfp_ctx *ctx, uint64_t *raw_off, uint64_t *text_pos);
/* The reading side of PfParser::add_fasta, pfparser.hpp:300-307 (gzopen + kseq_init + the kseq_read loop, include/kseq.h:178-228):
 * `path` ("-" = stdin; gzip or plain, detected by content) is read in 64 MiB blocks into page-locked buffers -- a plain regular
 * file by several threads at independent offsets, a gzip stream or a pipe by one -- and the blocks go through
 * pfp_parse_feed_fasta while the next ones are being read.  FASTQ (the first record starts with '@') is read record by record
 * on the host with kseq's rules (quality lines skipped).  flags: PFP_FASTA_RECORDS collects (name, start) of every record for
 * pfp_parse_docs / pfp_parse_doc_get (--print-docs); the stream is always closed (PFP_FASTA_FINAL implied).  info (nullable):
 * raw bytes read, records, text length behind the file, time the consumer waited for the readers, wall time, and which
 * reader ran (1 parallel pread, 2 single zlib / pipe reader, 3 host record reader). */
typedef struct pfp_ingest_info { uint64_t raw_bytes, records, n; double read_wait_ms, total_ms; int mode; } pfp_ingest_info;
int pfp_parse_feed_fasta_file(pfp_ctx *ctx, const char *path, unsigned flags, pfp_ingest_info *info);
int pfp_parse_docs(pfp_ctx *ctx, uint64_t *count);
int pfp_parse_doc_get(pfp_ctx *ctx, uint64_t i, const char **name, uint64_t *start);   /* valid until the next pfp_parse_feed_fasta_file */
/* Announce the size of the text that is going to be fed (e.g. the size of the FASTA file): sizes the ADDRESS range of the text
 * buffer -- HBM itself is committed as the text grows.  Optional; without it the range is four times what the first feed needs
 * (at least 64 MiB) and doubles when the text outgrows it (the text is then moved once: a device-to-device copy). */
int pfp_parse_reserve(pfp_ctx *ctx, uint64_t text_bytes);
/* Back to feeding with the text kept: after pfp_parse_finalize the (normalised) text is still on the device, more
 * records can be appended and pfp_parse_finalize run again -- what PfParser::operator+= (pfparser.hpp:194-263) needs
 * when the right-hand parse is appended as text (pfp_text_view of its context + pfp_parse_feed_device) instead of being
 * merged phrase by phrase.  PFP_E_STATE for a context whose state came from pfp_merge_shards or pfp_bwt_load. */
int pfp_parse_reopen(pfp_ctx *ctx);
/* device pointer and length of the text fed so far (NULL / 0 when the context holds none) */
int pfp_text_view(pfp_ctx *ctx, const uint8_t **d_text, uint64_t *n);
/* PfParser::finalize pfparser.hpp:484-517 (+ process_phrase :595-601 for every phrase): trigger scan,
 * phrase de-duplication, dictionary sort, ranks, occ, last, sai.  Results stay on the device. */
int pfp_parse_finalize(pfp_ctx *ctx, pfp_parse_sizes *out);
/* The same for a shard that is only going to be merged (pfp_shard_view_get + pfp_merge_shards): phrases, dictionary words
 * and phrase ids, but no dictionary sort, ranks, .parse or .occ -- the merge produces those for the united dictionary.  The
 * context then answers pfp_shard_view_get only (pfp_parse_get / pfp_parse_bwt: PFP_E_STATE), like one filled by
 * pfp_shard_load. */
int pfp_parse_finalize_shard(pfp_ctx *ctx, pfp_parse_sizes *out);
/* save_parser pfbwt_io.hpp:234-249 getters: copy results to caller-owned host buffers (NULL skips).
 * dict: dsize bytes (.dict image); occ: dwords U-wide; parse: m uint32 (1-based ranks);
 * last: m bytes; sai: m U-wide (only with PFP_FLAG_SAI). */
int pfp_parse_get(pfp_ctx *ctx, uint8_t *dict, void *occ, uint32_t *parse, uint8_t *last, void *sai);
/* PfParser::bwt_of_parse pfparser.hpp:379-467 (sacak_int :425 included) */
int pfp_parse_bwt(pfp_ctx *ctx);
/* the three vectors handed to OutFn at pfparser.hpp:466: bwlast m+1 bytes, ilist / bwsai m+1 U-wide */
int pfp_parse_bwt_get(pfp_ctx *ctx, uint8_t *bwlast, void *ilist, void *bwsai);

/* ---- multi-GPU: sharded parse (semantics of PfParser::operator+=, pfparser.hpp:194-263) ---- */
/* A shard is a run of whole sequences.  Shard r > 0 is fed the w 'A's that end shard r-1 first (pfp_parse_feed(ctx,
 * "AAAA...", w, 0) or pfp_parse_feed_left_context) as left context, then its sequences; every shard is parsed with pfp_parse_finalize on its own GPU.
 * pfp_shard_view_get exposes the device arrays of a finished local parse (to be copied into send buffers with
 * pfp_device_copy and exchanged with one RCCL all-gather); pfp_merge_shards takes the N views -- device pointers on
 * the calling context's GPU -- and leaves that context in the state pfp_parse_finalize would have produced on the
 * concatenated text (then pfp_parse_get / pfp_parse_bwt / pfp_bwt_build as usual). */
typedef struct pfp_shard_view {
    uint64_t n, m, dwords, dsize;   /* n counts the w context bytes of a shard r > 0 */
    const uint8_t *d_dict;          /* dsize bytes: distinct phrases, each + 0x01, then 0x00 (any order) */
    const uint32_t *d_ws;           /* dwords+1 word starts in d_dict */
    const uint32_t *d_pid;          /* m: word id of every phrase */
    const uint64_t *d_ye;           /* m: 1-based end position of every phrase in the shard's text (= sai) */
    const uint8_t *d_last;          /* m.  d_ye and d_last may BOTH be NULL: pfp_merge_shards then derives them from d_dict / d_ws / d_pid
                                       (a phrase ends where its predecessor ended + its word's length - w), so that only the dictionary
                                       and 4 bytes per phrase have to travel between GPUs */
    uint64_t left_context;          /* bytes of left context in front of the shard's own text: w (pfp_parse_feed_left_context: the
                                       shard was parsed knowing that w 'A's precede it) or 0 (a stand-alone parse, e.g. one made by
                                       `pfbwt-f --parse-only` and loaded with pfp_shard_load: the merge then re-tests the first w
                                       windows of the shard like PfParser::operator+=, pfparser.hpp:226-245) */
} pfp_shard_view;
/* the w 'A's that end the previous shard, fed as left context of a shard r > 0 (must be the first feed of the shard) */
int pfp_parse_feed_left_context(pfp_ctx *ctx);
/* A parse that was saved to <prefix>.dict / <prefix>.parse (pfbwt_io.hpp:234-249; load_parser :211-222 + init_from_dict_ranks,
 * pfparser.hpp:549-567) becomes a shard on the device: dict = the .dict image (dsize bytes), parse = m 1-based ranks.
 * The context then answers pfp_shard_view_get (left_context = 0) -- nothing else; it holds no text. */
int pfp_shard_load(pfp_ctx *ctx, const uint8_t *dict, uint64_t dsize, const uint32_t *parse, uint64_t m);
int pfp_shard_view_get(pfp_ctx *ctx, pfp_shard_view *view);
int pfp_device_copy(pfp_ctx *ctx, void *d_dst, const void *d_src, uint64_t bytes);
int pfp_merge_shards(pfp_ctx *ctx, int nshards, const pfp_shard_view *views, pfp_parse_sizes *out);

/* ---- multi-GPU from ONE process: N devices, N host threads, RCCL called directly ------------------------------------------------
 * The reference's only parallelism has this shape: src/merge_pfp.cpp:131-152 gives every std::thread its own PfParser over a
 * contiguous slice of the inputs and folds the per-thread parsers with PfParser::operator+= (include/pfparser.hpp:194-263).  Here
 * rank r owns device devices[r] (NULL: 0 .. ndev-1): the caller feeds rank r's run of whole sequences into pfp_sharded_ctx(s, r) with
 * any pfp_parse_feed* call (the w 'A's that end the previous run are already in front of it, pfparser.hpp:335-337; ranks may be fed
 * from different threads), then ONE call builds everything: every rank finalizes its shard (pfp_parse_finalize_shard), the ranks
 * exchange {dictionary, word starts, phrase ids} in one ncclAllGather over xGMI (ncclCommInitAll at creation; librccl.so is loaded
 * at run time, only when distinct devices have to talk -- ranks that share a device, a rehearsal of the protocol on one card,
 * copy device to device), every rank merges (pfp_merge_shards), sorts dictionary and parse, and emits slice r of the output rows
 * (pfp_bwt_build_slice).  Afterwards pfp_bwt_get / pfp_bwt_device_ptrs / pfp_bwt_write of rank r's context deliver slice r; the
 * slices in rank order are the reference's .bwt / .sa / .ssa / .esa.  psz: sizes of the whole collection's parse; bsz / slice_begin /
 * slice_rows / esa_pairs: ndev entries each (NULL skips).  A failing rank (an invalid character in its shard, ...) makes every rank
 * give up before the collective; pfp_sharded_error names it (every rank needs at least one sequence: PFP_E_ARG otherwise).
 * pfp_sharded_reset: all ranks ready for the next collection. */
typedef struct pfp_sharded pfp_sharded;
pfp_sharded *pfp_sharded_create(int w, uint64_t p, unsigned flags, int ndev, const int *devices, uint64_t workspace_bytes, int *status);
void pfp_sharded_destroy(pfp_sharded *s);
int pfp_sharded_ranks(pfp_sharded *s);
pfp_ctx *pfp_sharded_ctx(pfp_sharded *s, int rank);
int pfp_sharded_build(pfp_sharded *s, int want_sa, int want_rssa, pfp_parse_sizes *psz, pfp_bwt_sizes *bsz,
                      uint64_t *slice_begin, uint64_t *slice_rows, uint64_t *esa_pairs);
int pfp_sharded_reset(pfp_sharded *s);
const char *pfp_sharded_error(pfp_sharded *s);

/* ---- stage 2: BWT / SA ------------------------------------------------------------------------- */
/* PrefixFreeBWT ctor pfbwt.hpp:64-81, for --pfbwt-only: upload .dict .occ .bwlast .ilist [.bwsai]
 * images (host memory).  Not needed when pfp_parse_finalize + pfp_parse_bwt ran in this context.
 * ilist and bwsai must hold nrows elements each, like bwlast (the caller compares the file sizes).
 * n_hint: the value of the .n file (src/pfbwt-f.cpp:282-285), sizes the workspace and arms the
 * "exactly n + 1 rows" check of the emission; 0 = unknown.  PFP_E_CORRUPT: the images are inconsistent
 * (dictionary not terminated, word count != entries of occ, sum(occ) + 1 != nrows, an ilist entry >= nrows). */
int pfp_bwt_load(pfp_ctx *ctx, const uint8_t *dict, uint64_t dsize, const void *occ, uint64_t dwords,
                 const uint8_t *bwlast, const void *ilist, const void *bwsai, uint64_t nrows, uint64_t n_hint);
/* PrefixFreeBWT::generate_bwt_lcp pfbwt.hpp:96-194 (sort_dict_suffixes :206-223 = gsacak included)
 * fused with the CLI's out_fn src/pfbwt-f.cpp:298-328: BWT bytes, SA (row 0 := n), run samples. */
int pfp_bwt_build(pfp_ctx *ctx, int want_sa, int want_rssa, pfp_bwt_sizes *out);
/* The same, with the rows on their way to host memory while the emission is still running: host_bwt (n + 1 bytes, n from
 * pfp_parse_sizes / pfp_text_length) and, with want_sa, host_sa (n + 1 U-wide values) are filled window by window -- the DMA
 * transfer of a window overlaps the emission of the next (page-locked destinations run at the link's rate).  The run samples
 * are fetched afterwards with pfp_bwt_get(ctx, NULL, NULL, ssa, esa) once out->r says how large they are. */
int pfp_bwt_build_stream(pfp_ctx *ctx, int want_sa, int want_rssa, uint8_t *host_bwt, void *host_sa, pfp_bwt_sizes *out);
/* PfParser::get_n(): bytes of text fed so far (the w 'A's behind every record included) */
int pfp_text_length(pfp_ctx *ctx, uint64_t *n);
/* Multi-GPU emission: every rank holds the same parse state (after pfp_merge_shards + pfp_parse_bwt) and emits
 * only output rows [nout*slice/nslices, nout*(slice+1)/nslices).  out->r counts the runs that START in the slice
 * (the sum over slices is r); pfp_bwt_get / pfp_bwt_device_ptrs then refer to the slice (slice_rows entries).
 * want_rssa: the slice's part of the run samples (src/pfbwt-f.cpp:306-315, 325-328) -- out->r (row, sa) pairs for the
 * run starts in the slice and *esa_pairs pairs for the run ends they imply (the row in front of every run start, which
 * for the first start of a slice > 0 lies in the previous slice, plus the last row of the output in the last slice).
 * Concatenated over the slices in order they are the reference's .ssa / .esa files; no rank needs another rank's data. */
int pfp_bwt_build_slice(pfp_ctx *ctx, int want_sa, int want_rssa, int slice, int nslices, pfp_bwt_sizes *out,
                        uint64_t *slice_begin, uint64_t *slice_rows, uint64_t *esa_pairs);
/* copy results to host (NULL skips): bwt nout bytes; sa nout U-wide; ssa 2*r, esa 2*r (slices: 2*esa_pairs) U-wide */
int pfp_bwt_get(pfp_ctx *ctx, uint8_t *bwt, void *sa, void *ssa, void *esa);
/* `.bwt` into host memory through its run-length form (needs the state left by pfp_bwt_build(want_rssa = 1) over the whole output):
 * row `.ssa[k]` starts run k, so one byte per run crosses PCIe (r bytes instead of n + 1 -- 84 MB instead of 32 GB on a 1000-haplotype
 * collection) and `threads` host threads write the runs out.  ssa_host: the 2 * r U-wide values pfp_bwt_get returned for `.ssa`
 * (NULL: fetched again).  host_bwt receives exactly the n + 1 bytes of pfp_bwt_get.  threads < 1: one per CPU the process may run on;
 * the output is cut into 64 MiB blocks of BYTES that the threads claim from the front (runs are cut at the blocks' borders); when
 * host_bwt is page-locked (pfp_host_register) the copy engine claims blocks from the back and moves those rows over PCIe while the threads write. */
int pfp_bwt_get_expanded(pfp_ctx *ctx, uint8_t *host_bwt, const void *ssa_host, int threads);
/* The results of the last build straight to file descriptors (-1 skips one): what out_fn of src/pfbwt-f.cpp:298-328 writes with two to
 * four fwrite calls per base.  The bytes leave the device in 64 MiB blocks through page-locked buffers; the transfer of a block
 * overlaps the write of the one before; a regular file is written by several threads (pwrite at the block's offset), a pipe --
 * `-c bwt`: stdout -- in order by one.  PFP_E_IO: a write failed. */
int pfp_bwt_write(pfp_ctx *ctx, int fd_bwt, int fd_sa, int fd_ssa, int fd_esa);
/* device pointers of the same results (valid until the next pfp_* call that rebuilds them) */
int pfp_bwt_device_ptrs(pfp_ctx *ctx, const void **d_bwt, const void **d_sa, const void **d_ssa, const void **d_esa);

/* ---- marker-array post-pass --------------------------------------------------- */
/* write_marker_array, include/marker_array.hpp:138-174 (the tool src/mps_to_ma.cpp): mps = the marker-positions stream
 * written by MarkerPositionsWriter (:60-136; records: first text position, last text position, packed markers
 * (include/marker.hpp:9-52), 0xFFFFFFFFFFFFFFFF; intervals ascending and disjoint).  Every suffix-array value is looked up
 * (rle_window_arr::at, include/rle_window_array.hpp:118-131) and consecutive rows with equal, non-empty marker lists become
 * one record: first row, last row, the markers, 0xFFFFFFFFFFFFFFFF.  sa_host == NULL: fused with the build -- the suffix
 * array pfp_bwt_build(want_sa = 1) left on the device is used (the reference pipes it through `tee`, vcf_to_bwt.py:259-285);
 * otherwise sa_host holds nrows U-wide values in BWT order (row 0 = n, src/pfbwt-f.cpp:301) and the context is reset.
 * *out_words = 64-bit words of the .ma stream, fetched with pfp_marker_array_get. */
int pfp_marker_array(pfp_ctx *ctx, const uint64_t *mps, uint64_t mps_words, const void *sa_host, uint64_t nrows, uint64_t *out_words);
int pfp_marker_array_get(pfp_ctx *ctx, uint64_t *dst);

/* ---- drop-ins for the suffix-sorting C ABI, gsa/gsacak.h:76-103 ------------------------------- */
/* int sacak_int(int_text *s, uint_t *SA, uint_t n, uint_t k): s[n-1]==0, symbols < k.  Returns the
 * number of refinement rounds (>= 1; the reference returns its recursion depth) or -1 on error. */
int pfp_sacak_int_u32(const uint32_t *s, uint32_t *SA, uint32_t n, uint32_t k);
int pfp_sacak_int_u64(const uint32_t *s, uint64_t *SA, uint64_t n, uint64_t k);

/* int gsacak(unsigned char *s, uint_t *SA, int_t *LCP, int_t *DA, uint_t n), gsa/gsacak.h:86-96 -- the call made by
 * PrefixFreeBWT::sort_dict_suffixes, include/pfbwt.hpp:211.  s: strings over ANY byte alphabet (the parser's dictionaries use '-', A,
 * C, G, N, T and Dollar = 2 and take the tuned path), each followed by the separator 1, s[n-1] == 0 and no other 0.  SA: suffixes
 * compared up to their separator, byte-identical ones in position order (gsacak.c:877-912); LCP (nullable): stops at the
 * separator (:64), LCP[0] = 0; DA (nullable): index of the string a suffix starts in.  Returns the number of refinement rounds
 * (>= 1; the reference returns its recursion depth) or -1: NULL s / SA as the reference; also n >= 2^32 - 64 in either width (device
 * suffix indices are 32-bit: dictionaries below 4 GiB) and a 0 byte that is not the last one.  The engine itself never
 * materialises LCP (pfp_bwt_build uses class heads instead). */
int pfp_gsacak_u32(const uint8_t *s, uint32_t *SA, int32_t *LCP, int32_t *DA, uint32_t n);
int pfp_gsacak_u64(const uint8_t *s, uint64_t *SA, int64_t *LCP, int64_t *DA, uint64_t n);

/* Page-lock / release caller-owned host memory (hipHostRegister): sources of pfp_parse_feed_fasta / pfp_parse_feed_batch and
 * destinations of pfp_bwt_build_stream / pfp_bwt_get then move at the link's rate, and a caller of this C ABI need not link
 * the HIP runtime itself. */
int pfp_host_register(void *p, uint64_t bytes);
int pfp_host_unregister(void *p);

/* library build info: "hip-gfx950" for the product library */
const char *pfp_backend(void);

#ifdef __cplusplus
}
#endif
#endif
