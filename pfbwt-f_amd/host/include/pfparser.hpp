// pfbwt-f_amd/host/include/pfparser.hpp -- pfbwtf::PfParser<Hasher> with the callable surface of the
// reference's include/pfparser.hpp:72-617, implemented on top of the MI355X engine (include/pfbwt_hip.h).
//
//   reference member                         here
//   PfParser(PfParserParams) :82-84          pfp_create (w, p, flags)
//   add_fasta :299-369                       pfp_parse_feed_fasta_file: the engine reads the file in blocks on helper threads
//                                            into page-locked memory and strips headers / line ends on the device.  The
//                                            text lives in HBM ONLY: the mirror keeps no copy (a 32 Gbase collection
//                                            would not fit a host string)
//   finalize :484-517                        pfp_parse_finalize + pfp_parse_get (dict, occ, ranks, last, sai)
//   bwt_of_parse :379-467                    pfp_parse_bwt + pfp_parse_bwt_get, then OutFn(bwlast, ilist, bwsai)
//   operator+= :194-263                      exact by construction: the merged parse IS the parse of the
//                                            concatenated texts (tests/test_parser.cpp:188-234), so += copies rhs' text
//                                            device to device behind its own (pfp_parse_reopen + pfp_text_view +
//                                            pfp_parse_feed_device) and finalize() re-parses on the GPU
//   load ctor :89-132, init_from_dict_ranks  text rebuilt from (dict, ranks) in a transient buffer, fed, parsed
//   getters :469-544                         host vectors filled by finalize()
// Not carried over: the std::map FreqMap (get_freqs) -- use get_sorted_phrases()/get_occs().
#ifndef PFBWTF_PFPARSER_HPP
#define PFBWTF_PFPARSER_HPP
#include <string>
#include <vector>
#include "hash.hpp"
#include "pfbwtf_common.hpp"

namespace pfbwtf {

struct PfParserParams {
    constexpr PfParserParams() {}
    constexpr PfParserParams(size_t wsize, size_t pmod, bool s, bool d, bool v, bool trim, bool ntoa)
        : w(wsize), p(pmod), get_sai(s), store_docs(d), verbose(v), trim_non_acgt(trim), non_acgt_to_a(ntoa) {}
    size_t w = 10;
    size_t p = 100;
    bool get_sai = false;
    bool store_docs = false;
    bool verbose = false;
    bool trim_non_acgt = false;   // disabled code in the reference too (pfparser.hpp:338-341)
    bool non_acgt_to_a = false;
};

// run of non-ACGT characters (pfparser.hpp:61-67); only filled by the reference's disabled --trim-non-acgt code
// (:338-341), i.e. always empty -- kept so that run_parser (src/pfbwt-f.cpp:236-240) compiles unchanged
struct ntab_entry {
    size_t pos = 0;
    size_t l = 0;
    void clear() { pos = 0; l = 0; }
};

template <typename Hasher = WangHash> struct PfParser {
  public:
    using UIntType = uint_t;
    using IntType = int_text;

    PfParser() {}
    explicit PfParser(PfParserParams p) : params_(p) { check_w(p.w); }
    // load from a sorted dictionary and the parse ranks (pfparser.hpp:89-132)
    PfParser(PfParserParams params, const std::vector<std::string> &sorted_phrases, const std::vector<IntType> &parse_ranks,
             const std::vector<UIntType> &doc_starts = std::vector<UIntType>(), const std::vector<std::string> &doc_names = std::vector<std::string>())
        : params_(params), doc_starts_(doc_starts), doc_names_(doc_names)
    {
        check_w(params.w);
        text_from_dict_ranks(sorted_phrases, parse_ranks);
        finalize();
    }
    PfParser(const PfParser &rhs) { copy_from(rhs); }
    PfParser &operator=(const PfParser &rhs) { if (this != &rhs) { release(); copy_from(rhs); } return *this; }
    PfParser(PfParser &&rhs) noexcept { move_from(rhs); }
    PfParser &operator=(PfParser &&rhs) noexcept { if (this != &rhs) { release(); move_from(rhs); } return *this; }
    ~PfParser() { release(); }

    // append another parse (pfparser.hpp:194-263); call finalize() afterwards
    PfParser &operator+=(const PfParser &rhs)
    {
        if (!n_fed_ && !nseqs_) { const PfParserParams keep = params_; const bool had = have_params_; *this = rhs; if (had) { params_.store_docs = keep.store_docs; } finalized_ = false; return *this; }
        if (rhs.params_.w != params_.w) { fprintf(stderr, "invalid w\n"); exit(1); }
        if (rhs.params_.p != params_.p) { fprintf(stderr, "invalid p\n"); exit(1); }
        const size_t prev_n = n_fed_;
        for (auto s : rhs.doc_starts_) doc_starts_.push_back((UIntType)(s + prev_n));
        for (auto &nm : rhs.doc_names_) doc_names_.push_back(nm);
        append_device_text(rhs);
        nseqs_ += rhs.nseqs_;
        finalized_ = false;
        return *this;
    }
    bool operator==(const PfParser &rhs) const
    {
        return get_n() == rhs.get_n() && parse_ranks_ == rhs.parse_ranks_ && last_ == rhs.last_ && dict_ == rhs.dict_ && occs_ == rhs.occs_ &&
               (!params_.get_sai || sai_ == rhs.sai_);
    }

    // stores parse information from a fasta file (pfparser.hpp:299-369); returns pos_
    size_t add_fasta(std::string fasta_fname)
    {
        ensure_ctx();
        if (finalized_) { engine_check(ctx_, pfp_parse_reopen(ctx_), "pfp_parse_reopen"); finalized_ = false; }
        // the whole reading side lives in the engine (csrc/ingest.h): blocks of raw bytes are read on helper threads into
        // page-locked memory, cross PCIe while the next ones are read, and are stripped of headers / line ends on the device;
        // the w 'A's of :335-337 are appended there
        pfp_ingest_info info;
        const int st = pfp_parse_feed_fasta_file(ctx_, fasta_fname.c_str(), params_.store_docs ? PFP_FASTA_RECORDS : 0u, &info);
        if (st == PFP_E_IO) die("failed to open file!\n");
        engine_check(ctx_, st, "pfp_parse_feed_fasta_file");
        if (params_.store_docs) {
            uint64_t nd = 0; engine_check(ctx_, pfp_parse_docs(ctx_, &nd), "pfp_parse_docs");
            for (uint64_t i = 0; i < nd; ++i) {
                const char *nm = nullptr; uint64_t start = 0;
                engine_check(ctx_, pfp_parse_doc_get(ctx_, i, &nm, &start), "pfp_parse_doc_get");
                doc_starts_.push_back((UIntType)start); doc_names_.push_back(nm);
            }
        }
        if (params_.verbose) fprintf(stderr, "read %lu bytes, %lu records in %.3f s (reader %d, waited %.3f s for it)\n", (unsigned long)info.raw_bytes, (unsigned long)info.records,
                                     info.total_ms * 1e-3, info.mode, info.read_wait_ms * 1e-3);
        n_fed_ = info.n;
        nseqs_ += info.records;
        return get_pos();
    }

    void check_w(size_t x) { have_params_ = true; if (x > 32) { fprintf(stderr, "window size w must be < 32!\n"); exit(1); } }

    // sort dictionary, generate ranks (pfparser.hpp:484-517) -- here: the whole GPU parse
    void finalize()
    {
        if (finalized_) return;
        ensure_ctx();
        pfp_parse_sizes sz;
        engine_check(ctx_, pfp_parse_finalize(ctx_, &sz), "pfp_parse_finalize");
        fetch_results(sz);
    }
    // Exact merge of parses on the device (PfParser::operator+= over N operands, pfparser.hpp:194-263; what src/merge_pfp.cpp:97-113
    // folds): the shards' dictionaries are united and de-duplicated, every seam is re-hashed, the result is the parse of the
    // concatenated texts -- no text is rebuilt or re-parsed.  `views` come from pfp_shard_view_get of contexts that hold a parse
    // (pfp_parse_finalize) or a loaded one (pfp_shard_load).  The merged parser holds no text: it cannot be appended to.
    void merge_device_shards(const std::vector<pfp_shard_view> &views)
    {
        ensure_ctx();
        pfp_parse_sizes sz;
        engine_check(ctx_, pfp_merge_shards(ctx_, (int)views.size(), views.data(), &sz), "pfp_merge_shards");
        n_fed_ = sz.n; nseqs_ = views.size();
        fetch_results(sz);
    }
    void set_docs(const std::vector<UIntType> &starts, const std::vector<std::string> &names) { doc_starts_ = starts; doc_names_ = names; }
    void sort_dict() { finalize(); }
    void generate_ranks() { finalize(); }
    void regenerate_parse() { finalize(); }

    // generates bwlast and ilist (and bwsai) (pfparser.hpp:379-467)
    template <typename OutFn> void bwt_of_parse(OutFn out_fn)
    {
        finalize();
        if (parse_ranks_.size() == 1) die("error: only one dict word total. Re-run with a smaller p modulus");
        int st = pfp_parse_bwt(ctx_);
        if (st == PFP_E_ONE_WORD) die("error: only one dict word total. Re-run with a smaller p modulus");
        engine_check(ctx_, st, "pfp_parse_bwt");
        const size_t nr = parse_ranks_.size() + 1;
        std::vector<char> bwlast(nr); std::vector<UIntType> ilist(nr), bwsai(params_.get_sai ? nr : 0);
        engine_check(ctx_, pfp_parse_bwt_get(ctx_, (uint8_t *)bwlast.data(), ilist.data(), params_.get_sai ? bwsai.data() : nullptr), "pfp_parse_bwt_get");
        parse_bwt_done_ = true;
        out_fn(bwlast, ilist, bwsai);
    }

    size_t get_parse_size() const { return parse_ranks_.size(); }
    const std::vector<UIntType> get_occs() const { return occs_; }
    size_t get_n() const { return finalized_ ? n_ : n_fed_; }                  // includes the As at the end of each seq
    const std::vector<UIntType> &get_sai() const { return sai_; }
    const std::vector<char> &get_last() const { return last_; }
    const std::vector<int_text> &get_parse_ranks() const { return parse_ranks_; }
    const std::vector<const char *> &get_sorted_phrases() const { return sorted_phrases_; }
    const std::vector<char> &get_dict_image() const { return dict_; }          // the .dict bytes (pfbwt_io.hpp:71-82)
    const std::vector<ntab_entry> &get_ntab() const { return ntab_; }
    const std::vector<UIntType> &get_doc_starts() const { return doc_starts_; }
    const std::vector<std::string> &get_doc_names() const { return doc_names_; }
    const PfParserParams get_params() const { return params_; }
    size_t get_pos() const { return n_fed_ + (finalized_ ? params_.w : 1); }   // pos_ counts the Dollars (:612)
    // the engine context that holds this parse on the device (lets PrefixFreeBWT skip the file round trip)
    pfp_ctx *engine() const { return ctx_; }
    bool parse_bwt_done() const { return parse_bwt_done_; }

  private:
    void fetch_results(const pfp_parse_sizes &sz)
    {
        n_ = sz.n; dict_.resize(sz.dsize); occs_.resize(sz.dwords); parse_ranks_.resize(sz.m); last_.resize(sz.m);
        if (params_.get_sai) sai_.resize(sz.m); else sai_.clear();
        engine_check(ctx_, pfp_parse_get(ctx_, (uint8_t *)dict_.data(), occs_.data(), parse_ranks_.data(), (uint8_t *)last_.data(), params_.get_sai ? sai_.data() : nullptr), "pfp_parse_get");
        // NUL-terminated keys for get_sorted_phrases(): private copy of the dict image with EndOfWord -> 0
        keys_ = dict_; sorted_phrases_.clear(); sorted_phrases_.reserve(sz.dwords);
        size_t start = 0;
        for (size_t i = 0; i + 1 < keys_.size(); ++i) if (keys_[i] == EndOfWord) { keys_[i] = 0; sorted_phrases_.push_back(keys_.data() + start); start = i + 1; }
        finalized_ = true; parse_bwt_done_ = false;
    }
    void ensure_ctx()
    {
        if (ctx_) return;
        unsigned flags = (M64 ? PFP_FLAG_U64 : 0u) | (params_.non_acgt_to_a ? PFP_FLAG_NON_ACGT_TO_A : 0u) | PFP_FLAG_SAI;
        int st = 0, dev = 0;
        if (const char *e = getenv("PFBWT_DEVICE")) dev = atoi(e);
        ctx_ = pfp_create((int)params_.w, params_.p, flags, dev, 0, &st);
        if (!ctx_) { fprintf(stderr, "pfp_create: %s\n", pfp_strerror(st)); exit(1); }
    }
    void release() { if (ctx_) { pfp_destroy(ctx_); ctx_ = nullptr; } }
    // the text of `r` (device resident in r's context) behind this parser's own text, device to device
    void append_device_text(const PfParser &r)
    {
        if (!r.n_fed_) return;
        ensure_ctx();
        if (finalized_ || n_fed_) engine_check(ctx_, pfp_parse_reopen(ctx_), "pfp_parse_reopen");
        const uint8_t *dt = nullptr; uint64_t dn = 0;
        engine_check(r.ctx_, pfp_text_view(r.ctx_, &dt, &dn), "pfp_text_view");
        if (!dt || dn != r.n_fed_) { fprintf(stderr, "PfParser: the right-hand parse holds no text on the device\n"); exit(1); }
        engine_check(ctx_, pfp_parse_feed_device(ctx_, dt, dn, 0), "pfp_parse_feed_device");   // its pads are part of its text
        n_fed_ += dn;
    }
    void copy_from(const PfParser &r)
    {
        params_ = r.params_; have_params_ = r.have_params_; doc_starts_ = r.doc_starts_; doc_names_ = r.doc_names_; nseqs_ = 0; n_fed_ = 0;
        n_ = r.n_; dict_ = r.dict_; occs_ = r.occs_; parse_ranks_ = r.parse_ranks_; last_ = r.last_; sai_ = r.sai_; keys_ = r.keys_;
        finalized_ = false; parse_bwt_done_ = false; ctx_ = nullptr;      // device state is not shared: the copy gets its own text, the next finalize() parses it (cheap on the GPU)
        sorted_phrases_.clear();
        for (auto p : r.sorted_phrases_) sorted_phrases_.push_back(keys_.data() + (p - r.keys_.data()));
        append_device_text(r);
        nseqs_ = r.nseqs_;
    }
    void move_from(PfParser &r)
    {
        params_ = r.params_; have_params_ = r.have_params_; n_fed_ = r.n_fed_; doc_starts_ = std::move(r.doc_starts_);
        doc_names_ = std::move(r.doc_names_); nseqs_ = r.nseqs_; n_ = r.n_; dict_ = std::move(r.dict_); occs_ = std::move(r.occs_);
        parse_ranks_ = std::move(r.parse_ranks_); last_ = std::move(r.last_); sai_ = std::move(r.sai_); keys_ = std::move(r.keys_);
        sorted_phrases_ = std::move(r.sorted_phrases_); finalized_ = r.finalized_; parse_bwt_done_ = r.parse_bwt_done_; ctx_ = r.ctx_; r.ctx_ = nullptr; r.n_fed_ = 0; r.nseqs_ = 0;
    }
    // inverse of the parse: phrase 0 without its Dollar, every later phrase without its first w bytes,
    // the last one without its w Dollars (init_from_dict_ranks, pfparser.hpp:549-567); fed in pieces, never held whole
    void text_from_dict_ranks(const std::vector<std::string> &phrases, const std::vector<IntType> &ranks)
    {
        ensure_ctx();
        std::string buf;
        for (size_t j = 0; j < ranks.size(); ++j) {
            const std::string &ph = phrases[ranks[j] - 1];
            size_t from = j ? params_.w : 1, to = ph.size();
            if (j + 1 == ranks.size()) to -= params_.w;
            if (to > from) buf.append(ph, from, to - from);
            if (buf.size() >= ((size_t)64 << 20) || j + 1 == ranks.size()) {
                engine_check(ctx_, pfp_parse_feed(ctx_, (const uint8_t *)buf.data(), buf.size(), 0), "pfp_parse_feed");
                n_fed_ += buf.size(); buf.clear();
            }
        }
        nseqs_ = 1;
    }

    PfParserParams params_;
    bool have_params_ = false;
    size_t n_fed_ = 0;                        // bytes of text (bases + w 'A's per sequence) fed to the engine; the text itself is in HBM only
    std::vector<UIntType> doc_starts_;
    std::vector<std::string> doc_names_;
    std::vector<ntab_entry> ntab_;
    size_t nseqs_ = 0;
    // results
    size_t n_ = 0;
    std::vector<char> dict_, keys_, last_;
    std::vector<UIntType> occs_, sai_;
    std::vector<int_text> parse_ranks_;
    std::vector<const char *> sorted_phrases_;
    bool finalized_ = false, parse_bwt_done_ = false;
    pfp_ctx *ctx_ = nullptr;
};

} // namespace pfbwtf
#endif
