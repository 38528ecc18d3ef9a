// pfbwt-f_amd/host/include/pfbwt.hpp -- pfbwtf::PrefixFreeBWT<ReadConType, WriteConType> with the
// callable surface of the reference's include/pfbwt.hpp:54-287 on top of the MI355X engine.
//
//   ctor(PrefixFreeBWTParams) :64-81     reads prefix.{dict,bwlast,ilist,occ[,bwsai]} with the container
//                                        types and uploads them (pfp_bwt_load)
//   generate_bwt_lcp(out_fn) :96-194     pfp_bwt_build (dictionary suffix sort + emission on the GPU), then
//                                        replays out_fn(out_fn_arg) once per row, in order, on the caller's
//                                        thread -- exactly n+1 calls like the reference
// Differences a caller can observe: `dif` is always EASY1 (the engine reports the easy / hard totals, not a label per
// row); `sa` of row 0 already holds n (the CLI substitutes it anyway, pfbwt-f.cpp:301); with rssa && !sa the `sa` field
// is exact on run-boundary rows only (see generate_bwt_lcp).
#ifndef PFBWTF_PFBWT_HPP
#define PFBWTF_PFBWT_HPP
#include <string>
#include <vector>
#include "pfbwtf_common.hpp"

namespace pfbwtf {

enum class RunType { OTHER, START, END };
enum class Difficulty { EASY1, EASY2, HARD };

struct out_fn_arg {
    out_fn_arg(uint_t p, uint_t s, uint8_t pc, uint8_t c, Difficulty d = Difficulty::EASY1) : pos(p), sa(s), pbwtc(pc), bwtc(c), dif(d) {}
    uint_t pos;
    uint_t sa;
    uint8_t pbwtc;
    uint8_t bwtc;
    Difficulty dif;
};

struct PrefixFreeBWTParams {
    std::string prefix;
    size_t w;
    bool sa = false;
    bool rssa = false;
    bool verb = false;
};

template <template <typename, typename...> class ReadConType, template <typename, typename...> class WriteConType> class PrefixFreeBWT {
  public:
    using UIntType = uint_t;
    using IntType = int_t;

    explicit PrefixFreeBWT(PrefixFreeBWTParams args, uint64_t n_hint = 0)
        : fname(args.prefix), w(args.w), build_sa(args.sa), build_rssa(args.rssa), any_sa(args.sa | args.rssa), verbose(args.verb)
    {
        WriteConType<uint8_t> dict(args.prefix + "." + EXTDICT);
        ReadConType<uint8_t> bwlast(args.prefix + "." + EXTBWLST);
        ReadConType<UIntType> ilist(args.prefix + "." + EXTILIST);
        ReadConType<UIntType> occ(args.prefix + "." + EXTOCC);
        ReadConType<UIntType> bwsai;
        if (any_sa) bwsai = ReadConType<UIntType>(args.prefix + "." + EXTBWSAI);
        if (verbose) fprintf(stderr, "loaded files\n");
        dsize = dict.size(); dwords = occ.size();
        if (dsize < 1) die("error: dictionary not loaded\n");
        if (ilist.size() != bwlast.size() || (any_sa && bwsai.size() != bwlast.size())) {      // a truncated or mismatched file set
            fprintf(stderr, "error: %s.{bwlast,ilist,bwsai} do not hold the same number of rows (%lu, %lu, %lu)\n", args.prefix.c_str(),
                    (unsigned long)bwlast.size(), (unsigned long)ilist.size(), (unsigned long)bwsai.size());
            exit(1);
        }
        int st = 0, dev = 0;
        if (const char *e = getenv("PFBWT_DEVICE")) dev = atoi(e);
        ctx_ = pfp_create((int)w, 100, (M64 ? PFP_FLAG_U64 : 0u) | PFP_FLAG_SAI, dev, 0, &st);
        if (!ctx_) { fprintf(stderr, "pfp_create: %s\n", pfp_strerror(st)); exit(1); }
        engine_check(ctx_, pfp_bwt_load(ctx_, dict.data(), dict.size(), occ.data(), occ.size(), bwlast.data(), ilist.data(), any_sa ? bwsai.data() : nullptr, bwlast.size(), n_hint), "pfp_bwt_load");
        sizes_[0] = dict.size(); sizes_[1] = bwlast.size(); sizes_[2] = ilist.size(); sizes_[3] = bwsai.size();
    }
    // adopt the device-resident parse of a PfParser (skips the file round trip of src/pfbwt-f.cpp:224-235 -> :64-81)
    PrefixFreeBWT(pfp_ctx *parsed_ctx, PrefixFreeBWTParams args)
        : fname(args.prefix), w(args.w), build_sa(args.sa), build_rssa(args.rssa), any_sa(args.sa | args.rssa), verbose(args.verb), ctx_(parsed_ctx), owns_(false) {}
    PrefixFreeBWT(const PrefixFreeBWT &) = delete;
    PrefixFreeBWT &operator=(const PrefixFreeBWT &) = delete;
    ~PrefixFreeBWT() { if (ctx_ && owns_) pfp_destroy(ctx_); }

    // runs the build; afterwards bwt()/ssa()/esa() hold the whole output and sa() the full SA if it was asked for
    // (-s, or rows_need_sa: generate_bwt_lcp hands every row's SA value to out_fn even when only -r is set)
    void build(bool rows_need_sa = false)
    {
        const bool full_sa = build_sa || (rows_need_sa && any_sa);
        if (built_ && (!full_sa || !sa_.empty())) return;
        pfp_bwt_sizes bs;
        engine_check(ctx_, pfp_bwt_build(ctx_, full_sa ? 1 : 0, build_rssa ? 1 : 0, &bs), "pfp_bwt_build");
        nout_ = bs.nout; r_ = bs.r; easy_ = bs.easy_cases; hard_ = bs.hard_cases;
        bwt_.resize(nout_); if (full_sa) sa_.resize(nout_);
        if (build_rssa) { ssa_.resize(2 * r_); esa_.resize(2 * r_); }
        engine_check(ctx_, pfp_bwt_get(ctx_, bwt_.data(), full_sa ? sa_.data() : nullptr, build_rssa ? ssa_.data() : nullptr, build_rssa ? esa_.data() : nullptr), "pfp_bwt_get");
        built_ = true;
    }

    // the command line's fast path: the build, and the outputs straight from the device to the files (fd < 0: not wanted) --
    // no host copy of the whole output, no per-row callback.  runs() / easy_cases() / hard_cases() are valid afterwards.
    void build_to_files(int fd_bwt, int fd_sa, int fd_ssa, int fd_esa)
    {
        pfp_bwt_sizes bs;
        engine_check(ctx_, pfp_bwt_build(ctx_, build_sa ? 1 : 0, build_rssa ? 1 : 0, &bs), "pfp_bwt_build");
        nout_ = bs.nout; r_ = bs.r; easy_ = bs.easy_cases; hard_ = bs.hard_cases;
        engine_check(ctx_, pfp_bwt_write(ctx_, fd_bwt, build_sa ? fd_sa : -1, build_rssa ? fd_ssa : -1, build_rssa ? fd_esa : -1), "pfp_bwt_write");
    }

    template <typename Fn> void generate_bwt_lcp(Fn out_fn)
    {
        if (verbose) fprintf(stderr, "generating dict suffixes\n");
        // -r without -s: no full SA is built or copied (8 B per base; 256 GB at 32 Gbase).  The rows' SA values are
        // taken from the run samples: exact on every run start and run end -- the only rows on which the reference's
        // callback uses a.sa (src/pfbwt-f.cpp:306-315, 325-328) -- and 0 on the rows inside a run.
        const bool sampled = build_rssa && !build_sa;
        build(!sampled);
        if (verbose) fprintf(stderr, "processing words to build BWT\n");
        uint8_t pbwtc = 0;
        size_t ks = 0, ke = 0;                          // next .ssa / .esa pair
        for (size_t i = 0; i < nout_; ++i) {
            if (sampled) {
                uint_t v = 0;
                if (ks < r_ && ssa_[2 * ks] == (uint_t)i) v = ssa_[2 * ks++ + 1];
                if (ke < r_ && esa_[2 * ke] == (uint_t)i) v = esa_[2 * ke++ + 1];
                out_fn(out_fn_arg((uint_t)i, v, pbwtc, bwt_[i]));
            } else if (any_sa) out_fn(out_fn_arg((uint_t)i, sa_[i], pbwtc, bwt_[i]));      // UPDATE_SA, pfbwt.hpp:87-89
            else out_fn(out_fn_arg(0, 0, pbwtc, bwt_[i]));                            // UPDATE_BWT, :91-92
            pbwtc = bwt_[i];
        }
        fprintf(stderr, "# easy cases: %lu, # hard cases: %lu\n", (unsigned long)easy_, (unsigned long)hard_);
        fprintf(stderr, "sizes: dict: %lu, bwlast: %lu, ilist: %lu, bwsai: %lu, gsa: %lu, glcp: %lu\n", (unsigned long)sizes_[0], (unsigned long)sizes_[1],
                (unsigned long)sizes_[2], (unsigned long)sizes_[3], (unsigned long)sizes_[0], (unsigned long)sizes_[0]);
    }

    const std::vector<uint8_t> &bwt() const { return bwt_; }
    const std::vector<UIntType> &sa() const { return sa_; }
    const std::vector<UIntType> &ssa() const { return ssa_; }
    const std::vector<UIntType> &esa() const { return esa_; }
    uint64_t runs() const { return r_; }
    uint64_t easy_cases() const { return easy_; }
    uint64_t hard_cases() const { return hard_; }

  private:
    std::string fname;
    size_t w = 10;
    uint64_t dsize = 0, dwords = 0;
    bool build_sa = false, build_rssa = false, any_sa = false, verbose = false;
    pfp_ctx *ctx_ = nullptr; bool owns_ = true, built_ = false;
    uint64_t nout_ = 0, r_ = 0, easy_ = 0, hard_ = 0;
    std::vector<uint8_t> bwt_; std::vector<UIntType> sa_, ssa_, esa_;
    size_t sizes_[4] = {0, 0, 0, 0};
};

} // namespace pfbwtf
#endif
