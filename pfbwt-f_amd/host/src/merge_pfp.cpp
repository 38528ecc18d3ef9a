// pfbwt-f_amd/host/src/merge_pfp.cpp -- merges parses (loaded from <prefix>.dict/.parse or generated from a FASTA named
// <prefix>) into one, with the flags and outputs of the reference's src/merge_pfp.cpp:29-176.
// The reference folds the operands one by one with PfParser::operator+= (threads over slices of the operand list, :97-152),
// re-inserting every phrase of the right-hand side into a std::map.  Here every operand becomes a SHARD on the device --
// a saved parse is uploaded as it is (pfp_shard_load: dictionary image + ranks; no text is rebuilt), a FASTA is parsed on
// its own -- and ONE pfp_merge_shards unites the dictionaries, re-hashes every seam (pfparser.hpp:226-245) and produces the
// parse of the concatenation.  -t is accepted and ignored.
#include <getopt.h>
#include <iostream>
#include <string>
#include <vector>
#include "pfbwt_io.hpp"
#include "pfparser.hpp"

namespace {
struct Shard { pfp_ctx *ctx = nullptr; pfp_shard_view view; };

pfp_ctx *new_ctx(const pfbwtf::PfParserParams &p)
{
    int st = 0, dev = 0;
    if (const char *e = getenv("PFBWT_DEVICE")) dev = atoi(e);
    pfp_ctx *c = pfp_create((int)p.w, p.p, (M64 ? PFP_FLAG_U64 : 0u) | PFP_FLAG_SAI, dev, 0, &st);
    if (!c) { fprintf(stderr, "pfp_create: %s\n", pfp_strerror(st)); exit(1); }
    return c;
}
} // namespace

int main(int argc, char **argv)
{
    using parser_t = pfbwtf::PfParser<>;
    std::vector<std::string> prefixes;
    std::string output = "out";
    int w = 10, p = 100, store_docs = 0, parse_bwt = 0, sai = 0, c;
    static struct option lopts[] = {{"docs", no_argument, NULL, 'd'}, {"window-size", required_argument, NULL, 'w'}, {"mod-val", required_argument, NULL, 'p'},
                                    {"output", required_argument, NULL, 'o'}, {"threads", required_argument, NULL, 't'}, {"parse-bwt", no_argument, NULL, 1000},
                                    {"sai", no_argument, NULL, 's'}, {0, 0, 0, 0}};
    while ((c = getopt_long(argc, argv, "dw:p:o:t:s", lopts, NULL)) != -1) {
        switch (c) {
        case 'd': store_docs = 1; break;
        case 'w': w = atoi(optarg); break;
        case 'p': p = atoi(optarg); break;
        case 'o': output = optarg; break;
        case 't': break;
        case 's': sai = 1; break;
        case 1000: parse_bwt = 1; break;
        default: std::cerr << "Unknown option.\n"; fprintf(stderr, "usage: ./merge_pfp [--docs] -w <window size> -p <mod> -o <output prefix> -t <threads> <prefix 1> <prefix 2> ... \n"); exit(1);
        }
    }
    for (int i = optind; i < argc; ++i) prefixes.push_back(argv[i]);
    pfbwtf::PfParserParams params;
    params.store_docs = store_docs; params.w = (size_t)w; params.p = (size_t)p; params.get_sai = sai;
    fprintf(stderr, "not using threads (%lu files): the operands are merged by one pass on the GPU\n", (unsigned long)prefixes.size());
    std::vector<Shard> shards;
    std::vector<parser_t::UIntType> doc_starts; std::vector<std::string> doc_names;
    uint64_t n_so_far = 0;
    for (auto &prefix : prefixes) {
        Shard sh;
        if (pfbwtf::parse_files_exist(prefix)) {       // load_parser, pfbwt_io.hpp:211-222
            fprintf(stderr, "loading %s, %s, and maybe %s from file\n", (prefix + ".dict").data(), (prefix + ".parse").data(), (prefix + ".docs").data());
            std::vector<uint8_t> dict = pfbwtf::read_vec<uint8_t>(prefix + ".dict");
            std::vector<uint32_t> ranks = pfbwtf::read_vec<uint32_t>(prefix + ".parse");
            sh.ctx = new_ctx(params);
            pfbwtf::engine_check(sh.ctx, pfp_shard_load(sh.ctx, dict.data(), dict.size(), ranks.data(), ranks.size()), "pfp_shard_load");
            if (store_docs) {
                auto dp = pfbwtf::load_doc_info<parser_t::UIntType>(prefix + ".docs");
                for (size_t i = 0; i < dp.second.size(); ++i) { doc_names.push_back(dp.first[i]); doc_starts.push_back((parser_t::UIntType)(dp.second[i] + n_so_far)); }
            }
        } else if (pfbwtf::file_exists(prefix)) {      // parse_from_fasta, :264-270 -- a stand-alone parse of this operand
            fprintf(stderr, "generating parse for %s\n", prefix.data());
            sh.ctx = new_ctx(params);
            const int st = pfp_parse_feed_fasta_file(sh.ctx, prefix.c_str(), store_docs ? PFP_FASTA_RECORDS : 0u, nullptr);
            if (st == PFP_E_IO) die("failed to open file!\n");
            pfbwtf::engine_check(sh.ctx, st, "pfp_parse_feed_fasta_file");
            if (store_docs) {
                uint64_t nd = 0; pfbwtf::engine_check(sh.ctx, pfp_parse_docs(sh.ctx, &nd), "pfp_parse_docs");
                for (uint64_t i = 0; i < nd; ++i) {
                    const char *nm = nullptr; uint64_t start = 0;
                    pfbwtf::engine_check(sh.ctx, pfp_parse_doc_get(sh.ctx, i, &nm, &start), "pfp_parse_doc_get");
                    doc_names.push_back(nm); doc_starts.push_back((parser_t::UIntType)(n_so_far + start));
                }
            }
            pfbwtf::engine_check(sh.ctx, pfp_parse_finalize_shard(sh.ctx, nullptr), "pfp_parse_finalize_shard");
        } else {
            fprintf(stderr, "ERROR: %s not found, cannot add it to parse!\n", prefix.data());
            continue;
        }
        pfbwtf::engine_check(sh.ctx, pfp_shard_view_get(sh.ctx, &sh.view), "pfp_shard_view_get");
        n_so_far += sh.view.n;
        shards.push_back(sh);
    }
    if (shards.empty()) { fprintf(stderr, "nothing to merge\n"); exit(1); }
    parser_t parser(params);
    std::vector<pfp_shard_view> views;
    for (auto &sh : shards) views.push_back(sh.view);
    parser.merge_device_shards(views);
    for (auto &sh : shards) pfp_destroy(sh.ctx);
    if (store_docs) parser.set_docs(doc_starts, doc_names);
    pfbwtf::save_parser(parser, output);
    if (parse_bwt) pfbwtf::save_parse_bwt(parser, output, sai);
    return 0;
}
