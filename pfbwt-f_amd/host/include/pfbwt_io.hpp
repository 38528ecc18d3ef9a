// pfbwt-f_amd/host/include/pfbwt_io.hpp -- (de)serialisers of the parse artefacts with the on-disk
// layouts of the reference's include/pfbwt_io.hpp:44-297:
//   .dict   phrases in rank order, each followed by 0x01, then 0x00          (:71-82)
//   .occ    dwords x uint_t                                                   (:234-249)
//   .parse  m x uint32 ranks, no terminator                                   (:241)
//   .n      "%lu\n"            .docs  "name start\n"                          (:224-231, 246-248)
//   .bwlast m+1 bytes   .ilist / .bwsai  (m+1) x uint_t                        (:286-297)
#ifndef PFBWTF_PFBWT_IO_HPP
#define PFBWTF_PFBWT_IO_HPP
#include <sys/stat.h>
#include <string>
#include <vector>
#include "pfparser.hpp"

namespace pfbwtf {

template <typename T> void vec_to_file(const std::vector<T> &vec, std::string fname)
{
    FILE *fp = fopen(fname.data(), "wb");
    if (fp == NULL) die(fname.data());
    if (fwrite(vec.data(), sizeof(T), vec.size(), fp) != vec.size()) die("could not write file");
    fclose(fp);
}
template <typename T> void vec_to_file(const std::vector<T> &vec, size_t nelems, std::string fname)
{
    FILE *fp = fopen(fname.data(), "wb");
    if (fp == NULL) die(fname.data());
    if (fwrite(vec.data(), sizeof(T), nelems, fp) != nelems) die("could not write file");
    fclose(fp);
}
template <typename T> std::vector<T> vec_from_file(std::string path) { return read_vec<T>(path); }

inline void dict_to_file(const std::vector<const char *> &phrases, std::string fname)
{
    FILE *fp = fopen(fname.data(), "wb");
    if (fp == NULL) die("unable to open dict file");
    for (auto ph : phrases) { size_t l = strlen(ph); if (fwrite(ph, 1, l, fp) != l) die("Error writing to DICT file\n"); if (fputc(EndOfWord, fp) == EOF) die("Error writing EndOfWord to DICT file"); }
    if (fputc(EndOfDict, fp) == EOF) die("Error writing EndOfDict to DICT file");
    if (fclose(fp)) die("Error closing DICT file");
}
inline std::vector<std::string> load_dict(std::string dict_fname)
{
    std::vector<char> img = read_vec<char>(dict_fname);
    std::vector<std::string> d; size_t start = 0;
    for (size_t i = 0; i < img.size(); ++i) {
        if (img[i] == EndOfDict) break;
        if (img[i] == EndOfWord) { d.emplace_back(img.data() + start, i - start); start = i + 1; }
    }
    return d;
}
template <typename U> std::pair<std::vector<std::string>, std::vector<U>> load_doc_info(std::string fname)
{
    std::vector<std::string> names; std::vector<U> starts;
    FILE *fp = fopen(fname.data(), "r");
    if (fp == NULL) { fprintf(stderr, "error opening doc file %s", fname.data()); exit(1); }
    char name[4096]; unsigned long st;
    while (fscanf(fp, "%4095s %lu", name, &st) == 2) { names.push_back(name); starts.push_back((U)st); }
    fclose(fp);
    return std::make_pair(names, starts);
}
template <typename U> void docs_to_file(std::string fname, const std::vector<std::string> &doc_names, const std::vector<U> &doc_starts)
{
    FILE *fp = fopen(fname.data(), "w");
    if (fp == NULL) die(fname.data());
    for (size_t i = 0; i < doc_starts.size(); ++i) fprintf(fp, "%s %lu\n", doc_names[i].data(), (unsigned long)doc_starts[i]);
    fclose(fp);
}

/* loads parser from .dict and .parse files (pfbwt_io.hpp:211-222) */
inline PfParser<> load_parser(std::string prefix, PfParserParams p)
{
    auto dict = load_dict(prefix + ".dict");
    auto ranks = read_vec<PfParser<>::IntType>(prefix + ".parse");
    if (p.store_docs) { auto dp = load_doc_info<PfParser<>::UIntType>(prefix + ".docs"); return PfParser<>(p, dict, ranks, dp.second, dp.first); }
    return PfParser<>(p, dict, ranks);
}
/* saves parser to .dict, .occ, .parse, .n (and .docs) (pfbwt_io.hpp:234-249) */
inline void save_parser(const PfParser<> &parser, std::string prefix)
{
    vec_to_file(parser.get_dict_image(), prefix + ".dict");
    vec_to_file(parser.get_occs(), prefix + ".occ");
    vec_to_file(parser.get_parse_ranks(), parser.get_parse_size(), prefix + ".parse");
    if (parser.get_params().store_docs) docs_to_file(prefix + ".docs", parser.get_doc_names(), parser.get_doc_starts());
    FILE *fp = fopen((prefix + ".n").data(), "w");
    if (fp == NULL) die("n file");
    fprintf(fp, "%lu\n", (unsigned long)parser.get_n());
    fclose(fp);
}
inline PfParser<> parse_from_fasta(std::string fasta_fname, PfParserParams p) { PfParser<> parser(p); parser.add_fasta(fasta_fname); parser.finalize(); return parser; }
inline int file_exists(std::string fname) { struct stat b; return !stat(fname.data(), &b); }
inline int parse_files_exist(std::string prefix) { return file_exists(prefix + ".dict") && file_exists(prefix + ".parse"); }
inline PfParser<> load_or_generate_parser_w_log(std::string prefix, PfParserParams params, FILE *fp = stderr)
{
    PfParser<> parser;
    if (parse_files_exist(prefix)) {
        fprintf(fp, "loading %s, %s, and maybe %s from file\n", (prefix + ".dict").data(), (prefix + ".parse").data(), (prefix + ".docs").data());
        parser += load_parser(prefix, params);
    } else {
        fprintf(fp, "generating parse for %s\n", prefix.data());
        if (file_exists(prefix)) parser += parse_from_fasta(prefix, params);
        else fprintf(fp, "ERROR: %s not found, cannot add it to parse!\n", prefix.data());
    }
    return parser;
}
inline void save_parse_bwt(PfParser<> &parser, std::string output, bool sa = false)
{
    parser.bwt_of_parse([&](const std::vector<char> &bwlast, const std::vector<PfParser<>::UIntType> &ilist, const std::vector<PfParser<>::UIntType> &bwsai) {
        vec_to_file<char>(bwlast, output + ".bwlast");
        vec_to_file<PfParser<>::UIntType>(ilist, output + ".ilist");
        if (sa) vec_to_file<PfParser<>::UIntType>(bwsai, output + ".bwsai");
    });
}

} // namespace pfbwtf
#endif
