// pfbwt-f_amd/host/src/merge_pfp.cpp -- merges parses (loaded from <prefix>.dict/.parse or generated from
// a FASTA named <prefix>) into one, with the flags and outputs of the reference's src/merge_pfp.cpp:29-176.
// The fold itself is PfParser::operator+= of the host mirror; the merged parse is produced by one GPU
// parse at finalize(), so -t is accepted and ignored.
#include <getopt.h>
#include <iostream>
#include <string>
#include <vector>
#include "pfbwt_io.hpp"
#include "pfparser.hpp"

int main(int argc, char **argv)
{
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
    fprintf(stderr, "not using threads (%lu files): the merged parse is built by one GPU pass\n", (unsigned long)prefixes.size());
    pfbwtf::PfParser<> parser(params);
    for (auto &prefix : prefixes) parser += pfbwtf::load_or_generate_parser_w_log(prefix, params, stderr);
    parser.finalize();
    pfbwtf::save_parser(parser, output);
    if (parse_bwt) pfbwtf::save_parse_bwt(parser, output, sai);
    return 0;
}
