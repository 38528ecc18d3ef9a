// pfbwt-f_amd/host/src/mps_to_ma.cpp -- the reference's src/mps_to_ma.cpp:19-51 on the MI355X engine:
//   mps_to_ma [-o <output, default "out">] [-m] <marker positions (.mps)> <suffix array file | ->
// The suffix array (uint64 values in BWT order, as pfbwt-f64 -s / --stdout sa writes them) is read from a file or from
// stdin ("-": vcf_to_bwt.py:259-285 pipes it through tee); write_marker_array (include/marker_array.hpp:138-174) runs on
// the device (pfp_marker_array).  -m (mmap) is accepted and ignored: the working arrays live in HBM.
#include <getopt.h>
#include <cstdio>
#include <string>
#include <vector>
#include "pfbwtf_common.hpp"

int main(int argc, char **argv)
{
    std::string output = "out";
    static struct option lopts[] = {{"mmap", no_argument, NULL, 'm'}, {"output", required_argument, NULL, 'o'}, {NULL, 0, NULL, 0}};
    int c;
    while ((c = getopt_long(argc, argv, "o:mh", lopts, NULL)) != -1) {
        switch (c) {
        case 'm': break;
        case 'o': output = optarg; break;
        default: fprintf(stderr, "Unknown option.\n"); exit(1);
        }
    }
    if (argc - optind < 2) { fprintf(stderr, "usage: %s [-o output] [-m] <mps file> <sa file | ->\n", argv[0]); exit(1); }
    const std::string mai_fname = argv[optind], sa_fname = argv[optind + 1];
    std::vector<uint64_t> mps = pfbwtf::read_vec<uint64_t>(mai_fname);
    std::vector<uint64_t> sa;
    if (sa_fname == "-") {
        uint64_t buf[1 << 16]; size_t k;
        while ((k = fread(buf, 8, 1 << 16, stdin)) > 0) sa.insert(sa.end(), buf, buf + k);
    } else sa = pfbwtf::read_vec<uint64_t>(sa_fname);
    int st = 0, dev = 0;
    if (const char *e = getenv("PFBWT_DEVICE")) dev = atoi(e);
    pfp_ctx *ctx = pfp_create(10, 100, PFP_FLAG_U64, dev, 0, &st);
    if (!ctx) { fprintf(stderr, "pfp_create: %s\n", pfp_strerror(st)); exit(1); }
    uint64_t words = 0;
    if (!sa.empty()) pfbwtf::engine_check(ctx, pfp_marker_array(ctx, mps.data(), mps.size(), sa.data(), sa.size(), &words), "pfp_marker_array");
    std::vector<uint64_t> out(words);
    pfbwtf::engine_check(ctx, pfp_marker_array_get(ctx, out.data()), "pfp_marker_array_get");
    pfp_destroy(ctx);
    FILE *ofp = fopen(output.c_str(), "wb");
    if (ofp == NULL) die(output.c_str());
    if (words && fwrite(out.data(), 8, words, ofp) != words) die("could not write file");
    fclose(ofp);
    return 0;
}
