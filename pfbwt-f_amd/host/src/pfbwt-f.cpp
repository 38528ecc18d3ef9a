// pfbwt-f_amd/host/src/pfbwt-f.cpp -- command-line front end with the flag surface, stage timer
// lines, stderr statistics and output files of the reference's src/pfbwt-f.cpp (flags :113-153, timers
// :35-50, run_parser :209-245, run_pfbwt :275-349), driving the MI355X engine through the host mirror of
// the reference classes.  Build with -DM64 for pfbwt-f64 (uint_t = 64 bit), without for pfbwt-f.
#include <chrono>
#include <getopt.h>
#include <string>
#include "file_wrappers.hpp"
#include "pfbwt.hpp"
#include "pfbwt_io.hpp"
#include "pfparser.hpp"

namespace {

struct Options {
    std::string in_fname, output, stdout_ext;
    size_t w = 10, p = 100, n = 0;
    int sa = 0, rssa = 0, mmap = 0, parse_only = 0, trim_non_acgt = 0, non_acgt_to_a = 0, pfbwt_only = 0, verbose = 0, print_docs = 0;
};

struct StageTimer {   // "TASK\t<what>\t<sec>s" on destruction (src/pfbwt-f.cpp:35-50)
    explicit StageTimer(const char *m) : msg(m), t0(std::chrono::system_clock::now()) {}
    ~StageTimer() { fprintf(stderr, "%s%.2fs\n", msg, std::chrono::duration<double>(std::chrono::system_clock::now() - t0).count()); }
    const char *msg; std::chrono::time_point<std::chrono::system_clock> t0;
};

void usage()
{
    fprintf(stderr, "%s. use the prefix-free parsing algorithm to build a BWT for %sgenomic data (MI355X engine).\n\nusage\n    ./%s [options] <fasta file>\n\n", M64 ? "pfbwt-f64" : "pfbwt-f",
            M64 ? "BIG " : "", M64 ? "pfbwt-f64" : "pfbwt-f");
    fprintf(stderr, "results\n    BWT of input saved to <fasta file>.bwt. Header lines are excluded.\n\noptions\n"
                    "    -o                  output prefix.\n    -s                  Output full suffix array to <fasta file>.sa\n"
                    "    -r                  Output run-length sampled suffix arrray to <fasta file>.ssa (run-starts) and <fasta file>.esa (run-ends)\n"
                    "    -w <int>            window-size for parsing [default: 10]\n    -p <int>            modulo for parsing [default: 100]\n"
                    "    -m                  accepted for compatibility (the workspace lives in HBM)\n"
                    "    --parse-only        only produce parse (dict, occ, ilist, last, bwlast)\n"
                    "    --pfbwt-only        build pfbwt from parse + parse-bwt. Requires -o to match parse files' prefix.\n"
                    "    --non-acgt-to-a     map every character outside ACGT to A\n    --print-docs        write <prefix>.docs\n"
                    "    -c/--stdout <ext>   output file ending <ext> will be stdout instead (bwt, sa)\n    -h                  print this help message\n");
}

Options parse_options(int argc, char **argv)
{
    Options o;
    fputs("==== Command line:", stderr);
    for (int i = 0; i < argc; ++i) fprintf(stderr, " %s", argv[i]);
    fputs("\n", stderr);
    static struct option lopts[] = {{"parse-only", no_argument, NULL, 1000}, {"pfbwt-only", no_argument, NULL, 1001}, {"trim-non-acgt", no_argument, NULL, 1002},
                                    {"non-acgt-to-a", no_argument, NULL, 1003}, {"print-docs", no_argument, NULL, 1004}, {"stdout", required_argument, NULL, 'c'},
                                    {"verbose", no_argument, NULL, 1005}, {"sa", no_argument, NULL, 's'}, {"rssa", no_argument, NULL, 'r'}, {"mmap", no_argument, NULL, 'm'},
                                    {"output", required_argument, NULL, 'o'}, {"window-size", required_argument, NULL, 'w'}, {"mod-val", required_argument, NULL, 'p'}, {0, 0, 0, 0}};
    int c;
    while ((c = getopt_long(argc, argv, "w:p:o:c:hsrfm", lopts, NULL)) != -1) {
        switch (c) {
        case 1000: o.parse_only = 1; break;
        case 1001: o.pfbwt_only = 1; break;
        case 1002: o.trim_non_acgt = 1; break;
        case 1003: o.non_acgt_to_a = 1; break;
        case 1004: o.print_docs = 1; break;
        case 1005: o.verbose = 1; break;
        case 'f': break;
        case 's': o.sa = 1; break;
        case 'r': o.rssa = 1; break;
        case 'w': o.w = (size_t)atoi(optarg); break;
        case 'm': o.mmap = 1; break;
        case 'p': o.p = (size_t)atoi(optarg); break;
        case 'h': usage(); exit(0);
        case 'o': o.output.assign(optarg); break;
        case 'c': o.stdout_ext = optarg; break;
        default: fprintf(stderr, "Unknown option. Use -h for help.\n"); exit(1);
        }
    }
    if (argc == optind + 1) o.in_fname.assign(argv[optind]);
    else { fprintf(stderr, "reading from stdin. Parsing might be a bit slow.\n"); o.in_fname.assign("-"); }
    if (o.non_acgt_to_a && o.trim_non_acgt) die("cannot have both --non-acgt-to-a and --trim-non-acgt options enabled at same time");
    if (o.in_fname == "-" && o.output == "" && !o.pfbwt_only) die("if reading from stdin, need a prefix for output files (-o, --output)");
    if (o.in_fname != "-" && o.output == "") o.output = o.in_fname;
    if (o.parse_only && o.pfbwt_only) die("cannot simulatneously do parse_only and pfbwt_only");
    return o;
}

FILE *open_out(const Options &o, const char *ext)
{
    if (o.stdout_ext == ext) return stdout;
    std::string name = o.output + "." + ext;
    FILE *f = fopen(name.c_str(), "wb");
    if (f == NULL) die(name.c_str());
    return f;
}

using parser_t = pfbwtf::PfParser<WangHash>;

/* saves dict, occs, ilist, bwlast (and bwsai) to disk; returns the parser so that stage 2 can adopt its device state */
size_t run_parser(const Options &o, parser_t &p)
{
    size_t n = 0;
    fprintf(stderr, "starting...\n");
    { StageTimer t("TASK\tparsing input\t"); p.add_fasta(o.in_fname); }
    {
        StageTimer t("TASK\tfinalizing parse, writing dict, occs, and ranks\t");
        p.finalize(); n = p.get_n();
        pfbwtf::save_parser(p, o.output);
    }
    {
        StageTimer t("TASK\tranking and bwt-ing parse and processing last-chars\t");
        p.bwt_of_parse([&](const std::vector<char> &bwlast, const std::vector<parser_t::UIntType> &ilist, const std::vector<parser_t::UIntType> &bwsai) {
            pfbwtf::vec_to_file<char>(bwlast, o.output + "." + EXTBWLST);
            pfbwtf::vec_to_file<parser_t::UIntType>(ilist, o.output + "." + EXTILIST);
            if (o.sa || o.rssa) pfbwtf::vec_to_file<parser_t::UIntType>(bwsai, o.output + "." + EXTBWSAI);
        });
    }
    FILE *nf = fopen((o.output + ".n").c_str(), "w");
    if (nf == NULL) die("n file");
    fprintf(nf, "%lu\n", (unsigned long)n);
    fclose(nf);
    return n;
}

size_t read_n_file(const std::string &prefix)
{
    FILE *f = fopen((prefix + ".n").c_str(), "r");
    unsigned long n = 0;
    if (f == NULL || fscanf(f, "%lu", &n) != 1) die("could not read '.n' file");
    fclose(f);
    return n;
}

template <template <typename, typename...> class R, template <typename, typename...> class W> void run_pfbwt(const Options &o, parser_t *parsed)
{
    using pfbwt_t = pfbwtf::PrefixFreeBWT<R, W>;
    pfbwtf::PrefixFreeBWTParams a;
    a.prefix = o.output; a.w = o.w; a.sa = o.sa; a.rssa = o.rssa; a.verb = o.verbose;
    size_t n = o.n;
    if (!n) { fprintf(stderr, "reading n from file\n"); n = read_n_file(o.output); }
    FILE *bwt_fp = open_out(o, "bwt");
    // in one process the parse is still resident on the device: adopt it instead of re-reading the files
    pfbwt_t *p = (parsed && parsed->engine() && parsed->parse_bwt_done()) ? new pfbwt_t(parsed->engine(), a) : new pfbwt_t(a, n);
    {
        StageTimer t((o.sa || o.rssa) ? "TASK\tgenerating final BWT w/ full and/or run-length SA\t" : "TASK\tgenerating final BWT w/o SA\t");
        // generate_bwt_lcp + out_fn, fused on the device; the outputs go from the device to the files in blocks
        FILE *sa_fp = o.sa ? open_out(o, "sa") : NULL, *ssa_fp = o.rssa ? open_out(o, "ssa") : NULL, *esa_fp = o.rssa ? open_out(o, "esa") : NULL;
        fflush(stdout);
        p->build_to_files(fileno(bwt_fp), sa_fp ? fileno(sa_fp) : -1, ssa_fp ? fileno(ssa_fp) : -1, esa_fp ? fileno(esa_fp) : -1);
        for (FILE *f : {bwt_fp, sa_fp, ssa_fp, esa_fp}) if (f && f != stdout) fclose(f);
    }
    fprintf(stderr, "# easy cases: %lu, # hard cases: %lu\n", (unsigned long)p->easy_cases(), (unsigned long)p->hard_cases());
    fprintf(stderr, "n: %lu\n", (unsigned long)n);
    fprintf(stderr, "r: %lu\n", (unsigned long)p->runs());
    fprintf(stderr, "n/r: %.3f\n", static_cast<double>(n) / (double)p->runs());
    delete p;
}

} // namespace

int main(int argc, char **argv)
{
    Options o = parse_options(argc, argv);
    pfbwtf::PfParserParams pp;
    pp.w = o.w; pp.p = o.p; pp.get_sai = o.sa || o.rssa; pp.verbose = o.verbose; pp.trim_non_acgt = o.trim_non_acgt; pp.non_acgt_to_a = o.non_acgt_to_a; pp.store_docs = o.print_docs;
    parser_t parser(pp);
    bool have_parse = false;
    if (!o.pfbwt_only) {
        fprintf(stderr, "running parser...\n");
        o.n = run_parser(o, parser);
        have_parse = true;
    }
    if (!o.parse_only) {
        fprintf(stderr, "generating BWT using pfbwt algorithm...\n");
        fprintf(stderr, "workspace will be contained in device memory (HBM)\n");
        if (o.mmap) run_pfbwt<MMapFileSource, MMapFileSink>(o, have_parse ? &parser : nullptr);
        else run_pfbwt<VecFileSource, VecFileSinkPrivate>(o, have_parse ? &parser : nullptr);
    }
    return 0;
}
