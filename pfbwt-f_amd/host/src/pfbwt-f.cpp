// pfbwt-f_amd/host/src/pfbwt-f.cpp -- command-line front end with the flag surface, stage timer
// lines, stderr statistics and output files of the reference's src/pfbwt-f.cpp (flags :113-153, timers
// :35-50, run_parser :209-245, run_pfbwt :275-349), driving the MI355X engine through the host mirror of
// the reference classes.  Build with -DM64 for pfbwt-f64 (uint_t = 64 bit), without for pfbwt-f.
#include <chrono>
#include <fcntl.h>
#include <getopt.h>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include "file_wrappers.hpp"
#include "pfbwt.hpp"
#include "pfbwt_io.hpp"
#include "pfparser.hpp"

namespace {

struct Options {
    std::string in_fname, output, stdout_ext;
    size_t w = 10, p = 100, n = 0;
    int sa = 0, rssa = 0, mmap = 0, parse_only = 0, trim_non_acgt = 0, non_acgt_to_a = 0, pfbwt_only = 0, verbose = 0, print_docs = 0, gpus = 0;
    std::string devices;      // --devices 0,1,2 (default: 0 .. gpus-1)
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
                    "    -c/--stdout <ext>   output file ending <ext> will be stdout instead (bwt, sa)\n"
                    "    --gpus <int>        (extension) shard the records of a plain FASTA file over <int> devices of this node: sharded parse,\n"
                    "                        one RCCL all-gather of dictionaries, sliced emission; writes .bwt [.sa .ssa .esa] only\n"
                    "    --devices <list>    (extension) the device ids to use with --gpus, comma separated [default: 0,1,...]\n"
                    "    -h                  print this help message\n");
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
                                    {"output", required_argument, NULL, 'o'}, {"gpus", required_argument, NULL, 1006}, {"devices", required_argument, NULL, 1007}, {"window-size", required_argument, NULL, 'w'}, {"mod-val", required_argument, NULL, 'p'}, {0, 0, 0, 0}};
    int c;
    while ((c = getopt_long(argc, argv, "w:p:o:c:hsrfm", lopts, NULL)) != -1) {
        switch (c) {
        case 1000: o.parse_only = 1; break;
        case 1001: o.pfbwt_only = 1; break;
        case 1002: o.trim_non_acgt = 1; break;
        case 1003: o.non_acgt_to_a = 1; break;
        case 1004: o.print_docs = 1; break;
        case 1005: o.verbose = 1; break;
        case 1006: o.gpus = atoi(optarg); break;
        case 1007: o.devices = optarg; break;
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
    if (o.gpus && (o.parse_only || o.pfbwt_only || o.in_fname == "-" || o.print_docs)) die("--gpus builds the index of a plain FASTA file in one go (no --parse-only / --pfbwt-only / stdin / --print-docs)");
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

/* --gpus N (extension; the reference parallelises only merge_pfp, one PfParser per std::thread, src/merge_pfp.cpp:131-152): the
 * records of a plain FASTA file are cut into N runs of whole records of about equal size, rank r's run goes to device r
 * (pfp_parse_feed_fasta on its own host thread), pfp_sharded_build does the rest; the slices are appended to the output files in
 * rank order -- the same .bwt / .sa / .ssa / .esa as the single-device build. */
void run_sharded(const Options &o)
{
    std::vector<int> dev;
    for (size_t i = 0; i < o.devices.size();) { size_t j = o.devices.find(',', i); if (j == std::string::npos) j = o.devices.size(); dev.push_back(atoi(o.devices.substr(i, j - i).c_str())); i = j + 1; }
    if (!dev.empty() && (int)dev.size() != o.gpus) die("--devices must name --gpus devices");
    const int fd = open(o.in_fname.c_str(), O_RDONLY);
    struct stat sb;
    if (fd < 0 || fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size < 2) die(o.in_fname.c_str());
    const size_t fsz = (size_t)sb.st_size;
    const uint8_t *raw = (const uint8_t *)mmap(NULL, fsz, PROT_READ, MAP_PRIVATE, fd, 0);
    if (raw == MAP_FAILED) die("mmap");
    if (raw[0] == 0x1f && raw[1] == 0x8b) die("--gpus needs a plain (not gzip-compressed) FASTA file");
    std::vector<size_t> cut((size_t)o.gpus + 1, fsz);
    cut[0] = 0;
    for (int r = 1; r < o.gpus; ++r) {      // first record start at or behind r / N of the file (a '>' at a line start)
        size_t q = fsz / (size_t)o.gpus * (size_t)r;
        if (q < cut[(size_t)r - 1]) q = cut[(size_t)r - 1];
        while (q < fsz) { const uint8_t *nl = (const uint8_t *)memchr(raw + q, '\n', fsz - q); if (!nl) { q = fsz; break; } q = (size_t)(nl - raw) + 1; if (q < fsz && raw[q] == '>') break; }
        cut[(size_t)r] = q;
    }
    // fewer records than devices (or records of very different sizes): runs without a record are dropped, the build uses fewer ranks
    std::vector<size_t> cuts2; cuts2.push_back(0);
    for (int r = 1; r <= o.gpus; ++r) if (cut[(size_t)r] > cuts2.back()) cuts2.push_back(cut[(size_t)r]);
    if ((int)cuts2.size() - 1 < o.gpus) fprintf(stderr, "only %d run(s) of whole records: using %d device(s)\n", (int)cuts2.size() - 1, (int)cuts2.size() - 1);
    cut = cuts2;
    const_cast<Options &>(o).gpus = (int)cut.size() - 1;
    if (!dev.empty()) dev.resize((size_t)o.gpus);
    int st = 0;
    const unsigned flags = (M64 ? PFP_FLAG_U64 : 0u) | (o.non_acgt_to_a ? PFP_FLAG_NON_ACGT_TO_A : 0u) | ((o.sa || o.rssa) ? PFP_FLAG_SAI : 0u);
    pfp_sharded *sh = pfp_sharded_create((int)o.w, o.p, flags, o.gpus, dev.empty() ? NULL : dev.data(), 0, &st);
    if (!sh) { fprintf(stderr, "pfp_sharded_create: %s\n", pfp_strerror(st)); exit(1); }
    {
        StageTimer t("TASK\tparsing input\t");
        std::vector<int> rc((size_t)o.gpus, PFP_OK);
        std::vector<std::thread> th;
        auto feed = [&](int r) { uint64_t nrec = 0; if (cut[(size_t)r + 1] > cut[(size_t)r]) rc[(size_t)r] = pfp_parse_feed_fasta(pfp_sharded_ctx(sh, r), raw + cut[(size_t)r], cut[(size_t)r + 1] - cut[(size_t)r], PFP_FASTA_FINAL, &nrec); };
        const bool threads = !strcmp(pfp_backend(), "hip-gfx950");      // (the CPU interpreter of the tests runs one kernel at a time)
        for (int r = 0; r < o.gpus; ++r) { if (threads) th.emplace_back(feed, r); else feed(r); }
        for (auto &x : th) x.join();
        for (int r = 0; r < o.gpus; ++r) pfbwtf::engine_check(pfp_sharded_ctx(sh, r), rc[(size_t)r], "pfp_parse_feed_fasta");
    }
    pfp_parse_sizes ps; std::vector<pfp_bwt_sizes> bs((size_t)o.gpus);
    {
        StageTimer t((o.sa || o.rssa) ? "TASK\tgenerating final BWT w/ full and/or run-length SA\t" : "TASK\tgenerating final BWT w/o SA\t");
        st = pfp_sharded_build(sh, o.sa, o.rssa, &ps, bs.data(), NULL, NULL, NULL);
        if (st != PFP_OK) { for (int r = 0; r < o.gpus; ++r) if (st == PFP_E_INVALID_CHAR) pfbwtf::engine_check(pfp_sharded_ctx(sh, r), PFP_OK, ""); fprintf(stderr, "pfp_sharded_build: %s [%s]\n", pfp_strerror(st), pfp_sharded_error(sh)); exit(1); }
        FILE *bwt_fp = open_out(o, "bwt"), *sa_fp = o.sa ? open_out(o, "sa") : NULL, *ssa_fp = o.rssa ? open_out(o, "ssa") : NULL, *esa_fp = o.rssa ? open_out(o, "esa") : NULL;
        fflush(stdout);
        for (int r = 0; r < o.gpus; ++r)      // slice r behind slice r - 1
            pfbwtf::engine_check(pfp_sharded_ctx(sh, r), pfp_bwt_write(pfp_sharded_ctx(sh, r), fileno(bwt_fp), sa_fp ? fileno(sa_fp) : -1, ssa_fp ? fileno(ssa_fp) : -1, esa_fp ? fileno(esa_fp) : -1), "pfp_bwt_write");
        for (FILE *f : {bwt_fp, sa_fp, ssa_fp, esa_fp}) if (f && f != stdout) fclose(f);
    }
    uint64_t r_tot = 0, easy = 0, hard = 0;
    for (auto &b : bs) { r_tot += b.r; easy += b.easy_cases; hard += b.hard_cases; }
    FILE *nf = fopen((o.output + ".n").c_str(), "w");
    if (nf == NULL) die("n file");
    fprintf(nf, "%lu\n", (unsigned long)ps.n); fclose(nf);
    fprintf(stderr, "# easy cases: %lu, # hard cases: %lu\n", (unsigned long)easy, (unsigned long)hard);
    fprintf(stderr, "n: %lu\n", (unsigned long)ps.n);
    fprintf(stderr, "r: %lu\n", (unsigned long)r_tot);
    fprintf(stderr, "n/r: %.3f\n", static_cast<double>(ps.n) / (double)r_tot);
    pfp_sharded_destroy(sh);
    munmap((void *)raw, fsz); close(fd);
}

} // namespace

int main(int argc, char **argv)
{
    Options o = parse_options(argc, argv);
    if (o.gpus > 0) {
        fprintf(stderr, "sharded build over %d device(s)...\n", o.gpus);
        run_sharded(o);
        return 0;
    }
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
