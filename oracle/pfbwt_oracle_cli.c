/*
 * oracle/pfbwt_oracle_cli.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Command-line front end of the CPU restatement; stage structure, stage timer lines and output
 * files follow src/pfbwt-f.cpp:209-245 (run_parser) and :275-349 (run_pfbwt).  bench.py times this
 * binary as the single-threaded CPU baseline ("cpu_baseline.kind = port").
 *
 *   pfbwt_oracle [-s] [-r] [-w W] [-p P] [-o PREFIX] [--u32|--u64] [--non-acgt-to-a]
 *                [--parse-only] [--print-docs] <fasta | ->
 */
#define _GNU_SOURCE
#include "pfp_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <zlib.h>

static double now_s(void)
{
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void die(const char *msg) { perror(msg); exit(1); }

static void write_uvec(const char *prefix, const char *ext, const uint64_t *v, uint64_t cnt, int U)
{
    char name[4096]; FILE *f; uint64_t i;
    snprintf(name, sizeof name, "%s.%s", prefix, ext);
    f = fopen(name, "wb"); if (!f) die(name);
    if (U == 8) { if (fwrite(v, 8, (size_t)cnt, f) != cnt) die("fwrite"); }
    else {
        uint32_t *t = (uint32_t *)malloc(4 * (size_t)(cnt ? cnt : 1));
        for (i = 0; i < cnt; ++i) t[i] = (uint32_t)v[i];
        if (fwrite(t, 4, (size_t)cnt, f) != cnt) die("fwrite");
        free(t);
    }
    fclose(f);
}

static void write_bytes(const char *prefix, const char *ext, const void *v, uint64_t cnt)
{
    char name[4096]; FILE *f;
    snprintf(name, sizeof name, "%s.%s", prefix, ext);
    f = fopen(name, "wb"); if (!f) die(name);
    if (fwrite(v, 1, (size_t)cnt, f) != cnt) die("fwrite");
    fclose(f);
}

/* FASTA records: name = header up to first blank; sequence = all non-newline bytes of the
 * following lines up to the next '>' at line start (what include/kseq.h:178-205 yields for FASTA). */
typedef struct { uint8_t *seqs; uint64_t total, cap; uint64_t *len; char **name; uint64_t nseq, scap; } fasta_t;

static void fasta_read(const char *fname, fasta_t *fa)
{
    gzFile fp = strcmp(fname, "-") ? gzopen(fname, "r") : gzdopen(0, "r");
    static char buf[1 << 16];
    int at_line_start = 1, in_header = 0, have = 0;
    char hdr[1024]; size_t hl = 0; int n;
    if (!fp) die("failed to open file!\n");
    memset(fa, 0, sizeof *fa);
    while ((n = gzread(fp, buf, sizeof buf)) > 0) {
        int i;
        for (i = 0; i < n; ++i) {
            char c = buf[i];
            if (in_header) {
                if (c == '\n') {
                    in_header = 0; at_line_start = 1; hdr[hl] = 0;
                    { size_t k = 0; while (hdr[k] && hdr[k] != ' ' && hdr[k] != '\t') ++k; hdr[k] = 0; }
                    if (fa->nseq == fa->scap) {
                        fa->scap = fa->scap ? fa->scap * 2 : 16;
                        fa->len = (uint64_t *)realloc(fa->len, 8 * (size_t)fa->scap);
                        fa->name = (char **)realloc(fa->name, sizeof(char *) * (size_t)fa->scap);
                    }
                    fa->len[fa->nseq] = 0; fa->name[fa->nseq] = strdup(hdr); fa->nseq++; have = 1;
                } else if (hl + 1 < sizeof hdr) hdr[hl++] = c;
                continue;
            }
            if (c == '\n') { at_line_start = 1; continue; }
            if (at_line_start && c == '>') { in_header = 1; hl = 0; continue; }
            at_line_start = 0;
            if (!have) continue; /* bytes before the first header are skipped (kseq.h:182-186) */
            if (c == '\r') continue;
            if (fa->total == fa->cap) {
                fa->cap = fa->cap ? fa->cap * 2 : (1 << 20);
                fa->seqs = (uint8_t *)realloc(fa->seqs, (size_t)fa->cap);
                if (!fa->seqs) die("realloc");
            }
            fa->seqs[fa->total++] = (uint8_t)c; fa->len[fa->nseq - 1]++;
        }
    }
    gzclose(fp);
}

int main(int argc, char **argv)
{
    int w = 10, sa = 0, rssa = 0, U = 8, parse_only = 0, docs = 0, i;
    uint64_t p = 100; unsigned flags = 0;
    const char *in = NULL, *out = NULL;
    fasta_t fa; orc_parse_t ps; double t0, twall = now_s();
    for (i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "-s")) sa = 1;
        else if (!strcmp(argv[i], "-r")) rssa = 1;
        else if (!strcmp(argv[i], "-w") && i + 1 < argc) w = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-p") && i + 1 < argc) p = (uint64_t)atol(argv[++i]);
        else if (!strcmp(argv[i], "-o") && i + 1 < argc) out = argv[++i];
        else if (!strcmp(argv[i], "--u32")) U = 4;
        else if (!strcmp(argv[i], "--u64")) U = 8;
        else if (!strcmp(argv[i], "--non-acgt-to-a")) flags |= ORC_NON_ACGT_TO_A;
        else if (!strcmp(argv[i], "--parse-only")) parse_only = 1;
        else if (!strcmp(argv[i], "--print-docs")) docs = 1;
        else if (argv[i][0] == '-' && argv[i][1]) { fprintf(stderr, "Unknown option. Use -h for help.\n"); return 1; }
        else in = argv[i];
    }
    if (!in) { fprintf(stderr, "usage: pfbwt_oracle [-s] [-r] [-w W] [-p P] [-o PREFIX] [--u32|--u64] <fasta>\n"); return 1; }
    if (!out) out = in;

    t0 = now_s();
    fasta_read(in, &fa);
    fprintf(stderr, "TASK\treading input\t%.2fs\n", now_s() - t0);
    t0 = now_s();
    {
        int rc = orc_parse(fa.seqs, fa.len, fa.nseq, w, p, flags, &ps);
        if (rc == 1) { fprintf(stderr, "error, invalid character %d/%c -> %d\n", ps.err_char, ps.err_char, 5); return 1; }
        if (rc) { fprintf(stderr, "parse failed\n"); return 1; }
    }
    fprintf(stderr, "TASK\tparsing input + finalizing parse\t%.2fs\n", now_s() - t0);
    t0 = now_s();
    write_bytes(out, "dict", ps.dict, ps.dsize);
    write_uvec(out, "occ", ps.occ, ps.dwords, U);
    write_bytes(out, "parse", ps.parse, 4 * ps.m);
    {
        char name[4096]; FILE *f; uint64_t s, start = 0;
        snprintf(name, sizeof name, "%s.n", out);
        f = fopen(name, "w"); if (!f) die(name); fprintf(f, "%lu\n", (unsigned long)ps.n); fclose(f);
        if (docs) {   /* pfparser.hpp:321-325, pfbwt_io.hpp:224-231 */
            snprintf(name, sizeof name, "%s.docs", out);
            f = fopen(name, "w"); if (!f) die(name);
            for (s = 0; s < fa.nseq; ++s) { fprintf(f, "%s %lu\n", fa.name[s], (unsigned long)start); start += fa.len[s] + (uint64_t)w; }
            fclose(f);
        }
    }
    fprintf(stderr, "TASK\twriting dict, occs, and ranks\t%.2fs\n", now_s() - t0);
    t0 = now_s();
    if (orc_parse_bwt(&ps) == 2) { fprintf(stderr, "error: only one dict word total. Re-run with a smaller p modulus\n"); return 1; }
    write_bytes(out, "bwlast", ps.bwlast, ps.m + 1);
    write_uvec(out, "ilist", ps.ilist, ps.m + 1, U);
    if (sa || rssa) write_uvec(out, "bwsai", ps.bwsai, ps.m + 1, U);
    fprintf(stderr, "TASK\tranking and bwt-ing parse and processing last-chars\t%.2fs\n", now_s() - t0);
    if (!parse_only) {
        uint64_t nout = ps.n + 1, easy = 0, hard = 0, r;
        uint8_t *bwt = (uint8_t *)malloc((size_t)nout);
        uint64_t *sa_raw = (sa || rssa) ? (uint64_t *)malloc(8 * (size_t)nout) : NULL;
        uint64_t *sa_out = sa ? (uint64_t *)malloc(8 * (size_t)nout) : NULL;
        uint64_t *ssa = rssa ? (uint64_t *)malloc(16 * (size_t)nout) : NULL;
        uint64_t *esa = rssa ? (uint64_t *)malloc(16 * (size_t)nout) : NULL;
        int64_t got;
        t0 = now_s();
        got = orc_bwt(ps.dict, ps.dsize, ps.occ, ps.dwords, ps.bwlast, ps.ilist, ps.bwsai, ps.m + 1,
                      w, U, bwt, sa_raw, &easy, &hard);
        if (got != (int64_t)nout) { fprintf(stderr, "emission produced %ld outputs, expected %lu\n", (long)got, (unsigned long)nout); return 1; }
        r = orc_outfn(bwt, sa_raw, nout, ps.n, U, sa_out, ssa, esa);
        write_bytes(out, "bwt", bwt, nout);
        if (sa) write_uvec(out, "sa", sa_out, nout, U);
        if (rssa) { write_uvec(out, "ssa", ssa, 2 * r, U); write_uvec(out, "esa", esa, 2 * r, U); }
        fprintf(stderr, "# easy cases: %lu, # hard cases: %lu\n", (unsigned long)easy, (unsigned long)hard);
        fprintf(stderr, "TASK\tgenerating final BWT%s\t%.2fs\n", (sa || rssa) ? " w/ full and/or run-length SA" : " w/o SA", now_s() - t0);
        fprintf(stderr, "n: %lu\nr: %lu\nn/r: %.3f\n", (unsigned long)ps.n, (unsigned long)r, (double)ps.n / (double)r);
    }
    fprintf(stderr, "WALL\t%.3fs\n", now_s() - twall);
    return 0;
}
