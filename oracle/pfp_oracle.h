/*
 * oracle/pfp_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Single-threaded CPU restatement of the prefix-free-parsing BWT/SA path of alshai/pfbwt-f.
 * It exists to (1) check the HIP path bit-for-bit and (2) be timed as the CPU baseline
 * ("cpu_baseline.kind = port") in bench.py.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may link or call it.  Nothing under pfbwt-f_amd/ includes this header.
 *
 * Parity pinning (see oracle/README.md): checked against the reference's own goldens
 * tests/data/{single_chrom,mult_chroms}.{bwt,sa} and against binaries compiled from the reference's
 * own sources into oracle/_ref/ (merge_pfp for the parse files, gsacak/simplebwt for SA/BWT).
 *
 * Every function cites the reference file:line it restates.
 */
#ifndef PFP_ORACLE_H
#define PFP_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* special symbols -- include/utils.h:8-10 */
#define ORC_DOLLAR 2
#define ORC_ENDOFWORD 1
#define ORC_ENDOFDICT 0

/* parse flags */
#define ORC_NON_ACGT_TO_A 1u

typedef struct {
    /* sizes */
    uint64_t n;       /* text length incl. the w 'A's after every sequence (PfParser::get_n, pfparser.hpp:529-532) */
    uint64_t m;       /* number of phrases in the parse */
    uint64_t dwords;  /* number of distinct phrases */
    uint64_t dsize;   /* bytes of .dict = sum(len)+dwords+1 */
    /* parse stage outputs (malloc'd; free with orc_parse_free) */
    uint8_t  *text;   /* normalised text X, n bytes */
    uint8_t  *dict;   /* .dict image: phrases in rank order, each + 0x01, then 0x00 (pfbwt_io.hpp:71-82) */
    uint64_t *occ;    /* dwords  (.occ)   */
    uint32_t *parse;  /* m ranks, 1-based (.parse) */
    uint8_t  *last;   /* m */
    uint64_t *sai;    /* m */
    /* parse-BWT outputs, filled by orc_parse_bwt (m+1 entries each) */
    uint8_t  *bwlast;
    uint64_t *ilist;
    uint64_t *bwsai;
    /* error reporting: 0 ok; 1 invalid character (hash.hpp:31) */
    int      err;
    uint64_t err_pos;
    int      err_char;
} orc_parse_t;

/* hash.hpp:12-21 */
uint64_t orc_wang_hash(uint64_t key);

/* pfparser.hpp:299-369 + 484-517.  `seqs` = the raw sequence bytes of all records concatenated (no
 * newlines / headers), `seq_len[i]` their lengths.  Produces dict/occ/parse/last/sai. */
int orc_parse(const uint8_t *seqs, const uint64_t *seq_len, uint64_t nseq,
              int w, uint64_t p, unsigned flags, orc_parse_t *out);

/* pfparser.hpp:379-467: SA of the parse, bwlast / bwsai / ilist. Returns 0, or 2 if the parse has one phrase. */
int orc_parse_bwt(orc_parse_t *ps);

void orc_parse_free(orc_parse_t *ps);

/* pfbwt.hpp:96-194 (+206-239): sequential emission from the on-disk arrays.
 * bwt: n+1 bytes; sa_raw: n+1 values "bwsai - suff_len" exactly as handed to out_fn (row 0 NOT yet
 * replaced by n), may be NULL.  Returns number of outputs (n+1) or <0 on error. */
int64_t orc_bwt(const uint8_t *dict, uint64_t dsize, const uint64_t *occ, uint64_t dwords,
                const uint8_t *bwlast, const uint64_t *ilist, const uint64_t *bwsai, uint64_t nrows,
                int w, int U, uint8_t *bwt, uint64_t *sa_raw,
                uint64_t *easy_cases, uint64_t *hard_cases);

/* gSA and gLCP of a .dict image with gsacak's conventions (ties by position, LCP stops at the
 * separator), for pinning against oracle/_ref/libgsacak*.so */
int orc_gsa_lcp(const uint8_t *dict, uint64_t dsize, uint64_t dwords, uint64_t *gsa, uint64_t *lcp);

/* src/pfbwt-f.cpp:298-320,325-328: the CLI's out_fn.  sa_out (n+1) gets row 0 := n; ssa/esa get
 * (row, sa) pairs (2*r values each; caller allocates 2*(n+1)).  U = 4 or 8 gives the wrap width of
 * the raw values.  Returns r. */
uint64_t orc_outfn(const uint8_t *bwt, const uint64_t *sa_raw, uint64_t nout, uint64_t n, int U,
                   uint64_t *sa_out, uint64_t *ssa, uint64_t *esa);

/* suffix sorters used by the two stages (own SA-IS; role of gsa/gsacak.c:2499-2524) */
int orc_sais_int(const uint32_t *s, uint64_t *SA, uint64_t n, uint64_t k); /* s[n-1]==0 unique */
int orc_sais_bytes(const uint8_t *s, uint64_t *SA, uint64_t n);           /* s[n-1]==0 unique */

#ifdef __cplusplus
}
#endif
#endif
