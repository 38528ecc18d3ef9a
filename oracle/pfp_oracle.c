/*
 * oracle/pfp_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see pfp_oracle.h).
 *
 * CPU restatement (plain C, one thread) of the hot path of alshai/pfbwt-f:
 *   parse        include/pfparser.hpp:299-369, 484-517, 595-601   + include/hash.hpp:12-43
 *   parse-BWT    include/pfparser.hpp:379-467
 *   emission     include/pfbwt.hpp:96-194, 206-239, 259-268
 *   out_fn       src/pfbwt-f.cpp:298-328
 * The suffix sorter is an own SA-IS (the role gsa/gsacak.c plays in the reference); because every
 * file on the path is canonical (SURVEY.md 8c "extra facts"), any correct suffix sorter yields the
 * same bytes.  Pinned by tests/test_oracle_golden.py against the reference's own goldens and, when
 * /root/reference is present, against oracle/_ref binaries built from the reference's sources.
 */
#include "pfp_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------- */
/* hash.hpp:12-21 */
uint64_t orc_wang_hash(uint64_t key)
{
    key = (~key) + (key << 21);
    key = key ^ (key >> 24);
    key = (key + (key << 3)) + (key << 8);
    key = key ^ (key >> 14);
    key = (key + (key << 2)) + (key << 4);
    key = key ^ (key >> 28);
    key = key + (key << 31);
    return key;
}

/* src/utils.c:139-161 (seq_nt4_ntoa_table): A,a,N,n->0  C,c->1  G,g->2  T,t,'-'->3  else 5 */
static int ntoa_code(int c)
{
    switch (c) {
    case 'A': case 'a': case 'N': case 'n': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': case '-': return 3;
    default: return 5;
    }
}
/* src/utils.c:163-180 (seq_nt4_table): ACGTacgt -> 0..3, everything else 4 */
static int nt4_code(int c)
{
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
    }
}

/* ------------------------------------------------------------------------------------------- */
/* SA-IS over an integer text with a unique smallest sentinel T[n-1]==0.  Index type: int32 (texts and dictionaries below
 * 2^31 symbols: everything the test suite runs) or, built with -DORACLE_IDX64 (`make -C oracle big`: pfbwt_oracle64), int64 --
 * three times the memory, for the one-off full-size checks of texts / dictionaries beyond 2^31 (S-3G, prefixes of S-32G). */
#ifdef ORACLE_IDX64
typedef int64_t sidx;
#define SIDX_LIMIT 0x7fffffffffffffffULL
#else
typedef int32_t sidx;
#define SIDX_LIMIT 0x7fffffffULL
#endif
#define TGET(i) ((tb[(i) >> 3] >> ((i) & 7)) & 1)
#define TSET(i, b) (tb[(i) >> 3] = (uint8_t)((b) ? (tb[(i) >> 3] | (1u << ((i) & 7))) : (tb[(i) >> 3] & ~(1u << ((i) & 7)))))
#define ISLMS(i) ((i) > 0 && TGET(i) && !TGET((i) - 1))

static void sais_buckets(const sidx *T, sidx *bkt, sidx n, sidx K, int end)
{
    sidx i, sum = 0;
    for (i = 0; i < K; ++i) bkt[i] = 0;
    for (i = 0; i < n; ++i) bkt[T[i]]++;
    for (i = 0; i < K; ++i) { sum += bkt[i]; bkt[i] = end ? sum : sum - bkt[i]; }
}

static void sais_induce(const sidx *T, sidx *SA, sidx *bkt, const uint8_t *tb, sidx n, sidx K)
{
    sidx i, j;
    sais_buckets(T, bkt, n, K, 0);
    for (i = 0; i < n; ++i) {
        if (SA[i] > 0) { j = SA[i] - 1; if (!TGET(j)) SA[bkt[T[j]]++] = j; }
    }
    sais_buckets(T, bkt, n, K, 1);
    for (i = n - 1; i >= 0; --i) {
        if (SA[i] > 0) { j = SA[i] - 1; if (TGET(j)) SA[--bkt[T[j]]] = j; }
    }
}

static int sais_rec(const sidx *T, sidx *SA, sidx n, sidx K)
{
    sidx i, j, n1 = 0, name = 0, prev = -1;
    uint8_t *tb;
    sidx *bkt;
    if (n == 1) { SA[0] = 0; return 0; }
    tb = (uint8_t *)calloc((size_t)n / 8 + 1, 1);
    bkt = (sidx *)malloc(sizeof(sidx) * (size_t)K);
    if (!tb || !bkt) { free(tb); free(bkt); return -1; }
    TSET(n - 1, 1);
    for (i = n - 2; i >= 0; --i)
        TSET(i, (T[i] < T[i + 1] || (T[i] == T[i + 1] && TGET(i + 1))) ? 1 : 0);
    /* stage 1: sort LMS substrings */
    sais_buckets(T, bkt, n, K, 1);
    for (i = 0; i < n; ++i) SA[i] = -1;
    for (i = 1; i < n; ++i) if (ISLMS(i)) SA[--bkt[T[i]]] = i;
    sais_induce(T, SA, bkt, tb, n, K);
    for (i = 0; i < n; ++i) if (SA[i] >= 0 && ISLMS(SA[i])) SA[n1++] = SA[i];
    for (i = n1; i < n; ++i) SA[i] = -1;
    for (i = 0; i < n1; ++i) {
        sidx pos = SA[i], d; int diff = 0;
        for (d = 0; d < n; ++d) {
            if (prev == -1 || T[pos + d] != T[prev + d] || TGET(pos + d) != TGET(prev + d)) { diff = 1; break; }
            else if (d > 0 && (ISLMS(pos + d) || ISLMS(prev + d))) break;
        }
        if (diff) { name++; prev = pos; }
        SA[n1 + pos / 2] = name - 1;
    }
    for (i = n - 1, j = n - 1; i >= n1; --i) if (SA[i] >= 0) SA[j--] = SA[i];
    {
        sidx *SA1 = SA, *s1 = SA + n - n1;
        if (name < n1) {
            if (sais_rec(s1, SA1, n1, name) < 0) { free(tb); free(bkt); return -1; }
        } else {
            for (i = 0; i < n1; ++i) SA1[s1[i]] = i;
        }
        /* stage 3 */
        sais_buckets(T, bkt, n, K, 1);
        for (i = 1, j = 0; i < n; ++i) if (ISLMS(i)) s1[j++] = i;
        for (i = 0; i < n1; ++i) SA1[i] = s1[SA1[i]];
        for (i = n1; i < n; ++i) SA[i] = -1;
        for (i = n1 - 1; i >= 0; --i) { j = SA[i]; SA[i] = -1; SA[--bkt[T[j]]] = j; }
        sais_induce(T, SA, bkt, tb, n, K);
    }
    free(tb); free(bkt);
    return 0;
}

static int sais32(const sidx *T, sidx *SA, uint64_t n, uint64_t K)
{
    if (n >= SIDX_LIMIT || K >= SIDX_LIMIT) return -1;
    return sais_rec(T, SA, (sidx)n, (sidx)K);
}

int orc_sais_int(const uint32_t *s, uint64_t *SA, uint64_t n, uint64_t k)
{
#ifdef ORACLE_IDX64
    /* sidx is as wide as the caller's SA: sort in place, the text widened once */
    sidx *t = (sidx *)malloc(sizeof(sidx) * (size_t)(n ? n : 1));
    uint64_t i; int r;
    if (!t) return -1;
    for (i = 0; i < n; ++i) t[i] = (sidx)s[i];
    r = sais32(t, (sidx *)SA, n, k);
    free(t);
    return r;
#else
    sidx *sa = (sidx *)malloc(sizeof(sidx) * (size_t)(n ? n : 1));
    uint64_t i; int r;
    if (!sa) return -1;
    r = sais32((const sidx *)s, sa, n, k);
    if (r == 0) for (i = 0; i < n; ++i) SA[i] = (uint64_t)sa[i];
    free(sa);
    return r;
#endif
}

int orc_sais_bytes(const uint8_t *s, uint64_t *SA, uint64_t n)
{
    sidx *t = (sidx *)malloc(sizeof(sidx) * (size_t)(n ? n : 1));
    uint64_t i; int r = -1;
#ifdef ORACLE_IDX64
    if (t) { for (i = 0; i < n; ++i) t[i] = s[i]; r = sais32(t, (sidx *)SA, n, 256); }
    free(t);
#else
    sidx *sa = (sidx *)malloc(sizeof(sidx) * (size_t)(n ? n : 1));
    if (t && sa) {
        for (i = 0; i < n; ++i) t[i] = s[i];
        r = sais32(t, sa, n, 256);
        if (r == 0) for (i = 0; i < n; ++i) SA[i] = (uint64_t)sa[i];
    }
    free(t); free(sa);
#endif
    return r;
}

/* ------------------------------------------------------------------------------------------- */
/* parse: pfparser.hpp:299-369 (add_fasta), 484-492 (finalize), 494-517 (sort_dict,
 * generate_ranks), 595-601 (process_phrase), 471-480 (get_occs)                               */
typedef struct { uint64_t off; uint64_t len; uint64_t cnt; uint32_t rank; } phrase_t;

static const uint8_t *g_Y; /* qsort context */
static int phrase_cmp(const void *a, const void *b)
{
    const phrase_t *x = (const phrase_t *)a, *y = (const phrase_t *)b;
    uint64_t l = x->len < y->len ? x->len : y->len;
    int c = memcmp(g_Y + x->off, g_Y + y->off, (size_t)l);
    if (c) return c;
    return x->len < y->len ? -1 : (x->len > y->len ? 1 : 0);
}

static uint64_t bytes_hash(const uint8_t *p, uint64_t len)
{
    uint64_t h = 0xcbf29ce484222325ULL ^ len, i = 0;
    for (; i + 8 <= len; i += 8) {
        uint64_t v; memcpy(&v, p + i, 8);
        h = (h ^ v) * 0x9e3779b97f4a7c15ULL; h ^= h >> 29;
    }
    for (; i < len; ++i) { h = (h ^ p[i]) * 0x100000001b3ULL; }
    h ^= h >> 32; h *= 0xd6e8feb86659fd93ULL; h ^= h >> 32;
    return h;
}

int orc_parse(const uint8_t *seqs, const uint64_t *seq_len, uint64_t nseq,
              int w, uint64_t p, unsigned flags, orc_parse_t *out)
{
    uint64_t n = 0, i, s, pos, m = 0, mcap, src = 0;
    uint8_t *Y;           /* Dollar + X + Dollar^w  (phrase j is a substring of Y) */
    uint64_t *pstart, *plen;
    uint64_t kmer = 0, mask, phrase_start;
    memset(out, 0, sizeof(*out));
    if (w < 1 || w > 32 || p < 1) return -1;           /* check_w, pfparser.hpp:371-376 */
    /* hash.hpp:26 computes (1ULL << 2*k) - 1 with a run-time k; at k==32 the x86 shift count wraps
     * to 0 and the mask becomes 0 (undefined behaviour in C; mirrored as the observed value). */
    mask = (w == 32) ? 0 : ((1ULL << (2 * w)) - 1);
    for (s = 0; s < nseq; ++s) n += seq_len[s] + (uint64_t)w;
    Y = (uint8_t *)malloc((size_t)(n + 1 + (uint64_t)w + 1));
    if (!Y) return -1;
    Y[0] = ORC_DOLLAR;
    /* X = concat(map(seq) + 'A'^w), pfparser.hpp:335-344 */
    pos = 0;
    mcap = n / (p > 4 ? p / 4 : 1) + 16;
    pstart = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)mcap);
    plen = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)mcap);
    if (!pstart || !plen) return -1;
    phrase_start = 0; /* index into Y of the current phrase's first byte */
    for (s = 0; s < nseq; ++s) {
        uint64_t L = seq_len[s];
        for (i = 0; i < L + (uint64_t)w; ++i) {
            int c = i < L ? seqs[src + i] : 'A', x;
            if (c >= 'a' && c <= 'z') c -= 32;                 /* std::toupper, C locale */
            if ((flags & ORC_NON_ACGT_TO_A) && nt4_code(c) > 3) c = 'A';
            x = ntoa_code(c);
            if (x > 3) {                                        /* hash.hpp:31 */
                out->err = 1; out->err_pos = pos; out->err_char = c;
                free(Y); free(pstart); free(plen);
                return 1;
            }
            Y[1 + pos] = (uint8_t)c;
            kmer = ((kmer << 2) | (uint64_t)x) & mask;          /* hash.hpp:32 */
            /* pos_ in the reference is pos+1 here; trigger test pfparser.hpp:347 */
            if (pos + 1 > (uint64_t)w && orc_wang_hash(kmer) % p == 0) {
                if (m + 2 > mcap) {
                    mcap *= 2;
                    pstart = (uint64_t *)realloc(pstart, sizeof(uint64_t) * (size_t)mcap);
                    plen = (uint64_t *)realloc(plen, sizeof(uint64_t) * (size_t)mcap);
                    if (!pstart || !plen) return -1;
                }
                pstart[m] = phrase_start; plen[m] = (pos + 1) - phrase_start + 1; /* Y[phrase_start .. pos+1] */
                ++m;
                phrase_start = (pos + 1) - (uint64_t)w + 1;    /* keep the last w chars, :349 */
            }
            ++pos;
        }
        src += L;
    }
    /* finalize, pfparser.hpp:484-489: append w Dollars, last phrase */
    for (i = 0; i < (uint64_t)w; ++i) Y[1 + n + i] = ORC_DOLLAR;
    pstart[m] = phrase_start; plen[m] = (n + (uint64_t)w) - phrase_start + 1;
    ++m;

    out->n = n; out->m = m;
    out->text = (uint8_t *)malloc((size_t)(n ? n : 1));
    memcpy(out->text, Y + 1, (size_t)n);
    out->last = (uint8_t *)malloc((size_t)m);
    out->sai = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)m);
    out->parse = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)m);
    for (i = 0; i < m; ++i) {
        out->last[i] = Y[pstart[i] + plen[i] - (uint64_t)w - 1];                 /* :599 */
        out->sai[i] = (i + 1 < m) ? (pstart[i] + plen[i] - 1) : (n + (uint64_t)w); /* :600, :487-488 (pos_) */
    }
    /* dictionary: distinct phrases (std::map in the reference, :595-597); open addressing here */
    {
        uint64_t cap = 16, used = 0, dsize = 1, k;
        uint64_t *slot; phrase_t *ph; uint64_t *pid;
        while (cap < 2 * m) cap <<= 1;
        slot = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)cap);
        ph = (phrase_t *)malloc(sizeof(phrase_t) * (size_t)m);
        pid = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)m);
        if (!slot || !ph || !pid) return -1;
        memset(slot, 0xff, sizeof(uint64_t) * (size_t)cap);
        for (i = 0; i < m; ++i) {
            uint64_t h = bytes_hash(Y + pstart[i], plen[i]) & (cap - 1);
            for (;;) {
                uint64_t e = slot[h];
                if (e == UINT64_MAX) {
                    ph[used].off = pstart[i]; ph[used].len = plen[i]; ph[used].cnt = 1; ph[used].rank = 0;
                    slot[h] = used; pid[i] = used; ++used; break;
                }
                if (ph[e].len == plen[i] && !memcmp(Y + ph[e].off, Y + pstart[i], (size_t)plen[i])) {
                    ph[e].cnt++; pid[i] = e; break;
                }
                h = (h + 1) & (cap - 1);
            }
        }
        /* sort_dict :494-502 (strcmp order), generate_ranks :504-517 (1-based) */
        {
            uint64_t *order = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)used);
            phrase_t *sorted = (phrase_t *)malloc(sizeof(phrase_t) * (size_t)used);
            for (k = 0; k < used; ++k) { sorted[k] = ph[k]; sorted[k].rank = (uint32_t)k; /* remember id */ }
            g_Y = Y;
            qsort(sorted, (size_t)used, sizeof(phrase_t), phrase_cmp);
            for (k = 0; k < used; ++k) order[sorted[k].rank] = k;        /* id -> rank-1 */
            out->dwords = used;
            out->occ = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)used);
            for (k = 0; k < used; ++k) { out->occ[k] = sorted[k].cnt; dsize += sorted[k].len + 1; }
            out->dsize = dsize;
            out->dict = (uint8_t *)malloc((size_t)dsize);
            {
                uint64_t o = 0;
                for (k = 0; k < used; ++k) {                              /* dict_to_file, pfbwt_io.hpp:71-82 */
                    memcpy(out->dict + o, Y + sorted[k].off, (size_t)sorted[k].len);
                    o += sorted[k].len; out->dict[o++] = ORC_ENDOFWORD;
                }
                out->dict[o++] = ORC_ENDOFDICT;
            }
            for (i = 0; i < m; ++i) out->parse[i] = (uint32_t)(order[pid[i]] + 1);
            free(order); free(sorted);
        }
        free(slot); free(ph); free(pid);
    }
    free(Y); free(pstart); free(plen);
    return 0;
}

/* pfparser.hpp:379-467 */
int orc_parse_bwt(orc_parse_t *ps)
{
    uint64_t m = ps->m, i, k = 0;
    uint32_t *P; uint64_t *SA, *F, *W;
    if (m == 1) return 2;                                         /* :390-392 */
    P = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(m + 1));
    SA = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(m + 1));
    W = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(m + 1));
    memcpy(P, ps->parse, sizeof(uint32_t) * (size_t)m);
    P[m] = 0;                                                     /* :407-410 */
    for (i = 0; i < m; ++i) if (P[i] > k) k = P[i];               /* :412-415 */
    if (orc_sais_int(P, SA, m + 1, k + 1) < 0) return -1;         /* :425 */
    ps->bwlast = (uint8_t *)malloc((size_t)(m + 1));
    ps->bwsai = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(m + 1));
    ps->ilist = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(m + 1));
    /* row 0: SA[0]==m  (:430-435) */
    W[0] = P[m - 1]; ps->bwlast[0] = ps->last[m - 2]; ps->bwsai[0] = ps->sai[m - 1];
    for (i = 1; i < m + 1; ++i) {                                 /* :436-451 */
        if (!SA[i]) { W[i] = 0; ps->bwlast[i] = 0; ps->bwsai[i] = 0; }
        else {
            ps->bwlast[i] = (SA[i] == 1) ? ps->last[m - 1] : ps->last[SA[i] - 2];
            ps->bwsai[i] = ps->sai[SA[i] - 1];
            W[i] = P[SA[i] - 1];
        }
    }
    F = (uint64_t *)calloc((size_t)(ps->dwords + 1), sizeof(uint64_t));     /* :452-456 */
    F[1] = 1;
    for (i = 2; i < ps->dwords + 1; ++i) F[i] = F[i - 1] + ps->occ[i - 2];
    for (i = 0; i < m + 1; ++i) ps->ilist[F[W[i]]++] = i;          /* :459-462 */
    free(P); free(SA); free(W); free(F);
    return 0;
}

void orc_parse_free(orc_parse_t *ps)
{
    free(ps->text); free(ps->dict); free(ps->occ); free(ps->parse); free(ps->last); free(ps->sai);
    free(ps->bwlast); free(ps->ilist); free(ps->bwsai);
    memset(ps, 0, sizeof(*ps));
}

/* ------------------------------------------------------------------------------------------- */
/* emission: pfbwt.hpp:96-194.  The gSA comes from the own SA-IS over the dict bytes (0x01
 * separators sort as ordinary symbols; only the order among byte-identical suffixes differs from
 * gsacak, which the emission is insensitive to), the LCP from Kasai's algorithm.                */
typedef struct { uint64_t bwtp; uint8_t bwtc; } suff_t;
static int suff_cmp(const void *a, const void *b)
{
    uint64_t x = ((const suff_t *)a)->bwtp, y = ((const suff_t *)b)->bwtp;
    return x < y ? -1 : (x > y);
}

static int sidx_cmp(const void *a, const void *b) { sidx x = *(const sidx *)a, y = *(const sidx *)b; return x < y ? -1 : (x > y); }

static int64_t orc_bwt_impl(const uint8_t *dict, uint64_t dsize, const uint64_t *occ, uint64_t dwords,
                const uint8_t *bwlast, const uint64_t *ilist, const uint64_t *bwsai, uint64_t nrows,
                int w, int U, uint8_t *bwt, uint64_t *sa_raw,
                uint64_t *easy_cases, uint64_t *hard_cases, uint64_t *gsa_out, uint64_t *lcp_out)
{
    sidx *T, *gsa, *rank, *lcp;
    uint32_t *wid;        /* dict offset -> word index (role of dict_idx.rank, pfbwt.hpp:83-85) */
    uint64_t *wend;       /* word index -> offset of its EndOfWord (role of dict_idx.select) */
    uint64_t *F;          /* word index -> first ilist slot (role of ilist_idx, pfbwt.hpp:226-268) */
    uint64_t i, next, pos = 0, easy = 0, hard = 0, k, wcount = 0;
    const uint64_t umask = (U == 4) ? 0xffffffffULL : UINT64_MAX;
    const int any_sa = (sa_raw != NULL);
    suff_t *suffs = NULL; uint64_t suffs_cap = 0;
    uint8_t *chars = NULL; uint64_t *words = NULL; uint64_t cw_cap = 0;
    (void)nrows;
    if (dsize < 1 || dsize >= SIDX_LIMIT) return -1;
    T = (sidx *)malloc(sizeof(sidx) * (size_t)dsize);
    gsa = (sidx *)malloc(sizeof(sidx) * (size_t)dsize);
    rank = (sidx *)malloc(sizeof(sidx) * (size_t)dsize);
    lcp = (sidx *)malloc(sizeof(sidx) * (size_t)dsize);
    wid = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)dsize);
    wend = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(dwords + 1));
    F = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(dwords + 1));
    if (!T || !gsa || !rank || !lcp || !wid || !wend || !F) return -1;
    for (i = 0; i < dsize; ++i) {
        T[i] = dict[i]; wid[i] = (uint32_t)wcount;
        if (dict[i] == ORC_ENDOFWORD) { if (wcount < dwords) wend[wcount] = i; ++wcount; }
    }
    if (wcount != dwords) return -2;
    /* sort_dict_suffixes, pfbwt.hpp:206-223 */
    if (sais32(T, gsa, dsize, 256) < 0) return -1;
    for (i = 0; i < dsize; ++i) rank[gsa[i]] = (sidx)i;
    {   /* Kasai LCP */
        uint64_t h = 0;
        lcp[0] = 0;
        for (i = 0; i < dsize; ++i) {
            if (rank[i] > 0) {
                uint64_t j = (uint64_t)gsa[rank[i] - 1];
                while (i + h < dsize && j + h < dsize && dict[i + h] == dict[j + h]) ++h;
                lcp[rank[i]] = (sidx)h;
                if (h > 0) --h;
            } else h = 0;
        }
    }
    /* gsacak orders byte-identical suffixes of different words by position (separator i < separator j
     * for i < j, gsa/gsacak.c:877-912) and its LCP stops at the separator; the plain suffix sort above
     * orders them by what follows the separator.  The emission is sensitive to WHICH member of such a
     * group comes first (pfbwt.hpp:116 vs :129), so restore gsacak's order and LCP values here. */
    for (i = 1; i < dsize;) {
        uint64_t g = (uint64_t)gsa[i], sl = (dict[g] == ORC_ENDOFDICT) ? 0 : wend[wid[g]] - g, j = i + 1, x;
        while (j < dsize && (uint64_t)lcp[j] >= sl + 1) ++j;
        if (j - i > 1) {
            /* insertion sort by position (groups are small) or qsort for big ones */
            if (j - i > 32) qsort(gsa + i, (size_t)(j - i), sizeof(sidx), sidx_cmp);
            else for (x = i + 1; x < j; ++x) { sidx v = gsa[x]; uint64_t y = x; while (y > i && gsa[y - 1] > v) { gsa[y] = gsa[y - 1]; --y; } gsa[y] = v; }
            for (x = i + 1; x < j; ++x) lcp[x] = (sidx)sl;
        }
        i = j;
    }
    if (gsa_out) for (i = 0; i < dsize; ++i) gsa_out[i] = (uint64_t)gsa[i];
    if (lcp_out) for (i = 0; i < dsize; ++i) lcp_out[i] = (uint64_t)lcp[i];
    if (!bwt) { free(T); free(gsa); free(rank); free(lcp); free(wid); free(wend); free(F); return 0; }
    F[0] = 1;                                                     /* ilist[0] is the EOS row */
    for (k = 1; k <= dwords; ++k) F[k] = F[k - 1] + occ[k - 1];

#define EMIT(c, q, sl) do { bwt[pos] = (uint8_t)(c); if (any_sa) sa_raw[pos] = (bwsai[(q)] - (uint64_t)(sl)) & umask; ++pos; } while (0)

    for (i = dwords + (uint64_t)w + 1; i < dsize; i = next) {      /* :111 */
        uint64_t g = (uint64_t)gsa[i], wordi = wid[g], suff_len = wend[wordi] - g;   /* :113 */
        next = i + 1;
        if (suff_len <= (uint64_t)w) continue;                     /* :114 */
        if (g == 0 || dict[g - 1] == ORC_ENDOFWORD) {              /* full word, :116-128 */
            for (k = F[wordi]; k < F[wordi + 1]; ++k) { uint64_t q = ilist[k]; EMIT(bwlast[q], q, suff_len); ++easy; }
        } else {
            uint64_t nw = 0, j; uint8_t pc = (g - 1) ? dict[g - 1] : 0, c; int same_char = 1;
            if (cw_cap < 1) { cw_cap = 64; chars = (uint8_t *)malloc((size_t)cw_cap); words = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)cw_cap); }
            chars[nw] = pc; words[nw] = wordi; ++nw;
            for (j = i + 1; j < dsize && (uint64_t)lcp[j] >= suff_len; ++j) {   /* :137-145 */
                uint64_t g2 = (uint64_t)gsa[j], w2 = wid[g2];
                if (wend[w2] - g2 != suff_len) return -3;          /* "something went wrong!" */
                c = (g2 - 1) ? dict[g2 - 1] : 0;
                if (nw == cw_cap) { cw_cap *= 2; chars = (uint8_t *)realloc(chars, (size_t)cw_cap); words = (uint64_t *)realloc(words, sizeof(uint64_t) * (size_t)cw_cap); }
                chars[nw] = c; words[nw] = w2; ++nw;
                same_char = same_char ? (c == pc) : 0;
                pc = c;
            }
            if ((!any_sa && same_char) || (any_sa && nw == 1)) {    /* :146-159 */
                uint64_t x;
                for (x = 0; x < nw; ++x)
                    for (k = F[words[x]]; k < F[words[x] + 1]; ++k) { EMIT(chars[0], ilist[k], suff_len); ++easy; }
            } else {                                               /* :163-181 */
                uint64_t ns = 0, x;
                for (x = 0; x < nw; ++x) ns += F[words[x] + 1] - F[words[x]];
                if (ns > suffs_cap) { suffs_cap = ns * 2; suffs = (suff_t *)realloc(suffs, sizeof(suff_t) * (size_t)suffs_cap); }
                ns = 0;
                for (x = 0; x < nw; ++x)
                    for (k = F[words[x]]; k < F[words[x] + 1]; ++k) { suffs[ns].bwtc = chars[x]; suffs[ns].bwtp = ilist[k]; ++ns; }
                qsort(suffs, (size_t)ns, sizeof(suff_t), suff_cmp);
                for (x = 0; x < ns; ++x) { EMIT(suffs[x].bwtc, suffs[x].bwtp, suff_len); ++hard; }
            }
            next = j;
        }
    }
#undef EMIT
    if (easy_cases) *easy_cases = easy;
    if (hard_cases) *hard_cases = hard;
    free(T); free(gsa); free(rank); free(lcp); free(wid); free(wend); free(F);
    free(suffs); free(chars); free(words);
    return (int64_t)pos;
}

int64_t orc_bwt(const uint8_t *dict, uint64_t dsize, const uint64_t *occ, uint64_t dwords,
                const uint8_t *bwlast, const uint64_t *ilist, const uint64_t *bwsai, uint64_t nrows,
                int w, int U, uint8_t *bwt, uint64_t *sa_raw, uint64_t *easy_cases, uint64_t *hard_cases)
{
    return orc_bwt_impl(dict, dsize, occ, dwords, bwlast, ilist, bwsai, nrows, w, U, bwt, sa_raw, easy_cases, hard_cases, NULL, NULL);
}

/* gSA + gLCP of a .dict image exactly as gsacak(s, SA, LCP, NULL, n) returns them (gsa/gsacak.c:2504-2524) */
int orc_gsa_lcp(const uint8_t *dict, uint64_t dsize, uint64_t dwords, uint64_t *gsa, uint64_t *lcp)
{
    uint64_t *occ = (uint64_t *)calloc((size_t)dwords + 1, sizeof(uint64_t));
    int64_t r = orc_bwt_impl(dict, dsize, occ, dwords, NULL, NULL, NULL, 0, 0, 8, NULL, NULL, NULL, NULL, gsa, lcp);
    free(occ);
    return (int)r;
}

/* src/pfbwt-f.cpp:298-320 (out_fn), 325-328 (final run end) */
uint64_t orc_outfn(const uint8_t *bwt, const uint64_t *sa_raw, uint64_t nout, uint64_t n, int U,
                   uint64_t *sa_out, uint64_t *ssa, uint64_t *esa)
{
    uint64_t r = 0, i, pi = 0, psa = 0, ne = 0;
    uint8_t pbwtc = 0;
    const uint64_t umask = (U == 4) ? 0xffffffffULL : UINT64_MAX;
    for (i = 0; i < nout; ++i) {
        uint64_t a_sa = sa_raw ? sa_raw[i] : 0;
        if (sa_out) sa_out[i] = i ? a_sa : (n & umask);            /* :301 */
        if (bwt[i] != pbwtc) {                                     /* run start, :304 */
            if (ssa) { ssa[2 * r] = i; ssa[2 * r + 1] = i ? a_sa : (n & umask); }
            if (esa && i) { esa[2 * ne] = pi; esa[2 * ne + 1] = pi ? psa : (n & umask); ++ne; }
            ++r;
        }
        pi = i; psa = a_sa; pbwtc = bwt[i];
    }
    if (esa && nout) { esa[2 * ne] = pi; esa[2 * ne + 1] = psa; ++ne; }   /* :325-328, raw value */
    return r;
}
