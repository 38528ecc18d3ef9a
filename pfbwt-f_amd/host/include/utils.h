/* pfbwt-f_amd/host/include/utils.h -- what the reference's callers take from include/utils.h + src/utils.c,
 * header-only so that `extern "C" { #include "utils.h" }` (src/pfbwt-f.cpp:12-14, src/merge_pfp.cpp) keeps
 * compiling against this mirror without linking utils.c:
 *   special symbols and file-name extensions   include/utils.h:8-31
 *   die()                                      src/utils.c:13-17   (perror + exit(1))
 *   open_aux_file()                            src/utils.c:32-42   ("base.ext", die on failure)
 *   seq_nt4_table / seq_nt4_ntoa_table         src/utils.c:139-180 (base -> 2-bit code tables)
 * The multi-segment file helpers and get_myint (legacy, unused on the path) are not carried over. */
#ifndef PFBWTF_UTILS_H
#define PFBWTF_UTILS_H
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define Dollar 2
#define EndOfWord 1
#define EndOfDict 0

#define EXTPARSE "parse"
#define EXTPARS0 "parse_old"
#define EXTOCC "occ"
#define EXTDICT "dict"
#define EXTDICZ "dicz"
#define EXTLST "last"
#define EXTBWLST "bwlast"
#define EXTSAI "sai"
#define EXTBWSAI "bwsai"
#define EXTILIST "ilist"
#define EXTSA "sa"
#define EXTSSA "ssa"
#define EXTESA "esa"
#define EXTGSA "gsa"
#define EXTGLCP "glcp"

static inline void die(const char *s)
{
    perror(s);
    exit(1);
}

static inline FILE *open_aux_file(const char *base, const char *ext, const char *mode)
{
    size_t lb = strlen(base), le = strlen(ext);
    char *name = (char *)malloc(lb + le + 2);
    if (name == NULL) die("open_aux_file: malloc");
    memcpy(name, base, lb);
    name[lb] = '.';
    memcpy(name + lb + 1, ext, le + 1);
    FILE *f = fopen(name, mode);
    if (f == NULL) die(name);
    free(name);
    return f;
}

/* code of a base for hashing: A/a/N/n 0, C/c 1, G/g 2, T/t/'-' 3, anything else 5 (rejected, include/hash.hpp:30-31) */
static inline uint8_t pfbwtf_ntoa_code(int c)
{
    switch (c) {
    case 'A': case 'a': case 'N': case 'n': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': case '-': return 3;
    default: return 5;
    }
}
/* code of a base for the non-ACGT test (pfparser.hpp:342-344): A 0, C 1, G 2, T 3, anything else 4 */
static inline uint8_t pfbwtf_nt4_code(int c)
{
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
    }
}
#endif
