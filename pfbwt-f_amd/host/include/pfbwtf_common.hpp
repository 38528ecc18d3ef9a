// pfbwt-f_amd/host/include/pfbwtf_common.hpp -- shared bits of the host-side mirror of the reference
// interface: uint_t / int_t / int_text (gsa/gsacak.h:44-64), special symbols and file extensions
// (include/utils.h:8-31), die() (src/utils.c:13-17).  Reading FASTA / FASTQ (include/kseq.h:178-222; gz or plain,
// "-" = stdin) is the engine's job: pfp_parse_feed_fasta_file.
#ifndef PFBWTF_COMMON_HPP
#define PFBWTF_COMMON_HPP
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <zlib.h>
#include "pfbwt_hip.h"
extern "C" {
#include "utils.h"          // special symbols, file extensions, die(), open_aux_file() at global scope like the reference's
}

#ifndef M64
#define M64 0
#endif
#if M64
typedef int64_t int_t;
typedef uint64_t uint_t;
#else
typedef int32_t int_t;
typedef uint32_t uint_t;
#endif
typedef uint32_t int_text;

namespace pfbwtf {

// engine status -> the reference's message + exit(1)
inline void engine_check(pfp_ctx *ctx, int st, const char *what)
{
    if (st == PFP_OK) return;
    if (st == PFP_E_INVALID_CHAR) {             // include/hash.hpp:31
        uint64_t pos = 0; int ch = 0; pfp_error_detail(ctx, &pos, &ch);
        fprintf(stderr, "error, invalid character %d/%c -> %d\n", ch, ch, 5);
    } else if (st == PFP_E_TOO_LARGE) {
        fprintf(stderr, "%s: input too long, please use 64-bit version: %s\n", what, pfp_strerror(st));   // pfparser.hpp:326-331
    } else {
        fprintf(stderr, "%s: %s\n", what, pfp_strerror(st));
    }
    exit(1);
}

template <typename T> inline std::vector<T> read_vec(const std::string &path)
{
    FILE *fp = fopen(path.c_str(), "rb");
    if (fp == NULL) { fprintf(stderr, "%s: ", path.c_str()); die("error opening file"); }
    fseek(fp, 0, SEEK_END); size_t size = (size_t)ftell(fp); rewind(fp);
    std::vector<T> v(size / sizeof(T));
    if (v.size() && fread(v.data(), sizeof(T), v.size(), fp) != v.size()) { fprintf(stderr, "error reading from %s\n", path.c_str()); exit(1); }
    fclose(fp);
    return v;
}

} // namespace pfbwtf
#endif
