// pfbwt-f_amd/host/include/hash.hpp -- host mirror of include/hash.hpp:12-43.  The engine evaluates
// the same function on the device (csrc/parse.h k_trigger_scan); this header exists so code written
// against `WangHash` keeps compiling and so host tests can cross-check the known answers.
#ifndef PFBWTF_HASH_HPP
#define PFBWTF_HASH_HPP
#include <cstdint>
#include <cstdio>
#include <cstdlib>

inline uint64_t wang_hash(uint64_t key)
{
    key = (~key) + (key << 21);
    key ^= key >> 24;
    key = (key + (key << 3)) + (key << 8);
    key ^= key >> 14;
    key = (key + (key << 2)) + (key << 4);
    key ^= key >> 28;
    key += key << 31;
    return key;
}

struct WangHash {
    explicit WangHash(size_t w) : k(w), mask(w >= 32 ? 0 : ((1ULL << (2 * w)) - 1)) {}
    uint64_t update(char c)
    {
        int x;
        switch (c) { case 'A': case 'a': case 'N': case 'n': x = 0; break; case 'C': case 'c': x = 1; break; case 'G': case 'g': x = 2; break;
                     case 'T': case 't': case '-': x = 3; break; default: x = 5; }
        if (x > 3) { fprintf(stderr, "error, invalid character %d/%c -> %d\n", c, c, x); exit(1); }
        kmer = ((kmer << 2) | (uint64_t)x) & mask;
        hash = wang_hash(kmer);
        return hash;
    }
    uint64_t hashvalue() const { return hash; }
    size_t k;
    uint64_t kmer = 0, mask, hash = 0;
};
#endif
