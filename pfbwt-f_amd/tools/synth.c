/*
 * pfbwt-f_amd/tools/synth.c -- deterministic synthetic haplotype generator.
 * Host utility used by bench.py and the full-size tests; not on the hot path.
 *
 *   PRNG splitmix64(seed).  For every position i of the base sequence, in order:
 *     base[i] = "ACGT"[next() >> 62]
 *     site    = (next() % 100 == 0)                       (1 % of positions are variant sites)
 *     if site: alt = (base + 1 + next() % 3) % 4 ;  f = ((next() >> 11) * 2^-53)^4   (skewed low)
 *   Haplotype 0 = base.  Haplotype h >= 1 carries alt at site i iff  u(seed, h, i) < f  where
 *   u = (mix64(seed ^ h * 0x9E3779B97F4A7C15 ^ i * 0xD6E8FEB86659FD93) >> 11) * 2^-53.
 *   N-runs (optional): positions [r0, r0+l0) and [r1, r1+l1) are overwritten with 'N' in every haplotype.
 */
#include <stdint.h>
#include <stddef.h>

static uint64_t sm_next(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

/* writes L bytes of haplotype h into out */
void pfp_synth_haplotype(uint64_t seed, uint64_t L, uint64_t h, uint64_t r0, uint64_t l0, uint64_t r1, uint64_t l1, uint8_t *out)
{
    static const char nt[4] = {'A', 'C', 'G', 'T'};
    uint64_t s = seed, i;
    for (i = 0; i < L; ++i) {
        unsigned b = (unsigned)(sm_next(&s) >> 62);
        if (sm_next(&s) % 100 == 0) {
            unsigned alt = (b + 1 + (unsigned)(sm_next(&s) % 3)) % 4;
            double f = (double)(sm_next(&s) >> 11) * (1.0 / 9007199254740992.0);
            f = f * f; f = f * f;
            if (h) {
                double u = (double)(mix64(seed ^ (h * 0x9E3779B97F4A7C15ULL) ^ (i * 0xD6E8FEB86659FD93ULL)) >> 11) * (1.0 / 9007199254740992.0);
                if (u < f) b = alt;
            }
        }
        out[i] = (uint8_t)nt[b];
    }
    for (i = r0; i < r0 + l0 && i < L; ++i) out[i] = 'N';
    for (i = r1; i < r1 + l1 && i < L; ++i) out[i] = 'N';
}
