/* Fast seeded corpus stream for benchmarks/tests (see tokengeex_amd/synth.py).
 * Host-only helper, not on the product path: draws lexicon items from a
 * cumulative distribution and cuts samples of log-uniform length at item
 * boundaries.  splitmix64 seeded by the caller. */
#include <math.h>
#include <stdint.h>
#include <string.h>

static inline uint64_t splitmix64(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static inline double u01(uint64_t *s) { return (double)(splitmix64(s) >> 11) * (1.0 / 9007199254740992.0); }

/* Returns the number of samples written (offs[0..n_samples]); text gets exactly
 * offs[n_samples] bytes (<= cap_bytes).  quick[] is a 4096-entry guide table
 * over cdf to shorten the binary search. */
uint64_t synth_fill(const uint8_t *lex_flat, const uint32_t *lex_offs, const double *cdf, uint32_t n_items,
                    uint64_t seed, uint64_t n_bytes, uint32_t min_len, uint32_t max_len, uint8_t *text,
                    uint64_t cap_bytes, uint64_t *offs, uint64_t max_samples) {
    enum { G = 4096 };
    static uint32_t guide[G + 1];
    uint32_t j = 0;
    for (uint32_t g = 0; g <= G; g++) {
        double x = (double)g / G;
        while (j < n_items - 1 && cdf[j] < x) j++;
        guide[g] = j;
    }
    uint64_t rng = seed, pos = 0, ns = 0;
    const double lo = log((double)min_len), hi = log((double)max_len);
    offs[0] = 0;
    while (pos < n_bytes && ns < max_samples) {
        uint64_t target = (uint64_t)exp(lo + (hi - lo) * u01(&rng));
        uint64_t left = n_bytes - pos;
        if (left <= min_len || target > left) target = left;
        uint64_t end = pos + target;
        while (pos < end) {
            double x = u01(&rng);
            uint32_t g = (uint32_t)(x * G);
            uint32_t a = guide[g], b = guide[g + 1];
            while (a < b) { /* first index with cdf > x */
                uint32_t mid = (a + b) >> 1;
                if (cdf[mid] > x) b = mid; else a = mid + 1;
            }
            uint32_t l = lex_offs[a + 1] - lex_offs[a];
            if (pos + l > cap_bytes) { end = pos; n_bytes = pos; break; }
            memcpy(text + pos, lex_flat + lex_offs[a], l);
            pos += l;
        }
        if (pos > offs[ns]) offs[++ns] = pos;
    }
    return ns;
}
