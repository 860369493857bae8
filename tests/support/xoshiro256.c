/* xoshiro256** (Blackman & Vigna 2018, public domain algorithm) seeded through splitmix64 -- the random stream SURVEY.md 8(d) item 4 names for the synthetic
 * KKT system of BASELINE.json configs[3].  Test infrastructure: tests/support/kktgen.py draws the matrix values from it (one stream, consumed in the order
 * kktgen.grid_kkt documents); compiled by tests/support/Makefile into tests/support/lib/libxoshiro256.so. */
#include <stdint.h>
#include <stddef.h>

static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

/* state[4] <- splitmix64 sequence started at `seed` */
void xoshiro256_seed(uint64_t seed, uint64_t* state)
{
    for (int i = 0; i < 4; ++i) {
        uint64_t z = (seed += 0x9e3779b97f4a7c15ULL);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        state[i] = z ^ (z >> 31);
    }
}
/* out[0..n) <- the next n doubles of the stream, uniform in [0, 1): the top 53 bits of each 64-bit output */
void xoshiro256_fill(uint64_t* s, size_t n, double* out)
{
    uint64_t s0 = s[0], s1 = s[1], s2 = s[2], s3 = s[3];
    for (size_t i = 0; i < n; ++i) {
        const uint64_t r = rotl(s1 * 5, 7) * 9;
        const uint64_t t = s1 << 17;
        s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3; s2 ^= t; s3 = rotl(s3, 45);
        out[i] = (double)(r >> 11) * 0x1.0p-53;
    }
    s[0] = s0; s[1] = s1; s[2] = s2; s[3] = s3;
}
