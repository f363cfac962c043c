// tools/div_check.c -- brute-force check of the reciprocal form of the gather's division (pg_place.hip, k_gather_wave::conv).
// build: gcc -O2 -mfma -ffp-contract=off -o /tmp/div_check tools/div_check.c -lm ; run: /tmp/div_check <cases> <seed>
// is q1 = fma(r, y, q0), r = fma(-b, q0, a), q0 = a * y, y = 1.0 / b  always the correctly rounded a / b?  (b >= 1 normal, a moderate)
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
#include <string.h>
static inline uint64_t rng(uint64_t *s) { uint64_t x = *s; x ^= x << 13; x ^= x >> 7; x ^= x << 17; return *s = x; }
static inline double mk(uint64_t bits) { double d; memcpy(&d, &bits, 8); return d; }
static inline int check(double a, double b, uint64_t *bad, int verbose) {
    const double y = 1.0 / b, q0 = a * y, r = fma(-b, q0, a), q1 = fma(r, y, q0), ref = a / b;
    if (q1 != ref && !(q1 != q1 && ref != ref)) { if (verbose && *bad < 10) printf("MISMATCH a=%a b=%a q1=%a ref=%a\n", a, b, q1, ref); ++*bad; return 1; }
    return 0;
}
int main(int argc, char **argv) {
    uint64_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 100000000ull, seed = argc > 2 ? strtoull(argv[2], 0, 10) : 1;
    uint64_t s = 0x9E3779B97F4A7C15ull * (seed + 1), bad = 0, tot = 0;
    for (uint64_t i = 0; i < n; ++i) {
        // b: exponent 0..9 (1 <= b < 1024), random significand; sometimes significands with long runs of ones / zeros
        uint64_t mb = rng(&s) & 0xFFFFFFFFFFFFFull, ma = rng(&s) & 0xFFFFFFFFFFFFFull, t = rng(&s);
        if ((t & 7) == 0) mb |= ~0ull >> (12 + (t >> 8) % 40) ;            // low bits all ones
        if ((t & 7) == 1) mb &= ~(~0ull >> (12 + (t >> 8) % 40));           // low bits all zeros
        if ((t & 7) == 2) mb = 0xFFFFFFFFFFFFFull - ((t >> 8) & 0xFF);      // near all ones
        const double b = mk(((uint64_t)(1023 + (t >> 20) % 10) << 52) | mb);
        double a = mk(((uint64_t)(1023 - 40 + (t >> 30) % 50) << 52) | ma);
        if (t & (1ull << 40)) a = -a;
        tot++; check(a, b, &bad, 1);
        // adversarial: a = RN(b * q) for a q, and its neighbours: quotients very close to representable numbers / midpoints
        const double q = mk(((uint64_t)(1023 - 8 + (t >> 44) % 16) << 52) | (rng(&s) & 0xFFFFFFFFFFFFFull));
        const double a2 = b * q;
        tot += 3; check(a2, b, &bad, 1); check(nextafter(a2, 1e300), b, &bad, 1); check(nextafter(a2, -1e300), b, &bad, 1);
        // the domain of the gather: a = x - md with x, md around 50..200 (pA), b = MAD in [1, 60]
        const double x = 40.0 + (double)(rng(&s) >> 11) * (140.0 / 9007199254740992.0), md = 60.0 + (double)(rng(&s) >> 11) * (80.0 / 9007199254740992.0);
        const double bb = 1.0 + (double)(rng(&s) >> 11) * (59.0 / 9007199254740992.0);
        tot++; check(x - md, bb, &bad, 1);
    }
    printf("seed %llu: %llu checks, %llu mismatches\n", (unsigned long long)seed, (unsigned long long)tot, (unsigned long long)bad);
    return bad != 0;
}
