// pg_model.h -- fixed-point view of the kept samples for the per-k-mer model reduction (shared host + device code).
//
// The reference has no code for this step: its pipeline (scripts/poregen.sh:54-85, calculate_mean_stddev_all) reads
// the dump files back as TEXT and pipes them through `tr ';,' '\n' | tail -n +2 | datamash median 1` and
// `... | datamash sstdev 1`. What datamash sees is therefore not the double gmove held but its "%.8f" print-out
// (src/gmove.cpp:941-944): a decimal with eight fractional digits, i.e. an integer number of 1e-8 units. The device
// reduction works on exactly those integers, so the median is exact and the moments are exact integers.
#pragma once
#include <stdint.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#if defined(__HIPCC__)
#define PGM_HD __host__ __device__ __forceinline__
#else
#define PGM_HD inline
#endif

#define PG_MODEL_MAX_ABS 4.0e7            /* |sample| the fixed-point view accepts (|units| < 2^52) */
#define PG_MODEL_LIMB_BITS 20             /* deviations from the slot's first value are split in two 20-bit limbs */
#define PG_MODEL_MAX_DEV (1ll << 40)      /* |value - first value| in 1e-8 units the moment sums accept (~10995 pA) */
#define PG_MODEL_TINY_MAX 1024            /* files up to this many values and fewer than PG_MODEL_TINY_EVENTS events: one wave, 16 values per lane */
#define PG_MODEL_TINY_EVENTS 256          /* dwell fields (events + 1) the one-wave kernel holds in registers */
#define PG_MODEL_MID_MAX 2048             /* ... one wave, 32 values per lane (its own launch over the list of such files) */
#define PG_MODEL_MID_EVENTS 512
#define PG_MODEL_SHORT_MAX 4096           /* up to this many: one 256-thread workgroup, values in registers; beyond: re-read per pass */
enum { PG_MODEL_TINY = 0, PG_MODEL_MID = 1, PG_MODEL_SHORT = 2, PG_MODEL_LONG = 3, PG_MODEL_KINDS = 4 };
PGM_HD int pg_model_kind(uint64_t n_values, uint64_t n_events) {
    return (n_values <= PG_MODEL_TINY_MAX && n_events < PG_MODEL_TINY_EVENTS) ? PG_MODEL_TINY : (n_values <= PG_MODEL_MID_MAX && n_events < PG_MODEL_MID_EVENTS) ? PG_MODEL_MID
           : n_values <= PG_MODEL_SHORT_MAX ? PG_MODEL_SHORT : PG_MODEL_LONG;
}
#define PG_MODEL_MAX_VALUES (1ull << 23)  /* values per slot the moment sums accept (limb^2 sums stay below 2^63) */

// printf("%.8f", x) as an integer number of 1e-8 units: the correctly rounded (ties to even, on the exact binary value,
// like glibc) product x * 10^8. p + e is that product exactly (10^8 is a double; the FMA returns the rounding error).
PGM_HD int64_t pg_fixed8(double x, bool &bad) {
    if (!(fabs(x) < PG_MODEL_MAX_ABS)) { bad = true; return 0; } // also NaN
    const double p = x * 1e8;        // |p| < 2^52: ulp(p) <= 1/2
    const double e = fma(x, 1e8, -p);
    const double r = rint(p);        // ties to even in the default rounding mode
    const double f = p - r;          // exact, a multiple of ulp(p) in [-1/2, 1/2]; |e| <= ulp(p)/2
    int64_t v = (int64_t)r;
    // |f| < 1/2: f + e cannot reach 1/2. |f| = 1/2: p sits on a tie that rint broke to even; e says on which side the
    // exact product lies (e == 0: a true tie, already even).
    if (f == 0.5 && e > 0.0) v += 1;
    else if (f == -0.5 && e < 0.0) v -= 1;
    return v;
}

// One slot's reduction as the kernel leaves it (64 bytes).
struct PgSlotModel {
    uint64_t n;        // values that reach datamash (all samples of the slot but the first: `tail -n +2`)
    int64_t origin;    // fixed8 of the first value that counts; the sums below are over d = value - origin
    int64_t s1;        // sum d
    uint64_t s2_hh, s2_hl, s2_ll; // |d| = h * 2^20 + l:  sum h*h, sum h*l, sum l*l   (sum d^2 = hh*2^40 + hl*2^21 + ll)
    int64_t mid_lo, mid_hi;       // the two middle order statistics (equal when n is odd), 1e-8 units
};
struct PgSlotDwell {
    uint64_t n;        // fields awk sees in the file: kept events + the empty field behind the last ';'
    uint32_t mid_lo, mid_hi; // middle order statistics of (samples in the event - 1) and that one 0
    uint32_t flags;    // PG_MODEL_BAD_*
    uint32_t pad;
};
enum { PG_MODEL_BAD_VALUE = 1, PG_MODEL_BAD_SPREAD = 2, PG_MODEL_BAD_COUNT = 4 };

// Host side. sample standard deviation (datamash sstdev: sqrt(sum (x-mean)^2 / (n-1))) from the exact moments, in 1e-8 units.
// n*sum d^2 - (sum d)^2 is an exact 128-bit integer; one rounding to long double, one division, one square root.
inline long double pg_model_sstdev_units(const PgSlotModel &m) {
    if (m.n < 2) return NAN;
    const unsigned __int128 s2 = ((unsigned __int128)m.s2_hh << 40) + ((unsigned __int128)m.s2_hl << 21) + m.s2_ll;
    const __int128 s1 = m.s1;
    const unsigned __int128 num = (unsigned __int128)m.n * s2 - (unsigned __int128)(s1 * s1);
    const long double den = (long double)m.n * (long double)(m.n - 1);
    return sqrtl((long double)num / den);
}
// The sstdev text, "%.14Lg" of the sample standard deviation, as the CORRECTLY ROUNDED 14 digits of the exact value. The long double above
// is within half a unit of its 64-bit mantissa (19 digits), so its 14 digits are right unless the exact value sits within ~1e-19 (relative)
// of a rounding boundary -- seen once in ~1600 fuzzed model files: exact 2.89847887318104999994..., long double just above the boundary,
// text ...811 instead of ...81. The verdict at the two boundaries next to the printed digits is therefore taken in exact integer arithmetic:
// sd < (2D +- 1)/2 * 10^e  <=>  4 * num * 10^16... against (2D +- 1)^2 * n (n - 1), numbers of up to ~220 bits.
struct PgBig512 { uint64_t w[8]; };
inline void pg_big_set(PgBig512 &b, unsigned __int128 x) { for (int i = 0; i < 8; ++i) b.w[i] = 0; b.w[0] = (uint64_t)x; b.w[1] = (uint64_t)(x >> 64); }
inline void pg_big_mul(PgBig512 &b, uint64_t m) { unsigned __int128 c = 0; for (int i = 0; i < 8; ++i) { const unsigned __int128 t = (unsigned __int128)b.w[i] * m + c; b.w[i] = (uint64_t)t; c = t >> 64; } }
inline void pg_big_mul_pow10(PgBig512 &b, int k) { while (k >= 19) { pg_big_mul(b, 10000000000000000000ull); k -= 19; } uint64_t p = 1; while (k-- > 0) p *= 10; pg_big_mul(b, p); }
inline int pg_big_cmp(const PgBig512 &a, const PgBig512 &b) { for (int i = 7; i >= 0; --i) if (a.w[i] != b.w[i]) return a.w[i] < b.w[i] ? -1 : 1; return 0; }
// sign of  sqrt(num / (den * 10^16)) - (d2 / 2) * 10^e   (the standard deviation in sample units against a decimal boundary)
inline int pg_sstdev_cmp(unsigned __int128 num, uint64_t den, uint64_t d2, int e) {
    PgBig512 L, R;
    pg_big_set(L, num); pg_big_mul(L, 4);
    pg_big_set(R, (unsigned __int128)d2 * d2); pg_big_mul(R, den);
    const int f = 2 * e + 16;
    if (f >= 0) pg_big_mul_pow10(R, f); else pg_big_mul_pow10(L, -f);
    return pg_big_cmp(L, R);
}
// n >= 2 values, num = n * sum d^2 - (sum d)^2 in (1e-8 unit)^2. Returns snprintf's count.
inline int pg_model_sstdev_text(uint64_t n, unsigned __int128 num, char *buf, size_t cap) {
    const uint64_t den = n * (n - 1);
    long double sd = sqrtl((long double)num / ((long double)n * (long double)(n - 1))) / 1e8L;
    char t[64];
    snprintf(t, sizeof t, "%.13Le", sd); // d.ddddddddddddde+XX: the 14 digits "%.14Lg" prints
    if (num != 0 && t[1] == '.') {
        uint64_t D = (uint64_t)(t[0] - '0');
        int i = 2;
        for (; t[i] >= '0' && t[i] <= '9'; ++i) D = D * 10 + (uint64_t)(t[i] - '0');
        if (i == 15 && t[i] == 'e') {
            const int e = atoi(t + i + 1) - 13; // sd ~ D * 10^e
            if (e >= -40 && e <= 8) { // (in range of the 512-bit products)
                uint64_t D1 = D;
                const int up = pg_sstdev_cmp(num, den, 2 * D + 1, e);
                if (up > 0 || (up == 0 && (D & 1))) D1 = D + 1;                         // at or above the upper boundary (ties to even)
                else { const int lo = pg_sstdev_cmp(num, den, 2 * D - 1, e); if (lo < 0 || (lo == 0 && (D & 1))) D1 = D - 1; }
                if (D1 != D && D1 >= 10000000000000ull && D1 <= 100000000000000ull) {
                    char dec[64];
                    snprintf(dec, sizeof dec, "%llue%d", (unsigned long long)D1, e);
                    sd = strtold(dec, nullptr); // 14 digits: a long double holds them (19), "%.14Lg" prints them back
                }
            }
        }
    }
    return snprintf(buf, cap, "%.14Lg", sd);
}
// datamash median of the decimal texts: each text becomes the long double nearest to units/10^8 (strtold), the two
// middle ones are averaged in long double.
inline long double pg_model_median(const PgSlotModel &m) {
    const long double a = (long double)m.mid_lo / 1e8L, b = (long double)m.mid_hi / 1e8L;
    return m.mid_lo == m.mid_hi ? a : (a + b) / 2.0L;
}
