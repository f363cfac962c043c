// pg_select.h -- exact median / MAD of a read in the raw-code domain (shared host + device code).
//
// The reference converts every sample to double, zero-fills the out-of-range ones and takes exact
// order statistics of the doubles (src/gmove.cpp:754-771, 142-184; src/ksort.h:233-259). Here the
// same two values are obtained from a histogram of the int16 codes:
//   * pA(code) = ((double)code + offset) * (range / digitisation)  (src/poregen.h:30) is monotone
//     non-decreasing in code when range/digitisation > 0 (IEEE rounding is monotone), so the in-range
//     codes form one interval [c_lo, c_hi] and the sorted sample vector is
//         [in-range codes with pA < 0, ascending] [0.0 x nZ] [in-range codes with pA >= 0, ascending]
//     where nZ = number of zero-filled samples;
//   * |x - med| over in-range codes is monotone away from the split point on both sides, so the MAD
//     is the k-th value of three monotone sequences (up side, down side, the zero-filled class) and is
//     found by counting with the inclusive prefix sums of the histogram.
// Every double that is compared or returned is produced by exactly the reference's expression, so the
// results are bit-identical to sorted(x)[n/2] and 1.4826*sorted(|x-med|)[n/2].
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define PG_HD __host__ __device__ __forceinline__
#else
#define PG_HD inline
#endif

struct PgReadPlan {
    int32_t c_lo;   // first code with !(pA < pa_min)
    int32_t span;   // number of in-range codes (c_hi - c_lo + 1), 0 if none
    int32_t z0;     // number of in-range codes with pA < 0.0 (index of the first one with pA >= 0.0)
    int32_t status; // 0 ok, -1 range/digitisation not a positive finite number or offset not finite
};

PG_HD double pg_pa(int code, double offset, double scale) { return ((double)code + offset) * scale; }

// first code in [-32768, 32768] (32768 = none) for which pred(code) holds; pred must be monotone
template <class Pred> PG_HD int pg_first_code(Pred pred) {
    int lo = -32768, hi = 32768;
    while (lo < hi) {
        int mid = lo + ((hi - lo) >> 1);
        if (pred(mid)) hi = mid; else lo = mid + 1;
    }
    return lo;
}

PG_HD PgReadPlan pg_make_plan(double digitisation, double offset, double range, double pa_min, double pa_max) {
    PgReadPlan p;
    const double scale = range / digitisation;
    p.status = (scale > 0.0 && isfinite(scale) && isfinite(offset)) ? 0 : -1;
    if (p.status != 0) { p.c_lo = 0; p.span = 0; p.z0 = 0; return p; }
    // zero-fill test of the reference: pA < pa_min || pA > pa_max  (gmove.cpp:756)
    const int c_lo = pg_first_code([&](int c) { return !(pg_pa(c, offset, scale) < pa_min); });
    const int c_gt = pg_first_code([&](int c) { return pg_pa(c, offset, scale) > pa_max; });
    const int c_z = pg_first_code([&](int c) { return pg_pa(c, offset, scale) >= 0.0; });
    p.c_lo = c_lo;
    p.span = c_gt > c_lo ? c_gt - c_lo : 0;
    int z0 = c_z - c_lo;
    p.z0 = z0 < 0 ? 0 : (z0 > p.span ? p.span : z0);
    return p;
}

// pre[b], b in [0, span): number of samples whose code is in [c_lo, c_lo + b] (inclusive prefix).
// L = len_raw_signal. Returns med and mad (already scaled by 1.4826 and clamped to >= 1.0).
struct PgMedMad { double med, mad, mad_raw; }; // mad_raw: sorted(|x-med|)[n/2] before *1.4826 and the clamp

template <class PrePtr>
PG_HD PgMedMad pg_medmad_from_prefix(PrePtr pre, const PgReadPlan &pl, uint64_t L, double offset, double scale) {
    const int span = pl.span, c_lo = pl.c_lo, z0 = pl.z0;
    auto P = [&](int b) -> uint64_t { return (b < 0 || span == 0) ? 0 : (uint64_t)pre[b >= span ? span - 1 : b]; };
    auto first_above = [&](uint64_t j) { // smallest b with pre[b] > j
        int lo = 0, hi = span;
        while (lo < hi) { int mid = (lo + hi) >> 1; if ((uint64_t)pre[mid] > j) hi = mid; else lo = mid + 1; }
        return lo;
    };
    const uint64_t nV = span > 0 ? (uint64_t)pre[span - 1] : 0;
    const uint64_t nZ = L - nV;
    const uint64_t cb = P(z0 - 1);
    const uint64_t k = L / 2; // upper median: ks_ksmall(n, copy, n/2), gmove.cpp:146
    PgMedMad out;
    bool zmed = false;
    int bm = 0;
    if (k < cb) bm = first_above(k);
    else if (k < cb + nZ) zmed = true;
    else bm = first_above(k - nZ);
    const double med = zmed ? 0.0 : pg_pa(c_lo + bm, offset, scale);
    out.med = med;

    double mad_raw;
    if (L == 1) mad_raw = 0.0; // calc_madf, gmove.cpp:166-168
    else {
        // split point: first in-range index whose value is >= med
        int sp;
        if (zmed) sp = z0;
        else { sp = bm; while (sp > 0 && pg_pa(c_lo + sp - 1, offset, scale) >= med) --sp; }
        const int nU = span - sp, nD = sp;
        const uint64_t base = P(sp - 1);
        const double dZ = fabs(0.0 - med);
        const double inv = 1.0 / scale;
        auto U = [&](int t) { return fabs(pg_pa(c_lo + sp + t, offset, scale) - med); };     // t in [0,nU)
        auto D = [&](int t) { return fabs(pg_pa(c_lo + sp - 1 - t, offset, scale) - med); }; // t in [0,nD)
        // number of codes on one side whose deviation is <= v (deviations are monotone in t)
        auto count_leq = [&](bool up, double v) -> int {
            const int n = up ? nU : nD;
            if (n == 0) return 0;
            auto dev = [&](int t) { return up ? U(t) : D(t); };
            const double d0 = dev(0);
            int c;
            if (!(v >= d0)) c = 0;
            else {
                const double est = (v - d0) * inv + 1.0; // deviations are spaced ~scale apart
                c = est >= (double)n ? n : (int)est;
                if (c < 1) c = 1;
            }
            int guard = 0;
            while (c < n && dev(c) <= v) { ++c; if (++guard > 8) break; }
            while (guard <= 8 && c > 0 && dev(c - 1) > v) { --c; if (++guard > 8) break; }
            if (guard > 8) { // spacing assumption failed: plain binary search (first t with dev(t) > v)
                int lo = 0, hi = n;
                while (lo < hi) { int mid = (lo + hi) >> 1; if (dev(mid) > v) hi = mid; else lo = mid + 1; }
                c = lo;
            }
            return c;
        };
        auto CU = [&](int t) -> uint64_t { return t <= 0 ? 0 : P(sp + t - 1) - base; };
        auto CD = [&](int t) -> uint64_t { return t <= 0 ? 0 : base - P(sp - t - 1); };
        auto N = [&](double v) -> uint64_t { // number of samples with |x - med| <= v
            return CU(count_leq(true, v)) + CD(count_leq(false, v)) + (dZ <= v ? nZ : 0);
        };
        const uint64_t need = k + 1; // the k-th (0-based) smallest is the least v with N(v) >= k+1
        double best = INFINITY;
        for (int side = 0; side < 2; ++side) {
            const bool up = side == 0;
            const int n = up ? nU : nD;
            int lo = 0, hi = n; // smallest t with N(dev(t)) >= need
            while (lo < hi) {
                int mid = (lo + hi) >> 1;
                if (N(up ? U(mid) : D(mid)) >= need) hi = mid; else lo = mid + 1;
            }
            if (lo < n) { const double v = up ? U(lo) : D(lo); if (v < best) best = v; }
        }
        if (nZ > 0 && N(dZ) >= need && dZ < best) best = dZ;
        mad_raw = best;
    }
    out.mad_raw = mad_raw;
    double mad = mad_raw * 1.4826;       // gmove.cpp:162,183
    out.mad = (mad > 1.0) ? mad : 1.0;   // gmove.cpp:771
    return out;
}
