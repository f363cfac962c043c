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

// (x - median) / MAD of src/gmove.cpp:774 with the reciprocal y = 1.0 / b taken ONCE per event (b = the read's MAD >= 1.0) instead of a
// division per sample: q0 = a * y, r = a - b * q0 (one FMA), q = q0 + r * y (one FMA). q is the correctly rounded a / b -- the
// argument is DESIGN.md section 6 ("the gather's division"), the constructed cases tools/div_check.c and tests/test_select_host.py --
// for operands inside pg_div_domain_ok() below; outside it the dense gathers use the division itself.
PG_HD double pg_div_by_recip(double a, double b, double y) {
    const double q0 = a * y, r = __builtin_fma(-b, q0, a);
    return __builtin_fma(r, y, q0);
}
// The proof needs normal numbers end to end: no subnormal quotient, no inexact residual. With a read's scale = range / digitisation and
// |offset| inside [2^-200, 2^200] (or offset == 0): raw + offset is 0 or a multiple of 2^-252 below 2^201, so every non-zero pA of the
// read has magnitude 2^-452 .. 2^401, every non-zero difference x - median is >= 2^-504 (a multiple of the smaller operand's ulp), the
// MAD is < 2^403, a quotient is >= 2^-907 and the residual a - b * q0 is 0 or >= 2^-1012: normal numbers throughout. Every real
// calibration is inside (scale ~ 0.1 .. 1, offset ~ -300 .. 30); a read outside is still computed exactly, with the division itself.
PG_HD bool pg_div_domain_ok(double offset, double scale) {
    const double lo = 0x1p-200, hi = 0x1p200, ao = offset < 0.0 ? -offset : offset;
    return scale >= lo && scale <= hi && (ao == 0.0 || (ao >= lo && ao <= hi));
}

// first code in [-32768, 32768] (32768 = none) for which pred(code) holds; pred must be monotone
template <class Pred> PG_HD int pg_first_code(Pred pred) {
    int lo = -32768, hi = 32768;
    while (lo < hi) {
        int mid = lo + ((hi - lo) >> 1);
        if (pred(mid)) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// the same with an estimate x of the answer (pA is linear in the code up to rounding): a few evaluations next to the
// estimate instead of 17; any estimate is safe, a bad one only costs the plain binary search
template <class Pred> PG_HD int pg_first_code_near(Pred pred, double x) {
    if (!(x > -32768.0 && x < 32768.0)) return pg_first_code(pred); // also NaN
    int g = (int)x;
    for (int step = 0; step < 4; ++step) {
        if (pred(g)) { if (g == -32768 || !pred(g - 1)) return g; --g; }
        else { if (g == 32767) return 32768; ++g; if (pred(g)) return g; }
    }
    return pg_first_code(pred);
}

PG_HD PgReadPlan pg_make_plan(double digitisation, double offset, double range, double pa_min, double pa_max) {
    PgReadPlan p;
    const double scale = range / digitisation;
    p.status = (scale > 0.0 && isfinite(scale) && isfinite(offset)) ? 0 : -1;
    if (p.status != 0) { p.c_lo = 0; p.span = 0; p.z0 = 0; return p; }
    // zero-fill test of the reference: pA < pa_min || pA > pa_max  (gmove.cpp:756)
    const int c_lo = pg_first_code_near([&](int c) { return !(pg_pa(c, offset, scale) < pa_min); }, ceil(pa_min / scale - offset));
    const int c_gt = pg_first_code_near([&](int c) { return pg_pa(c, offset, scale) > pa_max; }, floor(pa_max / scale - offset) + 1.0);
    const int c_z = pg_first_code_near([&](int c) { return pg_pa(c, offset, scale) >= 0.0; }, ceil(-offset));
    p.c_lo = c_lo;
    p.span = c_gt > c_lo ? c_gt - c_lo : 0;
    int z0 = c_z - c_lo;
    p.z0 = z0 < 0 ? 0 : (z0 > p.span ? p.span : z0);
    return p;
}

// pre[b], b in [0, span): number of samples whose code is in [c_lo, c_lo + b] (inclusive prefix).
// L = len_raw_signal. Returns med and mad (already scaled by 1.4826 and clamped to >= 1.0).
struct PgMedMad { double med, mad, mad_raw; }; // mad_raw: sorted(|x-med|)[n/2] before *1.4826 and the clamp

// The selection as a small state machine so that the same arithmetic can be driven by a scalar binary
// search (host tests, pg_medmad_from_prefix below) or by a 64-lane search (k_read_stats):
//   begin() -> rank of the median;  med_pred(b) is monotone in b;  set_median(bm);
//   begin_mad();  mad_pred(up, t) is monotone in t for each side;  candidate(up, t), z_candidate().
template <class PrePtr>
struct PgSel {
    PrePtr pre;
    int span, c_lo, z0;
    uint64_t L; // < 2^31 (enforced by the callers), so sample counts fit 32 bits
    double offset, scale;
    // begin()
    uint32_t nV, nZ, cb, k, jmed;
    bool zmed;
    // set_median()
    double med;
    // begin_mad()
    int sp, nU, nD;
    uint32_t base, need;
    double dZ, inv;

    PG_HD uint32_t P(int b) const { return (b < 0 || span == 0) ? 0u : (uint32_t)pre[b >= span ? span - 1 : b]; }

    PG_HD void begin() {
        nV = span > 0 ? (uint32_t)pre[span - 1] : 0u;
        nZ = (uint32_t)L - nV;
        cb = P(z0 - 1);
        k = (uint32_t)(L / 2); // upper median: ks_ksmall(n, copy, n/2), gmove.cpp:146
        zmed = false; jmed = 0;
        if (k < cb) jmed = k;
        else if (k < cb + nZ) zmed = true;
        else jmed = k - nZ;
    }
    // smallest b with med_pred(b) is the bin of the median (only when !zmed)
    PG_HD bool med_pred(int b) const { return (uint32_t)pre[b] > jmed; }
    PG_HD void set_median(int bm) { med = zmed ? 0.0 : pg_pa(c_lo + bm, offset, scale); sp = zmed ? z0 : bm; }

    PG_HD void begin_mad() {
        // split point: first in-range index whose value is >= med
        if (!zmed) while (sp > 0 && pg_pa(c_lo + sp - 1, offset, scale) >= med) --sp;
        nU = span - sp; nD = sp;
        base = P(sp - 1);
        dZ = fabs(0.0 - med);
        inv = 1.0 / scale;
        need = k + 1; // the k-th (0-based) smallest deviation is the least v with N(v) >= k+1
    }
    PG_HD double U(int t) const { return fabs(pg_pa(c_lo + sp + t, offset, scale) - med); }     // t in [0,nU)
    PG_HD double D(int t) const { return fabs(pg_pa(c_lo + sp - 1 - t, offset, scale) - med); } // t in [0,nD)
    PG_HD double dev(bool up, int t) const { return up ? U(t) : D(t); }
    // number of codes on one side whose deviation is <= v (deviations are monotone in t). `hint` >= 0 says
    // "code hint-1 is known to be <= v" (the candidate's own side), which skips the estimate.
    PG_HD int count_leq(bool up, double v, int hint = -1) const {
        const int n = up ? nU : nD;
        if (n == 0) return 0;
        int c;
        if (hint >= 0) c = hint;
        else {
            const double d0 = dev(up, 0);
            if (!(v >= d0)) c = 0;
            else {
                const double est = (v - d0) * inv + 1.0; // deviations are spaced ~scale apart
                c = est >= (double)n ? n : (int)est;
                if (c < 1) c = 1;
            }
        }
        int guard = 0;
        while (c < n && dev(up, c) <= v) { ++c; if (++guard > 8) break; }
        if (hint < 0) // with a hint, code hint-1 is known to qualify: nothing to walk back over
            while (guard <= 8 && c > 0 && dev(up, c - 1) > v) { --c; if (++guard > 8) break; }
        if (guard > 8) { // spacing assumption failed: plain binary search (first t with dev(t) > v)
            int lo = 0, hi = n;
            while (lo < hi) { int mid = (lo + hi) >> 1; if (dev(up, mid) > v) hi = mid; else lo = mid + 1; }
            c = lo;
        }
        return c;
    }
    // count_leq when a guess m of the answer is at hand: three evaluations, no loop. The result is b + [dev(b) <= v] for
    // b = clamp(m, -1, n-1); it is exact iff dev(b-1) <= v < dev(b+1) (deviations are monotone in t), which `ok` reports.
    // Index -1 counts as "<= v", index n as "> v".
    PG_HD int count_chk(bool up, double v, int m, bool &ok) const {
        const int n = up ? nU : nD;
        const int b = m < -1 ? -1 : (m > n - 1 ? n - 1 : m);
        const int hi = n > 0 ? n - 1 : 0;
        auto le = [&](int x) {
            const int xc = x < 0 ? 0 : (x > hi ? hi : x);
            const bool r = dev(up, xc) <= v;
            return x < 0 ? true : (x >= n ? false : r);
        };
        const bool ea = le(b - 1), eb = le(b), ec = le(b + 1);
        ok = ea && !ec;
        return b + (eb ? 1 : 0);
    }
    PG_HD uint32_t CU(int t) const { return t <= 0 ? 0u : P(sp + t - 1) - base; }
    PG_HD uint32_t CD(int t) const { return t <= 0 ? 0u : base - P(sp - t - 1); }
    // number of samples with |x - med| <= v
    PG_HD uint32_t N(double v, int hintU = -1, int hintD = -1) const {
        return CU(count_leq(true, v, hintU)) + CD(count_leq(false, v, hintD)) + (dZ <= v ? nZ : 0u);
    }
    // monotone in t: does the t-th code of this side already cover the k-th smallest deviation?
    PG_HD bool mad_pred(bool up, int t) const {
        const double v = dev(up, t);
        return N(v, up ? t + 1 : -1, up ? -1 : t + 1) >= need;
    }
    PG_HD bool z_ok() const { return nZ > 0 && N(dZ) >= need; }

    // ---- integer-only approximation of mad_pred, used to NARROW the search (never to decide it) -------------
    // deviations on both sides are spaced ~scale apart: U(t) ~ u0 + t*scale, D(t) ~ d0 + t*scale
    int aShiftU, aShiftD, aZU, aZD; // other-side code count offset for a U / D candidate; first t whose value reaches dZ
    PG_HD void begin_approx() {
        const double u0 = nU > 0 ? U(0) : 0.0, d0 = nD > 0 ? D(0) : 0.0;
        const double q = (u0 - d0) * inv;
        aShiftU = (int)floor(q) + 1;   // D codes with d0 + t'*s <= u0 + t*s  <=>  t' <= t + q
        aShiftD = (int)floor(-q) + 1;
        const double zu = (dZ - u0) * inv, zd = (dZ - d0) * inv;
        aZU = zu <= 0.0 ? 0 : (zu >= 65536.0 ? 65536 : (int)ceil(zu));
        aZD = zd <= 0.0 ? 0 : (zd >= 65536.0 ? 65536 : (int)ceil(zd));
    }
    PG_HD bool approx_pred(bool up, int t) const {
        int o = t + (up ? aShiftU : aShiftD);
        const int no = up ? nD : nU;
        o = o < 0 ? 0 : (o > no ? no : o);
        const uint32_t own = up ? CU(t + 1) : CD(t + 1), other = up ? CD(o) : CU(o);
        return own + other + (t >= (up ? aZU : aZD) ? nZ : 0u) >= need;
    }
    PG_HD PgMedMad finish(double best) const {
        PgMedMad out;
        out.med = med;
        out.mad_raw = (L == 1) ? 0.0 : best; // calc_madf, gmove.cpp:166-168
        const double mad = out.mad_raw * 1.4826; // gmove.cpp:162,183
        out.mad = (mad > 1.0) ? mad : 1.0;       // gmove.cpp:771
        return out;
    }
};

// ---- the symmetric path: the usual read in ~1/5 of the instructions of the general search ---------------------------
// When the median is a sample (not the zero-filled class) it is the value of one code bin m: med == pA(c_lo + m) bit for
// bit, and the deviation of code m +- t is t * scale up to a few ulps. With scale well above those ulps (pg_sym_guard) the
// deviations of ring t (codes m - t and m + t) lie strictly between those of rings t - 1 and t + 1, so sorted(|x - med|)
// is: ring 0, ring 1, ... with only the ORDER INSIDE a ring left to rounding. Hence with W(t) = samples in bins
// [m - t, m + t] (two prefix look-ups), t* = least t with W(t) >= k + 1, the k-th smallest deviation is one of the (at
// most) two exact values of ring t*: the smaller one a if W(t* - 1) + count(a) >= k + 1, else the larger one. The
// zero-filled class (deviation dZ = |0 - med|) must lie strictly above ring t*, else the caller takes the general path.
// Every double compared or returned is still produced by the reference's expressions (dev()).
PG_HD bool pg_sym_guard(double offset, double scale, double pa_first, double pa_last) {
    // ring separation scale*(1 - 2^-11) must dominate the rounding of (code + offset) [<= 2^-13 for |.| < 2^40], of the
    // product and of the difference [a few ulps of the largest in-range |pA|]
    const double pmax = fabs(pa_first) > fabs(pa_last) ? fabs(pa_first) : fabs(pa_last);
    return fabs(offset) < 1099511627776.0 && scale * 1099511627776.0 > pmax + 1.0; // NaN fails both
}
template <class PrePtr> struct PgSym {
    const PgSel<PrePtr> *s;
    int m; // bin of the median
    PG_HD uint32_t W(int t) const { return s->P(m + t) - s->P(m - t - 1); } // P clamps on both sides
    PG_HD int t_max() const { const int a = m, b = s->span - 1 - m; return a > b ? a : b; }
    // the decision once t* is known (W(ts) >= need > W(ts - 1)); false = the zero-filled class interferes
    PG_HD bool decide(int ts, double &best) const {
        const bool has_up = m + ts < s->span, has_dn = ts > 0 && m - ts >= 0;
        const double du = has_up ? fabs(pg_pa(s->c_lo + m + ts, s->offset, s->scale) - s->med) : 0.0;
        const double dd = has_dn ? fabs(pg_pa(s->c_lo + m - ts, s->offset, s->scale) - s->med) : 0.0;
        const uint32_t hu = has_up ? s->P(m + ts) - s->P(m + ts - 1) : 0u, hd = has_dn ? s->P(m - ts) - s->P(m - ts - 1) : 0u;
        const uint32_t below = ts > 0 ? W(ts - 1) : 0u;
        double lo, hi; uint32_t hlo;
        if (has_up && has_dn) { const bool up_first = du <= dd; lo = up_first ? du : dd; hi = up_first ? dd : du; hlo = up_first ? hu : hd; if (du == dd) hlo = hu + hd; }
        else { lo = hi = has_up ? du : dd; hlo = hu + hd; }
        if (s->nZ > 0 && !(s->dZ > hi)) return false;
        best = below + hlo >= s->need ? lo : hi;
        return true;
    }
};
// scalar driver of the symmetric path (host tests): false = not applicable, take pg_medmad_from_prefix
template <class PrePtr>
PG_HD bool pg_medmad_sym(PrePtr pre, const PgReadPlan &pl, uint64_t L, double offset, double scale, PgMedMad &out) {
    PgSel<PrePtr> s;
    s.pre = pre; s.span = pl.span; s.c_lo = pl.c_lo; s.z0 = pl.z0; s.L = L; s.offset = offset; s.scale = scale;
    s.begin();
    if (s.zmed || s.span <= 0) return false;
    if (!pg_sym_guard(offset, scale, pg_pa(pl.c_lo, offset, scale), pg_pa(pl.c_lo + pl.span - 1, offset, scale))) return false;
    int bm; { int lo = 0, hi = s.span; while (lo < hi) { int mid = (lo + hi) >> 1; if (s.med_pred(mid)) hi = mid; else lo = mid + 1; } bm = lo; }
    s.set_median(bm);
    double best = 0.0;
    if (L > 1) {
        s.dZ = fabs(0.0 - s.med); s.need = s.k + 1;
        PgSym<PrePtr> y{&s, bm};
        const int tm = y.t_max();
        if (y.W(tm) < s.need) return false; // the in-range samples alone do not reach the rank: the zero-filled class decides
        int lo = 0, hi = tm; // least t with W(t) >= need
        while (lo < hi) { int mid = (lo + hi) >> 1; if (y.W(mid) >= s.need) hi = mid; else lo = mid + 1; }
        if (!y.decide(lo, best)) return false;
    }
    out = s.finish(best);
    return true;
}

// scalar driver (binary searches): host tests and the reference point for the 64-lane driver
template <class PrePtr>
PG_HD PgMedMad pg_medmad_from_prefix(PrePtr pre, const PgReadPlan &pl, uint64_t L, double offset, double scale) {
    PgSel<PrePtr> s;
    s.pre = pre; s.span = pl.span; s.c_lo = pl.c_lo; s.z0 = pl.z0; s.L = L; s.offset = offset; s.scale = scale;
    s.begin();
    int bm = 0;
    if (!s.zmed) { int lo = 0, hi = s.span; while (lo < hi) { int mid = (lo + hi) >> 1; if (s.med_pred(mid)) hi = mid; else lo = mid + 1; } bm = lo; }
    s.set_median(bm);
    double best = INFINITY;
    if (L > 1) {
        s.begin_mad();
        for (int side = 0; side < 2; ++side) {
            const bool up = side == 0;
            const int n = up ? s.nU : s.nD;
            int lo = 0, hi = n; // smallest t with N(dev(t)) >= need
            while (lo < hi) { int mid = (lo + hi) >> 1; if (s.mad_pred(up, mid)) hi = mid; else lo = mid + 1; }
            if (lo < n) { const double v = s.dev(up, lo); if (v < best) best = v; }
        }
        if (s.z_ok() && s.dZ < best) best = s.dZ;
    }
    return s.finish(best);
}
