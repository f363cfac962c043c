// pg_model.hip -- per-k-mer model reduction over the kept samples: exact median, exact moments, dwell median.
//
// Replaces the text round trip of the reference's pipeline (scripts/poregen.sh:54-85 calculate_mean_stddev_all and
// :33-52 calculate_dwell_times_medians): dump files -> tr/tail/awk -> datamash median / sstdev. See pg_model.h for
// why the arithmetic is done on integers of 1e-8 units.
//
// A k-mer's file is one contiguous range of the kept-sample array. Four kernels by file size (pg_model.h: pg_model_kind):
//   k_slot_model_wave       one WAVE per file (<= 1024 values, < 256 events; k = 9: 259 k of 262 k files), values in registers; runs over all
//                           slots and lists the others by kind
//   k_slot_model_wave_mid   the same code with 32 rows (<= 2048 values, < 512 events), over its list
//   k_slot_model<256, SHORT> one workgroup per file (<= 4096 values), values in registers; also every file the wave kernels hand on
//   k_slot_model<1024, LONG> one workgroup per file, the file read (2 + ceil(bits/12)) times (+1 when the count is even): min/max + moments,
//                           then a radix select that only walks the bits in which the values differ (36 bits for med-MAD scaled data: 3 passes),
//                           then the upper middle value; the file stays in L2 / MALL between the passes (150 k values = 1.2 MB at sample_limit 5000)
#include "pg_internal.h"
#include "pg_model.h"

#ifndef PG_MODEL_TINY_WAVES
#define PG_MODEL_TINY_WAVES 6 // waves per SIMD the one-wave kernel is compiled for: 80 registers and 20 bytes of scratch; k = 9 0.586 (5 waves, 85 registers) -> 0.550 ms, 8 waves (64 registers, 96 bytes of scratch) the same
#endif
namespace {

// loads in flight per thread: the 1024-thread variant must stay within 64 VGPRs (two workgroups per CU), its memory
// parallelism comes from 32 waves per CU; the 256-thread variant serves short files, where the round trips are the cost
template <int MT> struct ModelCfg {
    static constexpr int BITS = 12;    // window of the radix select
    static constexpr int BINS = 1 << BITS;
    static constexpr int U = MT >= 1024 ? 2 : 8;
    static constexpr int CACHE = MT >= 1024 ? 0 : 16; // keys a thread keeps in registers when the whole file fits (MT * CACHE values)
};

template <int MT> struct ModelSmem {
    static constexpr int CAND = MT * 4; // keys of one top-window bin that a long file parks in LDS (see block_middle_long)
    uint64_t cand[CAND];
    uint32_t n_cand;
    uint32_t hist[ModelCfg<MT>::BINS];
    uint64_t red[MT / 64];
    uint32_t wsum[MT / 64];
    uint32_t found_bin, found_below;
};

__device__ __forceinline__ uint64_t op_sum(uint64_t a, uint64_t b) { return a + b; }
__device__ __forceinline__ uint64_t op_min(uint64_t a, uint64_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint64_t op_max(uint64_t a, uint64_t b) { return a > b ? a : b; }

__device__ __forceinline__ uint32_t uniform32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uniform64(uint64_t v) { return ((uint64_t)uniform32((uint32_t)(v >> 32)) << 32) | uniform32((uint32_t)v); }

// every thread gets the reduction of v over the block (in scalar registers: the value is the same in every lane)
template <int MT, class Op> __device__ uint64_t block_reduce(ModelSmem<MT> &sm, uint64_t v, Op op) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = op(v, (uint64_t)__shfl_xor((unsigned long long)v, o, 64));
    __syncthreads(); // red[] of the previous call has been read by everyone
    if ((threadIdx.x & 63) == 0) sm.red[threadIdx.x >> 6] = v;
    __syncthreads();
    uint64_t r = sm.red[0];
#pragma unroll
    for (int w = 1; w < MT / 64; ++w) r = op(r, sm.red[w]);
    return uniform64(r);
}

// bin b < nbins with below(b) <= rank < below(b) + hist[b]; the total of hist[0..nbins) must exceed rank
template <int MT> __device__ void block_find_bin(ModelSmem<MT> &sm, uint32_t rank, uint32_t nbins, uint32_t &bin, uint32_t &below) {
    constexpr int PER = ModelCfg<MT>::BINS / MT;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    __syncthreads(); // histogram complete
    uint32_t h[PER], s = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) { h[i] = (uint32_t)(t * PER + i) < nbins ? sm.hist[t * PER + i] : 0u; s += h[i]; }
    uint32_t inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)inc, o, 64); if (lane >= o) inc += u; }
    if (lane == 63) sm.wsum[w] = inc;
    __syncthreads();
    uint32_t base = inc - s;
    for (int i = 0; i < w; ++i) base += sm.wsum[i];
    if (rank >= base && rank < base + s) { // exactly one thread
        uint32_t b = base;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            if (rank >= b && rank < b + h[i]) { sm.found_bin = (uint32_t)(t * PER + i); sm.found_below = b; }
            b += h[i];
        }
    }
    __syncthreads();
    bin = uniform32(sm.found_bin); below = uniform32(sm.found_below);
}

// fn(key_at(i)) for i in [0, n), this thread's share: U loads are issued before the first use, so that a pass costs
// n / (MT * U) memory round trips instead of n / MT (the LDS atomics in fn keep the compiler from hoisting loads itself)
template <int MT, int U, class T, class At, class Fn> __device__ __forceinline__ void for_each_mine(uint64_t n, At at, Fn fn) {
    for (uint64_t base = 0; base < n; base += (uint64_t)MT * U) {
        T k[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const uint64_t i = base + (uint64_t)u * MT + threadIdx.x; ok[u] = i < n; k[u] = at(ok[u] ? i : n - 1); }
#pragma unroll
        for (int u = 0; u < U; ++u) if (ok[u]) fn(k[u]);
    }
}

__device__ __forceinline__ uint64_t shr64(uint64_t v, int s) { return s >= 64 ? 0ull : v >> s; }

// A source of keys hands every thread its share: each(fn) calls fn(key).
template <int MT, class At> struct GlobalKeys { // re-read (L2 / MALL) and re-converted on every pass
    uint64_t n; At at;
    template <class Fn> __device__ __forceinline__ void each(Fn fn) const { for_each_mine<MT, ModelCfg<MT>::U, uint64_t>(n, at, fn); }
};
template <int C> struct RegKeys { // converted once, kept in registers: element u of thread t is value u * MT + t
    uint64_t k[C > 0 ? C : 1]; int cnt; uint64_t mn;
    template <class Fn> __device__ __forceinline__ void each(Fn fn) const {
#pragma unroll
        for (int u = 0; u < C; ++u) if (u < cnt) fn(k[u] - mn);
    }
};

// One window of the radix select: histogram of bits [hi - wbits, hi) of the keys whose bits above hi equal prefix;
// returns the bin holding `rank` and the number of such keys in the bins below it. sm.hist stays valid until the next call.
template <int MT, class Src> __device__ void block_window(ModelSmem<MT> &sm, const Src &src, int hi, int wbits, uint64_t prefix, uint32_t rank,
                                                           uint32_t &bin, uint32_t &below) {
    const int shift = hi - wbits;
    const uint32_t nbins = 1u << wbits;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nbins; i += MT) sm.hist[i] = 0;
    __syncthreads();
    src.each([&](uint64_t k) {
        if (shr64(k, hi) == prefix) atomicAdd(&sm.hist[(uint32_t)(k >> shift) & (nbins - 1)], 1u);
    });
    block_find_bin<MT>(sm, rank, nbins, bin, below);
}

// the rank-th smallest (0-based) of the source's keys; all keys < 2^bits. The bits are consumed from the top in windows of
// at most ModelCfg<MT>::BITS; the last (lowest) window is the short one, so that a narrow key range costs a small histogram.
template <int MT, class Src> __device__ uint64_t block_select(ModelSmem<MT> &sm, const Src &src, uint64_t rank, int bits) {
    uint64_t prefix = 0; // the bits above the current window
    int hi = bits;       // bits [hi, 64) are settled
    while (hi > 0) {
        constexpr int M_BITS = ModelCfg<MT>::BITS;
        const int wbits = hi >= M_BITS ? M_BITS : hi;
        uint32_t bin, below;
        block_window<MT>(sm, src, hi, wbits, prefix, (uint32_t)rank, bin, below);
        rank -= below;
        prefix = (prefix << wbits) | bin;
        hi -= wbits;
    }
    return prefix;
}

// order statistics r_lo and r_hi (r_hi = r_lo or r_lo + 1) of the source's keys
template <int MT, class Src> __device__ void block_pair(ModelSmem<MT> &sm, const Src &src, uint64_t r_lo, uint64_t r_hi, int bits, uint64_t &k_lo, uint64_t &k_hi) {
    k_lo = block_select<MT>(sm, src, r_lo, bits);
    k_hi = k_lo;
    if (r_hi != r_lo) { // the next order statistic is k_lo again or the smallest key above it
        uint64_t le = 0, gt = ~0ull;
        src.each([&](uint64_t k) { if (k <= k_lo) le++; else gt = op_min(gt, k); });
        le = block_reduce<MT>(sm, le, op_sum); gt = block_reduce<MT>(sm, gt, op_min);
        if (le <= r_hi) k_hi = gt;
    }
}
// datamash's two middle order statistics of n keys (equal when n is odd)
template <int MT, class Src> __device__ void block_middle(ModelSmem<MT> &sm, const Src &src, uint64_t n, int bits, uint64_t &k_lo, uint64_t &k_hi) {
    block_pair<MT>(sm, src, (n - 1) / 2, n / 2, bits, k_lo, k_hi);
}

template <int MT> struct LdsKeys { // the parked candidates (low bits of the keys of one top-window bin), in no particular order
    const uint64_t *k; uint32_t n;
    template <class Fn> __device__ __forceinline__ void each(Fn fn) const { for (uint32_t i = threadIdx.x; i < n; i += MT) fn(k[i]); }
};

// The same for a file that is re-read from memory on every pass: after the top window the median's bin holds a tiny part
// of the file (a 12-bit window over a bell curve: ~0.1 %), so the second pass over the file parks that bin's keys in LDS
// (and notes the smallest key above the bin, the upper middle value if the bin ends exactly at the lower one); the remaining
// windows and the even-count step run on LDS. Three passes over the file (moments, top window, park) instead of 2 + windows + 1.
template <int MT, class Src> __device__ void block_middle_long(ModelSmem<MT> &sm, const Src &src, uint64_t n, int bits, uint64_t &k_lo, uint64_t &k_hi) {
    const uint64_t r_lo = (n - 1) / 2, r_hi = n / 2;
    constexpr int M_BITS = ModelCfg<MT>::BITS;
    if (bits <= M_BITS) { block_pair<MT>(sm, src, r_lo, r_hi, bits, k_lo, k_hi); return; }
    const int low = bits - M_BITS; // bits below the top window
    uint32_t bin, below;
    block_window<MT>(sm, src, bits, M_BITS, 0, (uint32_t)r_lo, bin, below);
    const uint32_t cnt = uniform32(sm.hist[bin]);
    if (cnt > (uint32_t)ModelSmem<MT>::CAND) { block_pair<MT>(sm, src, r_lo, r_hi, bits, k_lo, k_hi); return; } // a spike of equal-ish values: generic path
    if (threadIdx.x == 0) sm.n_cand = 0;
    __syncthreads();
    uint64_t above = ~0ull;
    const uint64_t low_mask = (1ull << low) - 1;
    src.each([&](uint64_t k) {
        const uint64_t top = k >> low;
        if (top == bin) sm.cand[atomicAdd(&sm.n_cand, 1u)] = k & low_mask;
        else if (top > bin) above = op_min(above, k);
    });
    above = block_reduce<MT>(sm, above, op_min); // (its barriers also publish cand[])
    const LdsKeys<MT> lk{sm.cand, cnt};
    const uint64_t q_lo = r_lo - below, q_hi = r_hi - below; // ranks inside the bin; q_lo < cnt
    uint64_t c_lo, c_hi;
    block_pair<MT>(sm, lk, q_lo, q_hi < cnt ? q_hi : q_lo, low, c_lo, c_hi);
    k_lo = ((uint64_t)bin << low) | c_lo;
    k_hi = q_hi < cnt ? (((uint64_t)bin << low) | c_hi) : above;
}

// Where a slot's values and events lie: two dependent rounds of uniform (scalar) loads. Round 5 first put a record pass in front of the
// kernels (one 32-byte load instead) -- 50 us of scattered offset loads at k = 9 that the one-wave kernel, bound by its arithmetic,
// hides for nothing; the kernel over all slots now classifies on the way and lists the slots that belong to the others.
struct __attribute__((aligned(16))) FileRef { uint64_t first, n, e0; uint32_t nev, slot; }; // 32 bytes: also the entry of the per-kind lists (one load in the kernels that stride over them)
__device__ __forceinline__ FileRef file_ref(const uint64_t *__restrict__ ev_off, const uint64_t *__restrict__ samp_off, uint32_t s, uint32_t drop_first) {
    const uint64_t e0 = ev_off[s], e1 = ev_off[s + 1];
    const uint64_t a0 = samp_off[e0], a1 = samp_off[e1];
    const uint64_t skip = (a1 > a0 && drop_first) ? 1 : 0; // `tail -n +2`: the file's first value never reaches datamash
    FileRef f; f.first = a0 + skip; f.n = a1 - f.first; f.e0 = e0; f.nev = (uint32_t)(e1 - e0); f.slot = s;
    return f;
}

// pg_fixed8 for the device: below 2.2e7 the correctly rounded product sits in the low mantissa bits of p + 1.5 * 2^52 (|p| < 2^51; the sum
// rounds to even at an ulp of 1 exactly as rint does), the FMA's error decides an exact tie as in pg_fixed8 -- the same integer, without the
// conversion sequence (a dozen instructions per value and pass of the kernels below); anything else takes the shared function.
__device__ __forceinline__ int64_t fixed8_dev(double x, bool &bad) {
    if (!(fabs(x) < 2.2e7)) return pg_fixed8(x, bad);
    const double p = x * 1e8, e = fma(x, 1e8, -p);
    double t = p + 6755399441055744.0;
    const double f = p - (t - 6755399441055744.0);
    if (f == 0.5 && e > 0.0) t += 1.0;
    else if (f == -0.5 && e < 0.0) t -= 1.0;
    return (int64_t)((uint64_t)__double_as_longlong(t) & ((1ull << 52) - 1)) - (1ll << 51);
}

// The two workgroup kernels, by file size: SHORT (256 threads, <= 4096 values) converts the whole file once into 16 registers per
// thread; LONG (1024 threads) re-reads the file per pass. Both stride over the list of their slots. Separate kernels because each needs
// its own register budget; the host launches only the ones with work.
template <int MT, int KIND> __global__ __launch_bounds__(MT, KIND == PG_MODEL_LONG ? 8 : 4) void k_slot_model(const FileRef *__restrict__ list, const uint32_t *__restrict__ list_n, const uint32_t *ev_len,
                                                                      const double *samples, PgSlotModel *out, PgSlotDwell *dwell) {
    constexpr bool SHORT = KIND != PG_MODEL_LONG; // register-resident
    __shared__ ModelSmem<MT> sm;
    constexpr int C = SHORT ? ModelCfg<MT>::CACHE : 0;
    const uint32_t n_mine = uniform32(*list_n);
    for (uint32_t it = blockIdx.x; it < n_mine; it += gridDim.x) {
    const uint4 q0 = *reinterpret_cast<const uint4 *>(list + it), q1 = *(reinterpret_cast<const uint4 *>(list + it) + 1);
    const uint64_t first = uniform64((uint64_t)q0.x | ((uint64_t)q0.y << 32)), n = uniform64((uint64_t)q0.z | ((uint64_t)q0.w << 32)), e0 = uniform64((uint64_t)q1.x | ((uint64_t)q1.y << 32));
    const uint32_t nev32 = uniform32(q1.z), s = uniform32(q1.w);
    if (it != blockIdx.x) __syncthreads(); // the previous slot's shared state has been read by everyone
    uint32_t flags = 0;
    PgSlotModel m{};
    m.n = n;
    if (n > 0) {
        bool bad = false, wide = false;
        const int64_t origin = fixed8_dev(samples[first], bad);
        // pass A: range and moments about the first value
        uint64_t mn = ~0ull, mx = 0, hh = 0, hl = 0, ll = 0;
        int64_t s1 = 0;
        auto account = [&](double x) {
            const int64_t v = fixed8_dev(x, bad);
            const uint64_t key = (uint64_t)v ^ (1ull << 63); // order-preserving
            mn = op_min(mn, key); mx = op_max(mx, key);
            const int64_t d = v - origin;
            const uint64_t ad = (uint64_t)(d < 0 ? -d : d);
            if (ad >= (uint64_t)PG_MODEL_MAX_DEV) wide = true;
            const uint64_t h = (ad >> PG_MODEL_LIMB_BITS) & ((1u << PG_MODEL_LIMB_BITS) - 1), l = ad & ((1u << PG_MODEL_LIMB_BITS) - 1);
            s1 += d; hh += h * h; hl += h * l; ll += l * l;
            return key;
        };
        RegKeys<C> rk;
        rk.cnt = 0;
        if constexpr (SHORT) {
            static_assert(MT * ModelCfg<MT>::CACHE == PG_MODEL_SHORT_MAX, "the kernel holds the whole file in registers");
            double x[C];
            rk.cnt = n > threadIdx.x ? (int)((n - threadIdx.x + MT - 1) / MT) : 0;
#pragma unroll
            for (int u = 0; u < C; ++u) { const uint64_t i = (uint64_t)u * MT + threadIdx.x; x[u] = samples[first + (i < n ? i : n - 1)]; }
#pragma unroll
            for (int u = 0; u < C; ++u) rk.k[u] = u < rk.cnt ? account(x[u]) : 0ull;
        } else for_each_mine<MT, ModelCfg<MT>::U, double>(n, [&](uint64_t i) { return samples[first + i]; }, [&](double x) { (void)account(x); });
        mn = block_reduce<MT>(sm, mn, op_min); mx = block_reduce<MT>(sm, mx, op_max);
        m.s1 = (int64_t)block_reduce<MT>(sm, (uint64_t)s1, op_sum);
        m.s2_hh = block_reduce<MT>(sm, hh, op_sum); m.s2_hl = block_reduce<MT>(sm, hl, op_sum); m.s2_ll = block_reduce<MT>(sm, ll, op_sum);
        const uint64_t fl = block_reduce<MT>(sm, (bad ? PG_MODEL_BAD_VALUE : 0u) | (wide ? PG_MODEL_BAD_SPREAD : 0u), [](uint64_t a, uint64_t b) { return a | b; });
        flags = (uint32_t)fl | (n > PG_MODEL_MAX_VALUES ? PG_MODEL_BAD_COUNT : 0u);
        m.origin = origin;
        if (!(flags & (PG_MODEL_BAD_VALUE | PG_MODEL_BAD_COUNT))) {
            const uint64_t spread = mx - mn;
            const int bits = spread ? 64 - __builtin_clzll(spread) : 0;
            uint64_t k_lo, k_hi;
            if constexpr (SHORT) { rk.mn = mn; block_middle<MT>(sm, rk, n, bits, k_lo, k_hi); }
            else {
                auto key_at = [&](uint64_t i) {
                    bool b2 = false;
                    return ((uint64_t)fixed8_dev(samples[first + i], b2) ^ (1ull << 63)) - mn;
                };
                const GlobalKeys<MT, decltype(key_at)> gk{n, key_at};
                block_middle_long<MT>(sm, gk, n, bits, k_lo, k_hi);
            }
            m.mid_lo = (int64_t)((k_lo + mn) ^ (1ull << 63));
            m.mid_hi = (int64_t)((k_hi + mn) ^ (1ull << 63));
        }
    }
    // dwell: awk prints one comma count per ';'-separated field: samples - 1 per event, and 0 for the empty last field
    PgSlotDwell dw{};
    const uint64_t nev = nev32;
    if (nev > 0) {
        const uint64_t nd = nev + 1;
        auto dkey = [&](uint64_t i) { const uint32_t len = i < nev ? ev_len[e0 + i] : 0u; return (uint64_t)(len ? len - 1 : 0u); };
        const GlobalKeys<MT, decltype(dkey)> dk{nd, dkey};
        uint64_t mx = 0;
        dk.each([&](uint64_t k) { mx = op_max(mx, k); });
        mx = block_reduce<MT>(sm, mx, op_max);
        const int bits = mx ? 64 - __builtin_clzll(mx) : 0;
        uint64_t k_lo, k_hi;
        block_middle<MT>(sm, dk, nd, bits, k_lo, k_hi);
        dw.n = nd; dw.mid_lo = (uint32_t)k_lo; dw.mid_hi = (uint32_t)k_hi;
    }
    dw.flags = flags;
    if (threadIdx.x == 0) { out[s] = m; dwell[s] = dw; }
    } // slots of this workgroup
}

// ---- One wave per file, rewritten in round 5 (the TINY kind: at k = 9 all 262 144 files) ----------------------------------------------
// The variant of k_slot_model above ran a file through eight DEPENDENT memory rounds (record, first value, values, then the event lengths
// once per pass of the dwell selection) and through five 8-bit selection windows with seven separate reductions in between: a wave lived
// 19 us for 4 us of instructions (profiles/r05_model_pmc.txt); with the rounds gone the kernel is bound by its instruction count
// (262 144 waves on 1024 SIMDs), so the arithmetic per value is what this version is written for:
//  * the record, then ALL the file's values and event lengths at once (two memory rounds);
//  * pg_fixed8 through the 2^52 trick: t = p + 1.5 * 2^52 holds rint(p) in its low mantissa bits (|p| < 2^51), so t's BIT PATTERN minus
//    that of the smallest t is the selection key (one integer subtraction, no conversion), and t - t0 is the deviation from the first
//    value as an exact double; a file with a value of 2.2e7 or more, or a NaN, is handed to the 256-thread kernel (its list);
//  * the moments in FP64, every step exact: |d| = ah * 2^20 + al by a scaling and a truncation, the three limb products by FMA (each
//    < 2^40, a lane's sum of 16 < 2^44, the wave's < 2^50) -- v_mad_u64_u32 is a quarter-rate instruction, v_fma_f64 half rate;
//  * no validity tests per value: a lane without a value loads the FIRST one (deviation 0, neither minimum nor maximum) and gets a key
//    above every real one for the selection;
//  * reductions as DPP chains without LDS or barriers (independent, so they interleave);
//  * the selection leaves its 8-bit windows as soon as ranking the remaining candidates against each other (one per lane) is cheaper
//    than another window -- for a med-MAD file after one or two -- in 32-bit arithmetic when the spread allows.
#define WV_MAGIC 6755399441055744.0             /* 1.5 * 2^52 (even): p + WV_MAGIC has ulp 1 */
#define WV_MAX_ABS 2.2e7                        /* |x| * 1e8 < 2^51 */
struct WaveSmem { uint32_t hist[256]; uint64_t cand[64]; };

__device__ __forceinline__ uint64_t readlane64(uint64_t v, int l) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
}
__device__ __forceinline__ uint32_t readlane_k(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ uint64_t readlane_k(uint64_t v, int l) { return readlane64(v, l); }
template <int CTRL, int ROW_MASK> __device__ __forceinline__ uint64_t dpp64(uint64_t v, uint64_t ident) { // a lane without a source keeps `ident`
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)ident, (int)(uint32_t)v, CTRL, ROW_MASK, 0xF, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(ident >> 32), (int)(uint32_t)(v >> 32), CTRL, ROW_MASK, 0xF, false);
    return ((uint64_t)hi << 32) | lo;
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ uint32_t dpp32(uint32_t v, uint32_t ident) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)ident, (int)v, CTRL, ROW_MASK, 0xF, false);
}
// op over the wave's 64 lanes (all active), the result uniform: row_shr 1/2/4/8, row_bcast 15 / 31 as wave_incl_scan_u32 (pg_dev.h)
template <class Op> __device__ __forceinline__ uint64_t wave_reduce64(uint64_t v, uint64_t ident, Op op) {
    v = op(v, dpp64<0x111, 0xF>(v, ident));
    v = op(v, dpp64<0x112, 0xF>(v, ident));
    v = op(v, dpp64<0x114, 0xF>(v, ident));
    v = op(v, dpp64<0x118, 0xF>(v, ident));
    v = op(v, dpp64<0x142, 0xA>(v, ident));
    v = op(v, dpp64<0x143, 0xC>(v, ident));
    return readlane64(v, 63);
}
template <class Op> __device__ __forceinline__ uint32_t wave_reduce32(uint32_t v, uint32_t ident, Op op) {
    v = op(v, dpp32<0x111, 0xF>(v, ident));
    v = op(v, dpp32<0x112, 0xF>(v, ident));
    v = op(v, dpp32<0x114, 0xF>(v, ident));
    v = op(v, dpp32<0x118, 0xF>(v, ident));
    v = op(v, dpp32<0x142, 0xA>(v, ident));
    v = op(v, dpp32<0x143, 0xC>(v, ident));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// FP64 reductions. The sum: lanes without a source read zero (bound_ctrl), so no register has to be prepared for them. Minimum and
// maximum through the instruction itself: fmin() / fmax() would first canonicalise both operands (two more FP64 instructions per step)
// against signalling NaNs that values which came out of an addition cannot be.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dpp_f64_zero(double v) {
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, ROW_MASK, 0xF, true);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, ROW_MASK, 0xF, true);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
__device__ __forceinline__ double wave_sum_f64(double v) {
    v += dpp_f64_zero<0x111, 0xF>(v); v += dpp_f64_zero<0x112, 0xF>(v); v += dpp_f64_zero<0x114, 0xF>(v); v += dpp_f64_zero<0x118, 0xF>(v);
    v += dpp_f64_zero<0x142, 0xA>(v); v += dpp_f64_zero<0x143, 0xC>(v);
    return __longlong_as_double((long long)readlane64((uint64_t)__double_as_longlong(v), 63));
}
__device__ __forceinline__ double min_f64_raw(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double max_f64_raw(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
template <class Op> __device__ __forceinline__ double wave_reduce_f64(double v, double ident, Op op) {
    return __longlong_as_double((long long)wave_reduce64((uint64_t)__double_as_longlong(v), (uint64_t)__double_as_longlong(ident),
        [&](uint64_t a, uint64_t b) { return (uint64_t)__double_as_longlong(op(__longlong_as_double((long long)a), __longlong_as_double((long long)b))); }));
}
// an integer-valued double as int64: |v| < 2^51 (one addition and the mantissa's low bits instead of the conversion sequence)
__device__ __forceinline__ int64_t small_f64_to_i64(double v) {
    const uint64_t b = (uint64_t)__double_as_longlong(v + WV_MAGIC);
    return (int64_t)(b & ((1ull << 52) - 1)) - (1ll << 51);
}
__device__ __forceinline__ uint32_t wave_min_k(uint32_t v) { return wave_reduce32(v, ~0u, [](uint32_t a, uint32_t b) { return a < b ? a : b; }); }
__device__ __forceinline__ uint64_t wave_min_k(uint64_t v) { return wave_reduce64(v, ~0ull, op_min); }
template <int CTRL, int ROW_MASK> __device__ __forceinline__ uint32_t dpp32z(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xF, true); }
__device__ __forceinline__ uint32_t wave_scan32(uint32_t v) {
    v += dpp32z<0x111, 0xF>(v); v += dpp32z<0x112, 0xF>(v); v += dpp32z<0x114, 0xF>(v); v += dpp32z<0x118, 0xF>(v);
    v += dpp32z<0x142, 0xA>(v); v += dpp32z<0x143, 0xC>(v);
    return v;
}
__device__ __forceinline__ uint32_t lanes_below(uint64_t m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }

// Keys of a wave: element u of lane l is key u * 64 + l of n; a lane's elements beyond n hold ~0, real keys are < 2^bits (bits < the
// key type's width), rows beyond n are never touched.
// the smallest key whose bits above `hi` exceed `prefix` (the caller knows there is one)
template <class K, int C> __device__ __forceinline__ K wave_above(const K (&k)[C], uint32_t n, int hi, K prefix) {
    K a = (K)~(K)0;
#pragma unroll
    for (int u = 0; u < C; ++u)
        if ((uint32_t)u * 64 < n) { if ((K)(k[u] >> hi) > prefix && k[u] < a) a = k[u]; }
    return wave_min_k(a);
}

// datamash's two middle order statistics of the n keys. 8-bit windows from the top while that is cheaper than the alternative: the
// candidates' remaining bits go to LDS, one per lane (<= 64), and every lane counts the candidates below / not above its own -- the
// lane whose interval holds a rank owns that order statistic.
template <class K, int C> __device__ void wave_middle(WaveSmem &sm, const K (&k)[C], uint32_t n, int bits, K &k_lo, K &k_hi) {
    const uint32_t lane = threadIdx.x;
    uint32_t r_lo = (n - 1) / 2, r_hi = n / 2, cnt = n; // ranks among the candidates; r_lo < cnt, r_hi <= cnt
    K prefix = 0;
    int hi = bits; // candidates: the keys with (key >> hi) == prefix
    const uint32_t window_cost = 40 + 6 * ((n + 63) / 64); // in instructions, against ~5 per candidate of the ranking loop
    while (hi > 0 && (cnt > 64 || cnt * 5 > window_cost)) {
        const int wbits = hi >= 8 ? 8 : hi, shift = hi - wbits;
        const uint32_t mask = (1u << wbits) - 1;
        __syncthreads();
        reinterpret_cast<uint4 *>(sm.hist)[lane] = make_uint4(0, 0, 0, 0);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < C; ++u)
            if ((uint32_t)u * 64 < n) { if ((K)(k[u] >> hi) == prefix) atomicAdd(&sm.hist[(uint32_t)(k[u] >> shift) & mask], 1u); }
        __syncthreads();
        const uint4 h = reinterpret_cast<const uint4 *>(sm.hist)[lane];
        const uint32_t sum = h.x + h.y + h.z + h.w, base = wave_scan32(sum) - sum;
        const int L = __builtin_ctzll(__ballot(r_lo >= base && r_lo < base + sum)); // exactly one lane: the counts add up to cnt > r_lo
        uint32_t below = (uint32_t)__builtin_amdgcn_readlane((int)base, L);
        const uint32_t hx = (uint32_t)__builtin_amdgcn_readlane((int)h.x, L), hy = (uint32_t)__builtin_amdgcn_readlane((int)h.y, L),
                       hz = (uint32_t)__builtin_amdgcn_readlane((int)h.z, L), hw = (uint32_t)__builtin_amdgcn_readlane((int)h.w, L);
        uint32_t bin = (uint32_t)L * 4, cb = hx;
        if (r_lo >= below + cb) { below += cb; cb = hy; ++bin;
            if (r_lo >= below + cb) { below += cb; cb = hz; ++bin;
                if (r_lo >= below + cb) { below += cb; cb = hw; ++bin; } } }
        r_lo -= below; r_hi -= below; cnt = cb;
        prefix = (K)((prefix << wbits) | bin);
        hi -= wbits;
    }
    if (hi == 0) { // every bit settled: the candidates are one value
        k_lo = prefix;
        k_hi = r_hi < cnt ? prefix : wave_above<K, C>(k, n, 0, prefix);
        return;
    }
    // (cnt <= 64 here)
    const K low_mask = (K)(((K)1 << hi) - 1), top = (K)(prefix << hi);
    K *cand = reinterpret_cast<K *>(sm.cand);
    __syncthreads();
    uint32_t run = 0;
#pragma unroll
    for (int u = 0; u < C; ++u)
        if ((uint32_t)u * 64 < n) {
            const bool m = (K)(k[u] >> hi) == prefix;
            const uint64_t b = __ballot(m);
            if (m) cand[run + lanes_below(b)] = k[u] & low_mask;
            run += (uint32_t)__builtin_popcountll(b);
        }
    __syncthreads();
    const bool have = lane < cnt;
    uint32_t lt = 0, le = 0;
    K mine = (K)~(K)0;
    if (hi <= 32) { // (always, for 32-bit keys)
        const uint32_t m32 = have ? (uint32_t)cand[lane] : ~0u;
        for (uint32_t j = 0; j < cnt; ++j) { const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)m32, (int)j); lt += c < m32; le += c <= m32; }
        mine = (K)m32;
    } else {
        mine = have ? cand[lane] : (K)~(K)0;
        for (uint32_t j = 0; j < cnt; ++j) { const K c = readlane_k(mine, (int)j); lt += c < mine; le += c <= mine; }
    }
    const int l_lo = __builtin_ctzll(__ballot(have && lt <= r_lo && r_lo < le));
    k_lo = top | readlane_k(mine, l_lo);
    if (r_hi == r_lo) k_hi = k_lo;
    else if (r_hi < cnt) k_hi = top | readlane_k(mine, __builtin_ctzll(__ballot(have && lt <= r_hi && r_hi < le)));
    else k_hi = wave_above<K, C>(k, n, hi, prefix);
}

// One file on one wave: C rows of 64 values, D rows of 64 events. Returns false when the file holds a value the conversion cannot take.
template <int C, int D> __device__ __forceinline__ bool wave_file(WaveSmem &sm, const FileRef &fr, uint32_t s, const uint32_t *__restrict__ ev_len,
                                                                   const double *__restrict__ samples, PgSlotModel *__restrict__ out, PgSlotDwell *__restrict__ dwell) {
    constexpr int WV_C = C, WV_D = D;
    const uint32_t lane = threadIdx.x;
    const uint64_t first = uniform64(fr.first), e0 = uniform64(fr.e0); // (scalar registers: the row guards below are scalar branches)
    const uint32_t n = uniform32((uint32_t)fr.n), nev = uniform32(fr.nev), nd = nev ? nev + 1 : 0; // n <= 64 C, nd <= 64 D (pg_model_kind)
    // every load of the file at once: its events' lengths and its values (a lane beyond the end takes the first one), each value turned into
    // t = pg_fixed8 + 1.5 * 2^52 as it arrives. (More than 32 rows: in two halves, so that at most 32 raw rows are held beside the t's.)
    uint32_t len[WV_D];
#pragma unroll
    for (int u = 0; u < WV_D; ++u) if ((uint32_t)u * 64 < nev) { const uint32_t i = (uint32_t)u * 64 + lane; len[u] = ev_len[e0 + (i < nev ? i : 0u)]; } else len[u] = 0;
    double t[WV_C];
    bool big = false;
    double tmin = INFINITY, tmax = -INFINITY;
    constexpr int CH = WV_C > 32 ? 32 : WV_C;
#pragma unroll
    for (int h = 0; h < WV_C; h += CH) {
        double x[CH];
#pragma unroll
        for (int v = 0; v < CH; ++v) { const int u = h + v; if ((uint32_t)u * 64 < n) { const uint32_t i = (uint32_t)u * 64 + lane; x[v] = samples[first + (i < n ? i : 0u)]; } else x[v] = 0.0; }
#pragma unroll
        for (int v = 0; v < CH; ++v) {
            const int u = h + v;
            if ((uint32_t)u * 64 < n) {
                big |= !(fabs(x[v]) < WV_MAX_ABS); // also NaN
                const double p = x[v] * 1e8, e = fma(x[v], 1e8, -p);
                double tt = p + WV_MAGIC; // rint(p), ties to even (WV_MAGIC is even)
                const double f = p - (tt - WV_MAGIC);
                if (__ballot(fabs(f) == 0.5)) { // p sits on a tie that the addition broke to even; e says on which side the exact product lies
                    if (f == 0.5 && e > 0.0) tt += 1.0;
                    else if (f == -0.5 && e < 0.0) tt -= 1.0;
                }
                t[u] = tt;
                tmin = min_f64_raw(tmin, tt); tmax = max_f64_raw(tmax, tt);
            } else t[u] = 0.0;
        }
    }
    PgSlotModel m{};
    m.n = n;
    uint32_t flags = 0;
    if (n > 0) {
        if (__ballot(big)) return false; // outside the 2^52 trick (or not a number): the 256-thread kernel converts the general way and reports
        const double t0 = __longlong_as_double((long long)readlane64((uint64_t)__double_as_longlong(t[0]), 0)); // the file's first value
        double s1 = 0.0, hh = 0.0, hl = 0.0, ll = 0.0;
#pragma unroll
        for (int u = 0; u < WV_C; ++u)
            if ((uint32_t)u * 64 < n) {
                const double d = t[u] - t0, ad = fabs(d);              // exact: both in [2^52, 2^53)
                const double ah = trunc(ad * 0x1p-20), al = fma(ah, -0x1p20, ad); // |d| = ah * 2^20 + al, exact
                s1 += d; hh = fma(ah, ah, hh); hl = fma(ah, al, hl); ll = fma(al, al, ll);
            }
        tmin = wave_reduce_f64(tmin, INFINITY, [](double a, double b) { return min_f64_raw(a, b); });
        tmax = wave_reduce_f64(tmax, -INFINITY, [](double a, double b) { return max_f64_raw(a, b); });
        s1 = wave_sum_f64(s1); hh = wave_sum_f64(hh); hl = wave_sum_f64(hl); ll = wave_sum_f64(ll); // exact: integers below 2^50
        const bool wide = !(tmax - t0 < 0x1p40 && t0 - tmin < 0x1p40); // some |d| >= PG_MODEL_MAX_DEV: the limb sums above mean nothing
        flags = wide ? PG_MODEL_BAD_SPREAD : 0u;
        const auto units_of = [](double t) { return (int64_t)((uint64_t)__double_as_longlong(t) & ((1ull << 52) - 1)) - (1ll << 51); }; // t = units + 1.5 * 2^52
        m.origin = units_of(t0);
        if (!wide) { m.s1 = small_f64_to_i64(s1); m.s2_hh = (uint64_t)small_f64_to_i64(hh); m.s2_hl = (uint64_t)small_f64_to_i64(hl); m.s2_ll = (uint64_t)small_f64_to_i64(ll); }
        // selection keys: the bit pattern of t above that of the smallest t (one exponent: the difference of the patterns is that of the units)
        const uint64_t bmin = (uint64_t)__double_as_longlong(tmin), spread = (uint64_t)__double_as_longlong(tmax) - bmin;
        const int bits = spread ? 64 - __builtin_clzll(spread) : 0; // <= 53
        const int64_t vmin = units_of(tmin);
        if (bits <= 31) {
            uint32_t key[WV_C];
#pragma unroll
            for (int u = 0; u < WV_C; ++u) key[u] = ((uint32_t)u * 64 + lane < n) ? (uint32_t)__double_as_longlong(t[u]) - (uint32_t)bmin : ~0u;
            uint32_t k_lo, k_hi;
            wave_middle<uint32_t, WV_C>(sm, key, n, bits, k_lo, k_hi);
            m.mid_lo = vmin + (int64_t)k_lo; m.mid_hi = vmin + (int64_t)k_hi;
        } else {
            uint64_t key[WV_C];
#pragma unroll
            for (int u = 0; u < WV_C; ++u) key[u] = ((uint32_t)u * 64 + lane < n) ? (uint64_t)__double_as_longlong(t[u]) - bmin : ~0ull;
            uint64_t k_lo, k_hi;
            wave_middle<uint64_t, WV_C>(sm, key, n, bits, k_lo, k_hi);
            m.mid_lo = vmin + (int64_t)k_lo; m.mid_hi = vmin + (int64_t)k_hi;
        }
    }
    // dwell: awk prints one comma count per ';'-separated field: samples - 1 per event, and 0 for the empty last field
    PgSlotDwell dw{};
    if (nd > 0) {
        uint32_t dk[WV_D], dmx = 0;
#pragma unroll
        for (int u = 0; u < WV_D; ++u) {
            const uint32_t i = (uint32_t)u * 64 + lane;
            const uint32_t v = (i < nev && len[u]) ? len[u] - 1 : 0u;
            dmx = v > dmx ? v : dmx;
            dk[u] = i < nd ? v : ~0u;
        }
        dmx = wave_reduce32(dmx, 0u, [](uint32_t a, uint32_t b) { return a > b ? a : b; });
        const int bits = dmx ? 32 - __builtin_clz(dmx) : 0;
        uint32_t k_lo = 0, k_hi = 0;
        if (bits <= 31) wave_middle<uint32_t, WV_D>(sm, dk, nd, bits, k_lo, k_hi);
        else k_lo = k_hi = 0; // (a window of 2^31 samples or more: op lengths are below 2^24)
        dw.n = nd; dw.mid_lo = k_lo; dw.mid_hi = k_hi;
    }
    dw.flags = flags;
    if (lane == 0) { out[s] = m; dwell[s] = dw; }
    return true;
}

// The kernel over ALL slots (a wave each): the TINY files are reduced here, the others go on the list of their kind (lists[kind - 1]).
__global__ __launch_bounds__(64, PG_MODEL_TINY_WAVES) void k_slot_model_wave(const uint64_t *__restrict__ ev_off, const uint64_t *__restrict__ samp_off, uint32_t n_slots, uint32_t drop_first,
                                                                              const uint32_t *__restrict__ ev_len, const double *__restrict__ samples, PgSlotModel *__restrict__ out,
                                                                              PgSlotDwell *__restrict__ dwell, FileRef *__restrict__ lists, uint32_t *__restrict__ counts, int short_kind) {
    __shared__ WaveSmem sm;
    const uint32_t s = blockIdx.x;
    if (s >= n_slots) return;
    const FileRef fr = file_ref(ev_off, samp_off, s, drop_first);
    int kind = (int)uniform32((uint32_t)pg_model_kind(fr.n, fr.nev));
    if (kind == PG_MODEL_TINY && !wave_file<PG_MODEL_TINY_MAX / 64, PG_MODEL_TINY_EVENTS / 64>(sm, fr, s, ev_len, samples, out, dwell)) kind = PG_MODEL_SHORT;
    if (kind == PG_MODEL_SHORT) kind = short_kind; // (the launcher may send the few SHORT files of a job with the LONG ones: one launch less)
    if (kind != PG_MODEL_TINY && threadIdx.x == 0) lists[(size_t)(kind - 1) * n_slots + atomicAdd(counts + (kind - 1), 1u)] = fr; // (any order: every slot writes its own result)
}
// The MID files (up to 2048 values and 511 events: at k = 9 the ~3 000 files above the one-wave size): the same code with 32 rows, over their list.
__global__ __launch_bounds__(64, 2) void k_slot_model_wave_mid(const uint64_t *__restrict__ ev_off, const uint64_t *__restrict__ samp_off, uint32_t n_slots, uint32_t drop_first,
                                                                const uint32_t *__restrict__ ev_len, const double *__restrict__ samples, PgSlotModel *__restrict__ out,
                                                                PgSlotDwell *__restrict__ dwell, FileRef *__restrict__ lists, uint32_t *__restrict__ counts, int short_kind) {
    __shared__ WaveSmem sm;
    const uint32_t n_mine = uniform32(counts[PG_MODEL_MID - 1]);
    for (uint32_t it = blockIdx.x; it < n_mine; it += gridDim.x) {
        const FileRef fr = lists[(size_t)(PG_MODEL_MID - 1) * n_slots + it];
        const uint32_t s = uniform32(fr.slot);
        if (it != blockIdx.x) __syncthreads();
        if (!wave_file<PG_MODEL_MID_MAX / 64, PG_MODEL_MID_EVENTS / 64>(sm, fr, s, ev_len, samples, out, dwell) && threadIdx.x == 0)
            lists[(size_t)(short_kind - 1) * n_slots + atomicAdd(counts + (short_kind - 1), 1u)] = fr;
    }
}

} // namespace

size_t pg_slot_model_scratch_bytes(uint32_t n_slots) { return 3 * (size_t)n_slots * sizeof(FileRef) + 64; }
hipError_t pg_launch_slot_model(hipStream_t st, uint32_t n_slots, const int any_kind[4], const uint64_t *ev_off, const uint64_t *samp_off,
                                const uint32_t *ev_len, const double *samples, uint32_t drop_first, PgSlotModel *out, PgSlotDwell *dwell, void *scratch) {
    if (n_slots == 0) return hipSuccess;
    (void)hipGetLastError(); // sticky per-thread state of an unrelated earlier failure
    // scratch: [3] list lengths (+ padding to 64 bytes), the lists of the MID, SHORT and LONG files (FileRef entries)
    uint32_t *counts = static_cast<uint32_t *>(scratch);
    FileRef *lists = reinterpret_cast<FileRef *>(counts + 16);
    // any_kind[] = how many files of each kind the host knows of (1 << 20 = "some" when it does not hold the offsets). A workgroup kernel's launch is
    // the lifetime of one workgroup (~55 us at k = 9 for 34 SHORT and 28 LONG files): when both kinds together fit one round of the
    // 1024-thread kernel, the SHORT files go on its list too.
    const bool merge = any_kind[PG_MODEL_SHORT] > 0 && any_kind[PG_MODEL_LONG] > 0 && any_kind[PG_MODEL_SHORT] + any_kind[PG_MODEL_LONG] <= 512 && !getenv("PGMOVE_MODEL_NO_MERGE");
    const int short_kind = merge ? PG_MODEL_LONG : PG_MODEL_SHORT;
    const hipError_t e = hipMemsetAsync(counts, 0, 64, st);
    if (e != hipSuccess) return e;
    // the one-wave kernel over all slots, always: it is also the pass that sorts the slots into the lists (a wave that is not TINY leaves after four
    // scalar loads). (Lists written by the host when it holds the offsets and no file is TINY: measured at k = 5, two small uploads cost what
    // the launch does -- 60.8 against 60.0 us for the model of the headline job.)
    hipLaunchKernelGGL(k_slot_model_wave, dim3(n_slots), dim3(64), 0, st, ev_off, samp_off, n_slots, drop_first, ev_len, samples, out, dwell, lists, counts, short_kind);
    if (any_kind[PG_MODEL_MID]) hipLaunchKernelGGL(k_slot_model_wave_mid, dim3(n_slots < 4096u ? n_slots : 4096u), dim3(64), 0, st, ev_off, samp_off, n_slots, drop_first, ev_len, samples, out, dwell, lists, counts, short_kind);
    // (the one-wave kernels hand files with values of 2.2e7 and beyond to the 256-thread one)
    if (!merge && (any_kind[PG_MODEL_SHORT] || any_kind[PG_MODEL_TINY] || any_kind[PG_MODEL_MID]))
        hipLaunchKernelGGL((k_slot_model<256, PG_MODEL_SHORT>), dim3(n_slots < 2048u ? n_slots : 2048u), dim3(256), 0, st,
                           (const FileRef *)(lists + (size_t)(PG_MODEL_SHORT - 1) * n_slots), (const uint32_t *)(counts + (PG_MODEL_SHORT - 1)), ev_len, samples, out, dwell);
    if (any_kind[PG_MODEL_LONG])
        hipLaunchKernelGGL((k_slot_model<1024, PG_MODEL_LONG>), dim3(n_slots < 512u ? n_slots : 512u), dim3(1024), 0, st,
                           (const FileRef *)(lists + (size_t)(PG_MODEL_LONG - 1) * n_slots), (const uint32_t *)(counts + (PG_MODEL_LONG - 1)), ev_len, samples, out, dwell);
    return hipGetLastError();
}
