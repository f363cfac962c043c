// pg_model.hip -- per-k-mer model reduction over the kept samples: exact median, exact moments, dwell median.
//
// Replaces the text round trip of the reference's pipeline (scripts/poregen.sh:54-85 calculate_mean_stddev_all and
// :33-52 calculate_dwell_times_medians): dump files -> tr/tail/awk -> datamash median / sstdev. See pg_model.h for
// why the arithmetic is done on integers of 1e-8 units.
//
// One workgroup per slot (= k-mer file). The slot's values are one contiguous range of the kept-sample array, read
// (2 + ceil(bits/12)) times (+1 when the count is even): min/max + moments, then a radix select that only walks the bits in
// which the slot's values differ (36 bits for med-MAD scaled data: 3 passes), then the upper middle value. The slot
// stays in L2 / MALL between the passes (150 k values = 1.2 MB at sample_limit 5000), so HBM sees it about once.
#include "pg_internal.h"
#include "pg_model.h"

#ifndef PG_MODEL_TINY_BITS
#define PG_MODEL_TINY_BITS 8
#endif
#ifndef PG_MODEL_TINY_WAVES
#define PG_MODEL_TINY_WAVES 4 // waves per SIMD the one-wave variant is compiled for
#endif
namespace {

// loads in flight per thread: the 1024-thread variant must stay within 64 VGPRs (two workgroups per CU), its memory
// parallelism comes from 32 waves per CU; the 256-thread variant serves short files, where the round trips are the cost
template <int MT> struct ModelCfg {
    static constexpr int BITS = MT >= 256 ? 12 : PG_MODEL_TINY_BITS;    // window of the radix select; a single wave scans 256 bins, not 4096
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

// Round 5: the three kernels used to be launched over ALL slots each, a workgroup leaving after two dependent loads when its k-mer belongs
// to another kernel -- at k = 9 (262 144 k-mers, nearly all of them "tiny") the SHORT and LONG launches were 0.37 + 0.50 ms of mostly
// empty workgroups next to the 1.24 ms of real work (rocprofv3 GRBM_GUI_ACTIVE, profiles/r05_model_pmc.txt). A record pass classifies the
// slots once: rec[s] = where the slot's values and events lie + its kind (ONE 32-byte load in the kernels instead of two dependent rounds),
// and the slots of the two rarer kinds go on lists that their kernels stride over with a small grid.
struct __attribute__((aligned(16))) PgModelRec { uint64_t first, n, e0; uint32_t nev, kind; };
__global__ __launch_bounds__(256) void k_model_classify(const uint64_t *__restrict__ ev_off, const uint64_t *__restrict__ samp_off, uint32_t n_slots, uint32_t drop_first,
                                                        PgModelRec *__restrict__ rec, uint32_t *__restrict__ lists /* [2][n_slots] */, uint32_t *__restrict__ counts /* [2] */) {
    const uint32_t s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n_slots) return;
    const uint64_t e0 = ev_off[s], e1 = ev_off[s + 1];
    const uint64_t a0 = samp_off[e0], a1 = samp_off[e1];
    const uint64_t skip = (a1 > a0 && drop_first) ? 1 : 0; // `tail -n +2`: the file's first value never reaches datamash
    PgModelRec r;
    r.first = a0 + skip; r.n = a1 - r.first; r.e0 = e0; r.nev = (uint32_t)(e1 - e0); r.kind = (uint32_t)pg_model_kind(r.n);
    rec[s] = r;
    if (r.kind != PG_MODEL_TINY) lists[(size_t)(r.kind - 1) * n_slots + atomicAdd(counts + (r.kind - 1), 1u)] = s; // (any order: every slot writes its own result)
}

// Three kernels share this body, by file size: TINY (one wave, <= 1024 values: k = 9 jobs have 262 144 such files, and a wave
// needs no workgroup barrier) and SHORT (256 threads, <= 4096 values) convert the whole file once into 16 registers per
// thread; LONG (1024 threads) re-reads the file per pass. A workgroup whose k-mer belongs to another kernel leaves after two
// loads. Separate kernels because each needs its own register budget; the host launches only the ones with work.
template <int MT, int KIND> __global__ __launch_bounds__(MT, KIND == PG_MODEL_LONG ? 8 : (KIND == PG_MODEL_TINY ? PG_MODEL_TINY_WAVES : 4)) void k_slot_model(const PgModelRec *__restrict__ rec, const uint32_t *__restrict__ list,
                                                                      const uint32_t *__restrict__ list_n, uint32_t n_slots, const uint32_t *ev_len,
                                                                      const double *samples, PgSlotModel *out, PgSlotDwell *dwell) {
    constexpr bool SHORT = KIND != PG_MODEL_LONG; // register-resident
    __shared__ ModelSmem<MT> sm;
    constexpr int C = SHORT ? ModelCfg<MT>::CACHE : 0;
    // TINY: a workgroup (= a wave) per slot; the rarer kinds: the workgroups stride over their list
    const uint32_t n_mine = KIND == PG_MODEL_TINY ? n_slots : uniform32(*list_n);
    for (uint32_t it = blockIdx.x; it < n_mine; it += gridDim.x) {
    const uint32_t s = KIND == PG_MODEL_TINY ? it : uniform32(list[it]);
    const uint4 rq0 = *reinterpret_cast<const uint4 *>(rec + s), rq1 = *(reinterpret_cast<const uint4 *>(rec + s) + 1);
    const uint64_t first = uniform64((uint64_t)rq0.x | ((uint64_t)rq0.y << 32)), n = uniform64((uint64_t)rq0.z | ((uint64_t)rq0.w << 32));
    const uint64_t e0 = uniform64((uint64_t)rq1.x | ((uint64_t)rq1.y << 32));
    const uint32_t nev32 = uniform32(rq1.z);
    if (KIND == PG_MODEL_TINY && uniform32(rq1.w) != (uint32_t)PG_MODEL_TINY) continue; // (gridDim.x == n_slots: the loop ends)
    if (it != blockIdx.x) __syncthreads(); // the previous slot's shared state has been read by everyone
    uint32_t flags = 0;
    PgSlotModel m{};
    m.n = n;
    if (n > 0) {
        bool bad = false, wide = false;
        const int64_t origin = pg_fixed8(samples[first], bad);
        // pass A: range and moments about the first value
        uint64_t mn = ~0ull, mx = 0, hh = 0, hl = 0, ll = 0;
        int64_t s1 = 0;
        auto account = [&](double x) {
            const int64_t v = pg_fixed8(x, bad);
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
            static_assert(MT * ModelCfg<MT>::CACHE == (KIND == PG_MODEL_TINY ? PG_MODEL_TINY_MAX : PG_MODEL_SHORT_MAX), "the kernel holds the whole file in registers");
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
                    return ((uint64_t)pg_fixed8(samples[first + i], b2) ^ (1ull << 63)) - mn;
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

} // namespace

size_t pg_slot_model_scratch_bytes(uint32_t n_slots) { return (size_t)n_slots * sizeof(PgModelRec) + 2 * (size_t)n_slots * 4 + 64; }
hipError_t pg_launch_slot_model(hipStream_t st, uint32_t n_slots, const int any_kind[3], const uint64_t *ev_off, const uint64_t *samp_off,
                                const uint32_t *ev_len, const double *samples, uint32_t drop_first, PgSlotModel *out, PgSlotDwell *dwell, void *scratch) {
    if (n_slots == 0) return hipSuccess;
    (void)hipGetLastError(); // sticky per-thread state of an unrelated earlier failure
    // scratch: [2] list lengths (+ padding to 64 bytes), records, the two lists
    uint32_t *counts = static_cast<uint32_t *>(scratch);
    PgModelRec *rec = reinterpret_cast<PgModelRec *>(static_cast<char *>(scratch) + 64);
    uint32_t *lists = reinterpret_cast<uint32_t *>(rec + n_slots);
    hipError_t e = hipMemsetAsync(counts, 0, 64, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_model_classify, dim3((n_slots + 255) / 256), dim3(256), 0, st, ev_off, samp_off, n_slots, drop_first, rec, lists, counts);
    if (any_kind[PG_MODEL_TINY]) hipLaunchKernelGGL((k_slot_model<64, PG_MODEL_TINY>), dim3(n_slots), dim3(64), 0, st, (const PgModelRec *)rec, (const uint32_t *)nullptr, (const uint32_t *)nullptr, n_slots, ev_len, samples, out, dwell);
    if (any_kind[PG_MODEL_SHORT]) hipLaunchKernelGGL((k_slot_model<256, PG_MODEL_SHORT>), dim3(n_slots < 2048u ? n_slots : 2048u), dim3(256), 0, st, (const PgModelRec *)rec, (const uint32_t *)lists, (const uint32_t *)counts, n_slots, ev_len, samples, out, dwell);
    if (any_kind[PG_MODEL_LONG]) hipLaunchKernelGGL((k_slot_model<1024, PG_MODEL_LONG>), dim3(n_slots < 512u ? n_slots : 512u), dim3(1024), 0, st, (const PgModelRec *)rec, (const uint32_t *)(lists + n_slots), (const uint32_t *)(counts + 1), n_slots, ev_len, samples, out, dwell);
    return hipGetLastError();
}
