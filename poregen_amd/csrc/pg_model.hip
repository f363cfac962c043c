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

namespace {

constexpr int M_BITS = 12, M_BINS = 1 << M_BITS;

template <int MT> struct ModelSmem {
    uint32_t hist[M_BINS];
    uint64_t red[MT / 64];
    uint32_t wsum[MT / 64];
    uint32_t found_bin, found_below;
};

__device__ __forceinline__ uint64_t op_sum(uint64_t a, uint64_t b) { return a + b; }
__device__ __forceinline__ uint64_t op_min(uint64_t a, uint64_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint64_t op_max(uint64_t a, uint64_t b) { return a > b ? a : b; }

// every thread gets the reduction of v over the block
template <int MT, class Op> __device__ uint64_t block_reduce(ModelSmem<MT> &sm, uint64_t v, Op op) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = op(v, (uint64_t)__shfl_xor((unsigned long long)v, o, 64));
    __syncthreads(); // red[] of the previous call has been read by everyone
    if ((threadIdx.x & 63) == 0) sm.red[threadIdx.x >> 6] = v;
    __syncthreads();
    uint64_t r = sm.red[0];
#pragma unroll
    for (int w = 1; w < MT / 64; ++w) r = op(r, sm.red[w]);
    return r;
}

// bin b with below(b) <= rank < below(b) + hist[b]; the total of hist must exceed rank
template <int MT> __device__ void block_find_bin(ModelSmem<MT> &sm, uint32_t rank, uint32_t &bin, uint32_t &below) {
    constexpr int PER = M_BINS / MT;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    __syncthreads(); // histogram complete
    uint32_t h[PER], s = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) { h[i] = sm.hist[t * PER + i]; s += h[i]; }
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
    bin = sm.found_bin; below = sm.found_below;
}

__device__ __forceinline__ uint64_t shr64(uint64_t v, int s) { return s >= 64 ? 0ull : v >> s; }

// the rank-th smallest (0-based) of key_at(0..n-1); all keys < 2^bits
template <int MT, class KeyAt> __device__ uint64_t block_select(ModelSmem<MT> &sm, uint64_t n, uint64_t rank, int bits, KeyAt key_at) {
    const int passes = bits <= 0 ? 0 : (bits + M_BITS - 1) / M_BITS;
    uint64_t prefix = 0; // the bits above the current window
    for (int p = passes - 1; p >= 0; --p) {
        const int shift = p * M_BITS;
        __syncthreads();
        for (int i = threadIdx.x; i < M_BINS; i += MT) sm.hist[i] = 0;
        __syncthreads();
        for (uint64_t i = threadIdx.x; i < n; i += MT) {
            const uint64_t k = key_at(i);
            if (shr64(k, shift + M_BITS) == prefix) atomicAdd(&sm.hist[(uint32_t)(k >> shift) & (M_BINS - 1)], 1u);
        }
        uint32_t bin, below;
        block_find_bin<MT>(sm, (uint32_t)rank, bin, below);
        rank -= below;
        prefix = (prefix << M_BITS) | bin;
    }
    return prefix;
}

template <int MT> __global__ __launch_bounds__(MT) void k_slot_model(const uint64_t *ev_off, const uint64_t *samp_off, const uint32_t *ev_len,
                                                                      const double *samples, uint32_t drop_first, PgSlotModel *out,
                                                                      PgSlotDwell *dwell) {
    __shared__ ModelSmem<MT> sm;
    const uint32_t s = blockIdx.x;
    const uint64_t e0 = ev_off[s], e1 = ev_off[s + 1];
    const uint64_t a0 = samp_off[e0], a1 = samp_off[e1];
    const uint64_t skip = (a1 > a0 && drop_first) ? 1 : 0; // `tail -n +2`: the file's first value never reaches datamash
    const uint64_t first = a0 + skip, n = a1 - first;
    uint32_t flags = 0;
    PgSlotModel m{};
    m.n = n;
    if (n > 0) {
        bool bad = false;
        const int64_t origin = pg_fixed8(samples[first], bad);
        // pass A: range and moments about the first value
        uint64_t mn = ~0ull, mx = 0, hh = 0, hl = 0, ll = 0;
        int64_t s1 = 0;
        bool wide = false;
        for (uint64_t i = threadIdx.x; i < n; i += MT) {
            const int64_t v = pg_fixed8(samples[first + i], bad);
            const uint64_t key = (uint64_t)v ^ (1ull << 63); // order-preserving
            mn = op_min(mn, key); mx = op_max(mx, key);
            const int64_t d = v - origin;
            const uint64_t ad = (uint64_t)(d < 0 ? -d : d);
            if (ad >= (uint64_t)PG_MODEL_MAX_DEV) wide = true;
            const uint64_t h = (ad >> PG_MODEL_LIMB_BITS) & ((1u << PG_MODEL_LIMB_BITS) - 1), l = ad & ((1u << PG_MODEL_LIMB_BITS) - 1);
            s1 += d; hh += h * h; hl += h * l; ll += l * l;
        }
        mn = block_reduce<MT>(sm, mn, op_min); mx = block_reduce<MT>(sm, mx, op_max);
        m.s1 = (int64_t)block_reduce<MT>(sm, (uint64_t)s1, op_sum);
        m.s2_hh = block_reduce<MT>(sm, hh, op_sum); m.s2_hl = block_reduce<MT>(sm, hl, op_sum); m.s2_ll = block_reduce<MT>(sm, ll, op_sum);
        const uint64_t fl = block_reduce<MT>(sm, (bad ? PG_MODEL_BAD_VALUE : 0u) | (wide ? PG_MODEL_BAD_SPREAD : 0u), [](uint64_t a, uint64_t b) { return a | b; });
        flags = (uint32_t)fl | (n > PG_MODEL_MAX_VALUES ? PG_MODEL_BAD_COUNT : 0u);
        m.origin = origin;
        if (!(flags & PG_MODEL_BAD_VALUE)) {
            const uint64_t spread = mx - mn;
            const int bits = spread ? 64 - __builtin_clzll(spread) : 0;
            auto key_at = [&](uint64_t i) {
                bool b2 = false;
                return ((uint64_t)pg_fixed8(samples[first + i], b2) ^ (1ull << 63)) - mn;
            };
            const uint64_t r_lo = (n - 1) / 2, r_hi = n / 2;
            const uint64_t k_lo = block_select<MT>(sm, n, r_lo, bits, key_at);
            uint64_t k_hi = k_lo;
            if (r_hi != r_lo) { // even count: the next order statistic is k_lo again or the smallest key above it
                uint64_t le = 0, gt = ~0ull;
                for (uint64_t i = threadIdx.x; i < n; i += MT) {
                    const uint64_t k = key_at(i);
                    if (k <= k_lo) le++; else gt = op_min(gt, k);
                }
                le = block_reduce<MT>(sm, le, op_sum); gt = block_reduce<MT>(sm, gt, op_min);
                if (le <= r_hi) k_hi = gt;
            }
            m.mid_lo = (int64_t)((k_lo + mn) ^ (1ull << 63));
            m.mid_hi = (int64_t)((k_hi + mn) ^ (1ull << 63));
        }
    }
    // dwell: awk prints one comma count per ';'-separated field: samples - 1 per event, and 0 for the empty last field
    PgSlotDwell dw{};
    const uint64_t nev = e1 - e0;
    if (nev > 0) {
        const uint64_t nd = nev + 1;
        auto dkey = [&](uint64_t i) { const uint32_t len = i < nev ? ev_len[e0 + i] : 0u; return (uint64_t)(len ? len - 1 : 0u); };
        uint64_t mx = 0;
        for (uint64_t i = threadIdx.x; i < nd; i += MT) mx = op_max(mx, dkey(i));
        mx = block_reduce<MT>(sm, mx, op_max);
        const int bits = mx ? 64 - __builtin_clzll(mx) : 0;
        const uint64_t r_lo = (nd - 1) / 2, r_hi = nd / 2;
        const uint64_t k_lo = block_select<MT>(sm, nd, r_lo, bits, dkey);
        uint64_t k_hi = k_lo;
        if (r_hi != r_lo) {
            uint64_t le = 0, gt = ~0ull;
            for (uint64_t i = threadIdx.x; i < nd; i += MT) {
                const uint64_t k = dkey(i);
                if (k <= k_lo) le++; else gt = op_min(gt, k);
            }
            le = block_reduce<MT>(sm, le, op_sum); gt = block_reduce<MT>(sm, gt, op_min);
            if (le <= r_hi) k_hi = gt;
        }
        dw.n = nd; dw.mid_lo = (uint32_t)k_lo; dw.mid_hi = (uint32_t)k_hi;
    }
    dw.flags = flags;
    if (threadIdx.x == 0) { out[s] = m; dwell[s] = dw; }
}

} // namespace

hipError_t pg_launch_slot_model(hipStream_t st, uint32_t n_slots, uint64_t n_samples, const uint64_t *ev_off, const uint64_t *samp_off,
                                const uint32_t *ev_len, const double *samples, uint32_t drop_first, PgSlotModel *out, PgSlotDwell *dwell) {
    if (n_slots == 0) return hipSuccess;
    // small files: four times as many workgroups per CU and a quarter of the barrier cost
    if (n_samples / n_slots < 8192)
        hipLaunchKernelGGL(k_slot_model<256>, dim3(n_slots), dim3(256), 0, st, ev_off, samp_off, ev_len, samples, drop_first, out, dwell);
    else
        hipLaunchKernelGGL(k_slot_model<1024>, dim3(n_slots), dim3(1024), 0, st, ev_off, samp_off, ev_len, samples, drop_first, out, dwell);
    return hipGetLastError();
}
