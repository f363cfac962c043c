// pg_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the gmove hot path.
//
// Pipeline per batch (reference lines are src/gmove.cpp of hiruna72/poregen):
//   k_walk_events   ss walk + event filters (lines 822-927), one wave per read, prefix sums by wave scans
//   k_sort_*        stable LSD radix sort of accepted events by k-mer slot: the deterministic stand-in for
//                   "first sample_limit events in PAF-line order, then event order" (lines 732, 891, 925-927)
//   k_slot_* / k_kept_meta / k_scan_*   per-slot counts, the sample_limit cut, output offsets
//   k_read_plan + k_read_stats          pA conversion, zero-fill, exact median and MAD (lines 754-771)
//   k_gather        window copy + normalisation of the kept events (lines 773-775, 928-944)
// All of this is HBM/LDS-bound integer and FP64 work: there is no contraction here, so no MFMA.
#include "pg_internal.h"
#include "pg_select.h"
#include <limits.h>

#define WAVE 64

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }

__device__ __forceinline__ uint64_t wave_incl_scan_u64(uint64_t v) {
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        uint64_t t = __shfl_up(v, d, WAVE);
        if (lane >= d) v += t;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        uint32_t t = __shfl_up(v, d, WAVE);
        if (lane >= d) v += t;
    }
    return v;
}

// =====================================================================================================
// k_walk_events: one wave per read, four reads per workgroup.
// =====================================================================================================

__device__ __forceinline__ uint8_t base_code(uint8_t ch, bool rna_read) {
    // DNA-oriented record: sequence as fetched, spelled with T. RNA-oriented record: the reference
    // replaces T by U before matching (gmove.cpp:815-817), so T and U are the same letter there.
    switch (ch) {
        case 'A': return 0;
        case 'C': return 1;
        case 'G': return 2;
        case 'T': return 3;
        case 'U': return rna_read ? 3 : 4;
        default: return 4;
    }
}

__device__ __forceinline__ void report_error(const PgWalkOut &O, uint32_t r, int code) {
    O.status[r] = code;
    int prev = atomicMin(&O.err[0], (int)r);
    (void)prev;
}

__global__ __launch_bounds__(256) void k_walk_events(PgDevBatch B, PgWalkParams W, PgWalkOut O) {
    const int lane = lane_id();
    const uint32_t r = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (r >= B.n_reads) return; // wave-uniform
    const uint64_t o0 = B.op_off[r];
    const uint32_t nops = (uint32_t)(B.op_off[r + 1] - o0);
    const uint64_t s0 = B.seq_off[r];
    const uint32_t slen = (uint32_t)(B.seq_off[r + 1] - s0);
    const uint64_t L = B.sig_off[r + 1] - B.sig_off[r];
    const int32_t ts = B.tstart[r], te = B.tend[r], qs = B.qstart[r];
    const uint32_t k = W.k;

    // everything this read owns in ev_slot starts out "not accepted"
    for (uint32_t i = lane; i < nops; i += WAVE) O.ev_slot[o0 + i] = PG_INVALID_SLOT;
    if (lane == 0) { O.n_match[r] = 0; O.status[r] = PGR_OK; }

    int status = PGR_OK;
    // gmove.cpp:752 assert(query_start < len); negative columns convert to huge size_t in the reference
    if (qs < 0 || ts < 0 || te < 0 || (uint64_t)qs >= L || L > 0x7fffffffull) status = PGR_ERR_NEG;
    const bool rna = ts > te;                       // gmove.cpp:793
    const int32_t st_k = rna ? te : ts, end_k = rna ? ts : te;
    if (status == PGR_OK && rna && !W.allow_rna) status = PGR_ERR_RNA; // gmove.cpp:795-798
    if (status == PGR_OK && slen < k) status = PGR_SKIPPED;            // gmove.cpp:806-808
    if (status != PGR_OK) {
        if (lane == 0) { if (status < 0) report_error(O, r, status); else O.status[r] = status; }
        return;
    }

    // ---- phase 1: the ss walk (gmove.cpp:831-871) as wave-wide prefix sums over the op list -------
    uint64_t raw_carry = (uint64_t)qs;   // i_raw
    uint32_t match_carry = 0;            // i_k_raw  (matched bases so far)
    uint64_t del_carry = 0;              // num_deletion
    uint32_t indel_carry = 0;            // entries pushed to indel_pos so far (interior only)
    int err = 0;
    for (uint32_t c = 0; c < nops; c += WAVE) {
        const uint32_t i = c + lane;
        const bool act = i < nops;
        const uint32_t n = act ? B.op_n[o0 + i] : 0u;
        const uint32_t t = act ? (uint32_t)B.op_t[o0 + i] : 3u;
        const bool is_m = act && t == 0, is_i = act && t == 1, is_d = act && t == 2;
        if (act && t > 2) err = PGR_ERR_OP;
        const uint64_t mm = __ballot(is_m), mi = __ballot(is_i || is_d);
        const uint32_t j = match_carry + (uint32_t)__popcll(mm & lanemask_lt());
        const uint32_t tix = indel_carry + (uint32_t)__popcll(mi & lanemask_lt());
        const uint64_t radv = (is_m || is_i) ? (uint64_t)n : 0ull;
        const uint64_t dadv = is_d ? (uint64_t)n : 0ull;
        const uint64_t rinc = wave_incl_scan_u64(radv), dinc = wave_incl_scan_u64(dadv);
        const uint64_t start = raw_carry + rinc - radv;
        if (is_m) {
            const uint64_t ik = (uint64_t)j + del_carry + dinc - dadv; // i_k at this op
            if (ik >= slen) err = PGR_ERR_SEQ_OVERRUN;
            else if (start + n > 0x7fffffffull) err = PGR_ERR_RANGE;
            else {
                const uint64_t src = rna ? (uint64_t)slen - 1 - ik : ik; // gmove.cpp:849-853
                O.m_start[o0 + j] = (uint32_t)start;                     // end_raw_idx[i_k_raw]
                O.m_len[o0 + j] = n;                                     // st_raw_idx - end_raw_idx
                O.m_base[o0 + j] = base_code(B.seq[s0 + src], rna);
            }
        }
        if (is_i || is_d) O.p_int[o0 + tix] = (int32_t)j;                // i_k - num_deletion == matched bases so far
        raw_carry += __shfl(rinc, WAVE - 1, WAVE);
        del_carry += __shfl(dinc, WAVE - 1, WAVE);
        match_carry += (uint32_t)__popcll(mm);
        indel_carry += (uint32_t)__popcll(mi);
    }
    const uint32_t n = match_carry;  // fastq_len after refinement (gmove.cpp:872)
    const uint32_t m = indel_carry;
    if (__ballot(err != 0)) {
        // lowest failing lane decides the code
        const uint64_t bm = __ballot(err != 0);
        const int code = __shfl(err, __ffsll((long long)bm) - 1, WAVE);
        if (lane == 0) report_error(O, r, code);
        return;
    }
    if (n < k) { if (lane == 0) report_error(O, r, PGR_ERR_SHORT); return; } // unsigned wrap at gmove.cpp:891
    if (lane == 0) O.n_match[r] = n;
    __threadfence_block(); // phase 2 reads m_* / p_int written by other lanes of this wave

    // ---- phase 2: the event loop (gmove.cpp:891-927), 64 events per iteration --------------------
    const int32_t *table = rna ? W.table_u : W.table_t;
    const int32_t M = W.pick_margin;
    const uint32_t n_ev = n - k + 1;
    for (uint32_t c = 0; c < n_ev; c += WAVE) {
        const uint32_t i = c + lane;
        if (i >= n_ev) continue;
        const uint32_t e = i + W.sig_move_offset;
        if (e >= n) continue; // end_raw_idx[e] == -1 (gmove.cpp:892-894)
        // k-mer of matched bases [i, i+k); RNA reads it reversed (gmove.cpp:883, 899)
        uint32_t code = 0; bool bad = false;
        for (uint32_t t = 0; t < k; ++t) {
            const uint8_t b = O.m_base[o0 + (rna ? i + k - 1 - t : i + t)];
            bad |= b > 3;
            code = (code << 2) | (b & 3u);
        }
        if (bad) continue;
        const int32_t slot = table[code];
        // pick_this_kmer (gmove.cpp:204-211) over indel_pos = [-st_k, interior..., end_k + M]
        const int32_t left = rna ? (int32_t)(n - i - k) : (int32_t)i;
        const int32_t X = left + (int32_t)k + M, Y = left - M;
        auto interior = [&](uint32_t u) -> int32_t { // sorted ascending in both orientations (gmove.cpp:877-882)
            return rna ? (int32_t)n - O.p_int[o0 + (m - 1 - u)] : O.p_int[o0 + u];
        };
        uint32_t lo = 0, hi = m; // first u with interior(u) >= X
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (interior(mid) >= X) hi = mid; else lo = mid + 1; }
        const int32_t prev = lo == 0 ? -st_k : interior(lo - 1);
        const bool pick = (lo < m) ? (prev <= Y) : (X <= end_k + M && prev <= Y);
        if (!pick) continue;
        const uint32_t len = O.m_len[o0 + e];
        if (len > W.max_dur || len < W.min_dur) continue; // gmove.cpp:916-921
        if (slot < 0) continue;                            // gmove.cpp:922-924
        // the window must be printable (gmove.cpp:928-944 is undefined otherwise)
        const uint32_t start = O.m_start[o0 + e];
        const uint64_t wend = (uint64_t)start + len + W.print_margin > L ? L : (uint64_t)start + len + W.print_margin;
        if (W.print_margin > start || wend <= (uint64_t)(start - W.print_margin)) { report_error(O, r, PGR_ERR_WINDOW); continue; }
        O.ev_slot[o0 + i] = (uint32_t)slot;
    }
}

// =====================================================================================================
// stable LSD radix sort, 8-bit digits, 256-thread workgroups, one tile = 4 waves x ROWS rows x 64 keys
// =====================================================================================================

// lanes holding the same digit (among valid lanes)
__device__ __forceinline__ uint64_t match_digit(uint32_t d, bool valid, int nbits) {
    uint64_t peers = __ballot(valid);
    for (int b = 0; b < nbits; ++b) {
        const bool bit = (d >> b) & 1u;
        const uint64_t mset = __ballot(valid && bit);
        peers &= bit ? mset : ~mset;
    }
    return peers;
}

__global__ __launch_bounds__(256) void k_sort_count(const uint32_t *__restrict__ keys, uint32_t n_scalar,
                                                    const uint32_t *__restrict__ n_ptr, uint32_t shift, int nbits,
                                                    uint32_t n_tiles, uint32_t *__restrict__ hist, uint32_t *__restrict__ wcnt) {
    __shared__ uint32_t cnt[4][256];
    const uint32_t tid = threadIdx.x, tile = blockIdx.x, w = tid >> 6;
    const int lane = lane_id();
    for (uint32_t i = tid; i < 1024; i += 256) (&cnt[0][0])[i] = 0;
    __syncthreads();
    const uint32_t n = n_ptr ? *n_ptr : n_scalar;
    const uint32_t mask = (1u << nbits) - 1u;
    const uint64_t base = (uint64_t)tile * PG_SORT_TILE + (uint64_t)w * PG_SORT_ROWS * WAVE;
    for (int row = 0; row < PG_SORT_ROWS; ++row) {
        const uint64_t idx = base + (uint64_t)row * WAVE + lane;
        bool valid = idx < n;
        const uint32_t key = valid ? keys[idx] : PG_INVALID_SLOT;
        valid = valid && key != PG_INVALID_SLOT;
        const uint32_t d = (key >> shift) & mask;
        const uint64_t peers = match_digit(d, valid, nbits);
        if (valid && lane == __ffsll((long long)peers) - 1) cnt[w][d] += (uint32_t)__popcll(peers);
    }
    __syncthreads();
    uint32_t sum = 0;
    for (uint32_t ww = 0; ww < 4; ++ww) {
        const uint32_t c = cnt[ww][tid];
        wcnt[((uint64_t)tile * 4 + ww) * 256 + tid] = c;
        sum += c;
    }
    hist[(uint64_t)tid * n_tiles + tile] = sum;
}

__global__ __launch_bounds__(64) void k_sort_scan(uint32_t *__restrict__ hist, uint32_t n_tiles, uint32_t *__restrict__ totals) {
    const uint32_t d = blockIdx.x;
    const int lane = lane_id();
    uint32_t run = 0;
    for (uint32_t c = 0; c < n_tiles; c += WAVE) {
        const uint32_t i = c + lane;
        const uint32_t v = i < n_tiles ? hist[(uint64_t)d * n_tiles + i] : 0u;
        const uint32_t inc = wave_incl_scan_u32(v);
        if (i < n_tiles) hist[(uint64_t)d * n_tiles + i] = run + inc - v;
        run += __shfl(inc, WAVE - 1, WAVE);
    }
    if (lane == 0) totals[d] = run;
}

__global__ __launch_bounds__(256) void k_sort_dbase(const uint32_t *__restrict__ totals, uint32_t *__restrict__ dbase,
                                                    uint32_t *__restrict__ count_out) {
    __shared__ uint32_t wsum[4];
    const uint32_t tid = threadIdx.x;
    const uint32_t v = totals[tid];
    const uint32_t inc = wave_incl_scan_u32(v);
    if (lane_id() == WAVE - 1) wsum[tid >> 6] = inc;
    __syncthreads();
    uint32_t off = 0;
    for (uint32_t w = 0; w < (tid >> 6); ++w) off += wsum[w];
    dbase[tid] = off + inc - v;
    if (tid == 255) *count_out = off + inc;
}

__global__ __launch_bounds__(256) void k_sort_scatter(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                      uint32_t n_scalar, const uint32_t *__restrict__ n_ptr, uint32_t shift,
                                                      int nbits, uint32_t n_tiles, const uint32_t *__restrict__ hist,
                                                      const uint32_t *__restrict__ dbase, const uint32_t *__restrict__ wcnt,
                                                      uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out) {
    __shared__ uint32_t wbase[4][256];
    const uint32_t tid = threadIdx.x, tile = blockIdx.x, w = tid >> 6;
    const int lane = lane_id();
    {
        uint32_t b = dbase[tid] + hist[(uint64_t)tid * n_tiles + tile];
        for (uint32_t ww = 0; ww < 4; ++ww) { wbase[ww][tid] = b; b += wcnt[((uint64_t)tile * 4 + ww) * 256 + tid]; }
    }
    __syncthreads();
    const uint32_t n = n_ptr ? *n_ptr : n_scalar;
    const uint32_t mask = (1u << nbits) - 1u;
    volatile uint32_t *mybase = wbase[w];
    const uint64_t base = (uint64_t)tile * PG_SORT_TILE + (uint64_t)w * PG_SORT_ROWS * WAVE;
    for (int row = 0; row < PG_SORT_ROWS; ++row) {
        const uint64_t idx = base + (uint64_t)row * WAVE + lane;
        bool valid = idx < n;
        const uint32_t key = valid ? keys[idx] : PG_INVALID_SLOT;
        valid = valid && key != PG_INVALID_SLOT;
        const uint32_t d = (key >> shift) & mask;
        const uint64_t peers = match_digit(d, valid, nbits);
        uint32_t b = 0;
        if (valid) b = mybase[d];                   // every peer reads the running base of its digit ...
        __builtin_amdgcn_wave_barrier();
        if (valid && lane == __ffsll((long long)peers) - 1) mybase[d] = b + (uint32_t)__popcll(peers); // ... then its leader advances it
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            const uint32_t dest = b + (uint32_t)__popcll(peers & lanemask_lt());
            keys_out[dest] = key;
            vals_out[dest] = vals ? vals[idx] : (uint32_t)idx;
        }
    }
}

// =====================================================================================================
// per-slot bookkeeping
// =====================================================================================================

__global__ __launch_bounds__(256) void k_slot_bounds(const uint32_t *__restrict__ skey, const uint32_t *__restrict__ m_ptr,
                                                     uint32_t *__restrict__ slot_start, uint32_t *__restrict__ slot_end) {
    const uint64_t pos = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t M = *m_ptr;
    if (pos >= M) return;
    const uint32_t s = skey[pos];
    if (pos == 0 || skey[pos - 1] != s) slot_start[s] = (uint32_t)pos;
    if (pos == M - 1 || skey[pos + 1] != s) slot_end[s] = (uint32_t)pos + 1;
}

__global__ __launch_bounds__(256) void k_slot_counts(const uint32_t *__restrict__ slot_start, const uint32_t *__restrict__ slot_end,
                                                     uint32_t n_slots, uint64_t *__restrict__ acc_cnt) {
    const uint32_t s = blockIdx.x * 256 + threadIdx.x;
    if (s < n_slots) acc_cnt[s] = (uint64_t)(slot_end[s] - slot_start[s]);
}

// single workgroup: the sample_limit cut (gmove.cpp:925-927, 945-950) and the output offsets
__global__ __launch_bounds__(1024) void k_slot_plan(const uint64_t *__restrict__ acc_cnt, const uint64_t *base, uint64_t *running,
                                                    uint32_t limit, uint32_t n_slots, uint64_t *__restrict__ keep,
                                                    uint64_t *__restrict__ ev_off, uint64_t *__restrict__ totals) {
    __shared__ uint64_t wsum[16];
    __shared__ uint64_t wfull[16];
    const uint32_t tid = threadIdx.x, w = tid >> 6;
    const int lane = lane_id();
    uint64_t carry = 0, full = 0;
    for (uint32_t c = 0; c < n_slots; c += 1024) {
        const uint32_t s = c + tid;
        uint64_t kp = 0, isfull = 0;
        if (s < n_slots) {
            const uint64_t cnt = acc_cnt[s], b = base[s];
            const uint64_t room = b >= limit ? 0 : (uint64_t)limit - b;
            kp = cnt < room ? cnt : room;
            isfull = (b + cnt >= limit) ? 1 : 0;
            if (running) running[s] = b + cnt;
            keep[s] = kp;
        }
        const uint64_t inc = wave_incl_scan_u64(kp);
        const uint64_t fsum = wave_incl_scan_u64(isfull);
        if (lane == WAVE - 1) { wsum[w] = inc; wfull[w] = fsum; }
        __syncthreads();
        uint64_t off = 0, tot = 0, ftot = 0;
        for (uint32_t ww = 0; ww < 16; ++ww) { if (ww < w) off += wsum[ww]; tot += wsum[ww]; ftot += wfull[ww]; }
        if (s < n_slots) ev_off[s] = carry + off + inc - kp;
        carry += tot; full += ftot;
        __syncthreads();
    }
    if (tid == 0) { ev_off[n_slots] = carry; totals[0] = carry; totals[1] = full; }
}

__global__ __launch_bounds__(256) void k_kept_meta(const uint32_t *__restrict__ skey, const uint32_t *__restrict__ sval,
                                                   const uint32_t *__restrict__ m_ptr, const uint32_t *__restrict__ slot_start,
                                                   const uint64_t *__restrict__ keep, const uint64_t *__restrict__ ev_off,
                                                   PgDevBatch B, PgWalkParams W, PgWalkOut O, PgKeptOut K) {
    const uint64_t pos = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (pos >= *m_ptr) return;
    const uint32_t s = skey[pos];
    const uint64_t rank = pos - slot_start[s];
    if (rank >= keep[s]) return;
    const uint64_t e = ev_off[s] + rank;
    const uint64_t g = sval[pos];
    uint32_t lo = 0, hi = B.n_reads; // the read that owns op index g: largest rd with op_off[rd] <= g
    while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (B.op_off[mid] <= g) lo = mid; else hi = mid - 1; }
    const uint32_t rd = lo;
    const uint64_t o0 = B.op_off[rd];
    const uint64_t gm = g + W.sig_move_offset; // the event's window is that of match i + sig_move_offset
    const uint32_t start = O.m_start[gm], len = O.m_len[gm];
    const uint64_t L = B.sig_off[rd + 1] - B.sig_off[rd];
    const uint32_t ws = start - W.print_margin; // validated in k_walk_events
    const uint64_t we64 = (uint64_t)start + len + W.print_margin;
    const uint32_t we = (uint32_t)(we64 > L ? L : we64);
    (void)o0;
    K.ev_len[e] = we - ws;
    K.ev_start[e] = ws;
    K.ev_read[e] = rd;
    if (K.read_needed) K.read_needed[rd] = 1;
}

// =====================================================================================================
// exclusive scan u32 -> u64 over n elements, out has n+1 entries
// =====================================================================================================
#define SCAN_CHUNK 4096

__global__ __launch_bounds__(256) void k_scan_partials(const uint32_t *__restrict__ in, uint64_t n_scalar,
                                                       const uint64_t *__restrict__ n_ptr, uint64_t *__restrict__ partial) {
    __shared__ uint64_t wsum[4];
    const uint64_t n = n_ptr ? *n_ptr : n_scalar;
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_CHUNK;
    uint64_t s = 0;
    for (uint32_t i = threadIdx.x; i < SCAN_CHUNK; i += 256) { const uint64_t a = base + i; if (a < n) s += in[a]; }
    const uint64_t inc = wave_incl_scan_u64(s);
    if (lane_id() == WAVE - 1) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(256) void k_scan_partials_scan(uint64_t *__restrict__ partial, uint32_t n_blocks) {
    __shared__ uint64_t wsum[4];
    uint64_t carry = 0;
    for (uint32_t c = 0; c < n_blocks; c += 256) {
        const uint32_t i = c + threadIdx.x;
        const uint64_t v = i < n_blocks ? partial[i] : 0;
        const uint64_t inc = wave_incl_scan_u64(v);
        if (lane_id() == WAVE - 1) wsum[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint64_t off = 0, tot = 0;
        for (uint32_t w = 0; w < 4; ++w) { if (w < (threadIdx.x >> 6)) off += wsum[w]; tot += wsum[w]; }
        if (i < n_blocks) partial[i] = carry + off + inc - v;
        carry += tot;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_scan_apply(const uint32_t *__restrict__ in, uint64_t n_scalar,
                                                    const uint64_t *__restrict__ n_ptr, const uint64_t *__restrict__ partial,
                                                    uint64_t *__restrict__ out) {
    __shared__ uint64_t wsum[4];
    const uint64_t n = n_ptr ? *n_ptr : n_scalar;
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_CHUNK + (uint64_t)threadIdx.x * 16;
    uint32_t v[16];
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const uint64_t a = base + i; v[i] = a < n ? in[a] : 0u; s += v[i]; }
    const uint64_t inc = wave_incl_scan_u64(s);
    if (lane_id() == WAVE - 1) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint64_t off = partial[blockIdx.x];
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) off += wsum[w];
    uint64_t run = off + inc - s;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint64_t a = base + i;
        if (a < n) out[a] = run;
        run += v[i];
        if (a + 1 == n) out[n] = run;
    }
    if (n == 0 && blockIdx.x == 0 && threadIdx.x == 0) out[0] = 0;
}

// =====================================================================================================
// read statistics: pA conversion + zero-fill + exact median / MAD (gmove.cpp:754-771)
// =====================================================================================================

__global__ __launch_bounds__(256) void k_read_plan(PgDevBatch B, double pa_min, double pa_max, const int32_t *__restrict__ status,
                                                   PgReadPlan *__restrict__ plan) {
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= B.n_reads) return;
    PgReadPlan p;
    if (status[r] != PGR_OK) { p.c_lo = 0; p.span = 0; p.z0 = 0; p.status = 1; } // skipped / failed reads are not scanned
    else p = pg_make_plan(B.dig[r], B.off[r], B.range[r], pa_min, pa_max);
    plan[r] = p;
}

// One workgroup per read. The signal is streamed once with 16-byte loads (8 int16 per lane) and
// binned by code into an LDS histogram of the in-range code interval; an inclusive prefix sum of
// the histogram then gives both order statistics (pg_select.h) without touching the signal again.
__global__ __launch_bounds__(256) void k_read_stats(PgDevBatch B, const PgReadPlan *__restrict__ plan,
                                                    const uint8_t *__restrict__ needed, double *__restrict__ med,
                                                    double *__restrict__ mad, int32_t *__restrict__ status, int32_t *__restrict__ err) {
    // TODO(wide): reads whose in-range code interval exceeds PG_STATS_BINS need the global-memory histogram path
    __shared__ uint32_t hist[PG_STATS_BINS + 32];
    __shared__ uint32_t wsum[4];
    const uint32_t r = blockIdx.x, tid = threadIdx.x;
    const PgReadPlan pl = plan[r];
    const bool skip = pl.status == 1 || (needed && !needed[r]);
    if (skip || pl.status != 0 || pl.span > PG_STATS_BINS) {
        if (tid == 0) {
            med[r] = __builtin_nan(""); mad[r] = __builtin_nan("");
            if (!skip) { status[r] = pl.status != 0 ? PGR_ERR_SCALE : PGR_ERR_WIDE; atomicMin(&err[0], (int)r); }
        }
        return;
    }
    for (uint32_t i = tid; i < PG_STATS_BINS + 32; i += 256) hist[i] = 0;
    __syncthreads();

    const uint64_t beg = B.sig_off[r], end = B.sig_off[r + 1];
    const int c_lo = pl.c_lo;
    const uint32_t span = (uint32_t)pl.span;
    const uint32_t trash = PG_STATS_BINS + (tid & 31u); // out-of-range samples: spread over 32 dummy bins
    auto bin = [&](int code) {
        const uint32_t idx = (uint32_t)(code - c_lo);
        atomicAdd(&hist[idx < span ? idx : trash], 1u);
    };
    const int16_t *__restrict__ sig = B.sig;
    const uint64_t v0 = beg >> 3, v1 = (end + 7) >> 3; // 16-byte vectors that overlap [beg, end)
    for (uint64_t v = v0 + tid; v < v1; v += 256) {
        const uint64_t s0 = v << 3;
        if (s0 >= beg && s0 + 8 <= end) {
            const int4 q = *reinterpret_cast<const int4 *>(sig + s0);
            bin((int)(short)(q.x & 0xffff)); bin(q.x >> 16);
            bin((int)(short)(q.y & 0xffff)); bin(q.y >> 16);
            bin((int)(short)(q.z & 0xffff)); bin(q.z >> 16);
            bin((int)(short)(q.w & 0xffff)); bin(q.w >> 16);
        } else {
            for (int e = 0; e < 8; ++e) { const uint64_t a = s0 + e; if (a >= beg && a < end) bin((int)sig[a]); }
        }
    }
    __syncthreads();

    // inclusive prefix over the PG_STATS_BINS bins: 8 consecutive bins per thread
    uint32_t h[8];
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { s += hist[tid * 8 + i]; h[i] = s; }
    const uint32_t inc = wave_incl_scan_u32(s);
    if (lane_id() == WAVE - 1) wsum[tid >> 6] = inc;
    __syncthreads();
    uint32_t off = inc - s;
    for (uint32_t w = 0; w < (tid >> 6); ++w) off += wsum[w];
#pragma unroll
    for (int i = 0; i < 8; ++i) hist[tid * 8 + i] = off + h[i];
    __syncthreads();

    if (tid == 0) {
        const double scale = B.range[r] / B.dig[r];
        const PgMedMad mm = pg_medmad_from_prefix((const uint32_t *)hist, pl, end - beg, B.off[r], scale);
        med[r] = mm.med;
        mad[r] = mm.mad;
    }
}

// =====================================================================================================
// k_gather: one wave per kept event (gmove.cpp:773-775, 938-944)
// =====================================================================================================
__global__ __launch_bounds__(256) void k_gather(PgDevBatch B, uint64_t n_kept, const uint32_t *__restrict__ ev_len,
                                                const uint32_t *__restrict__ ev_read, const uint32_t *__restrict__ ev_start,
                                                const uint64_t *__restrict__ samp_off, int scaling, double pa_min, double pa_max,
                                                const double *__restrict__ med, const double *__restrict__ mad,
                                                double *__restrict__ samples) {
    const int lane = lane_id();
    const uint64_t stride = (uint64_t)gridDim.x * 4;
    for (uint64_t e = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); e < n_kept; e += stride) {
        const uint32_t rd = ev_read[e], len = ev_len[e];
        const uint64_t src = B.sig_off[rd] + ev_start[e], dst = samp_off[e];
        const double offset = B.off[rd], scale = B.range[rd] / B.dig[rd];
        const double md = scaling ? med[rd] : 0.0, ma = scaling ? mad[rd] : 1.0;
        for (uint32_t t = lane; t < len; t += WAVE) {
            const double pA = ((double)B.sig[src + t] + offset) * scale; // TO_PICOAMPS, poregen.h:30
            double x = (pA < pa_min || pA > pa_max) ? 0.0 : pA;          // gmove.cpp:756-759
            if (scaling) x = (x - md) / ma;                              // gmove.cpp:774
            samples[dst + t] = x;
        }
    }
}

// =====================================================================================================
// launchers
// =====================================================================================================

void pg_launch_walk_events(hipStream_t st, const PgDevBatch &B, const PgWalkParams &W, const PgWalkOut &O) {
    if (B.n_reads == 0) return;
    hipLaunchKernelGGL(k_walk_events, dim3((B.n_reads + 3) / 4), dim3(256), 0, st, B, W, O);
}

int pg_launch_sort_events(hipStream_t st, const uint32_t *ev_slot, uint64_t n, uint32_t key_bits, const PgSortBufs &S) {
    const uint32_t n_tiles = (uint32_t)((n + PG_SORT_TILE - 1) / PG_SORT_TILE);
    if (key_bits == 0) key_bits = 1;
    const uint32_t passes = (key_bits + 7) / 8;
    const uint32_t *kin = ev_slot;
    const uint32_t *vin = nullptr;
    int out = 0;
    for (uint32_t p = 0; p < passes; ++p) {
        const uint32_t shift = p * 8;
        const int nbits = (int)((key_bits - shift) < 8 ? (key_bits - shift) : 8);
        const uint32_t *n_ptr = p == 0 ? nullptr : S.count;
        hipLaunchKernelGGL(k_sort_count, dim3(n_tiles), dim3(256), 0, st, kin, (uint32_t)n, n_ptr, shift, nbits, n_tiles, S.hist, S.wcnt);
        hipLaunchKernelGGL(k_sort_scan, dim3(256), dim3(64), 0, st, S.hist, n_tiles, S.totals);
        hipLaunchKernelGGL(k_sort_dbase, dim3(1), dim3(256), 0, st, (const uint32_t *)S.totals, S.dbase, S.count + 1);
        hipLaunchKernelGGL(k_sort_scatter, dim3(n_tiles), dim3(256), 0, st, kin, vin, (uint32_t)n, n_ptr, shift, nbits, n_tiles,
                           (const uint32_t *)S.hist, (const uint32_t *)S.dbase, (const uint32_t *)S.wcnt, S.keys[out], S.vals[out]);
        // count[0] = number of keys for the next pass (pass 0 drops the invalid ones; later passes keep all)
        (void)hipMemcpyAsync(S.count, S.count + 1, sizeof(uint32_t), hipMemcpyDeviceToDevice, st);
        kin = S.keys[out]; vin = S.vals[out];
        out ^= 1;
    }
    return out ^ 1;
}

void pg_launch_slot_bounds(hipStream_t st, const uint32_t *skey, const uint32_t *m_ptr, uint64_t n_upper, uint32_t *slot_start,
                           uint32_t *slot_end, uint32_t n_slots, uint64_t *acc_cnt) {
    (void)hipMemsetAsync(slot_start, 0, sizeof(uint32_t) * n_slots, st);
    (void)hipMemsetAsync(slot_end, 0, sizeof(uint32_t) * n_slots, st);
    if (n_upper) hipLaunchKernelGGL(k_slot_bounds, dim3((uint32_t)((n_upper + 255) / 256)), dim3(256), 0, st, skey, m_ptr, slot_start, slot_end);
    hipLaunchKernelGGL(k_slot_counts, dim3((n_slots + 255) / 256), dim3(256), 0, st, (const uint32_t *)slot_start,
                       (const uint32_t *)slot_end, n_slots, acc_cnt);
}

void pg_launch_slot_plan(hipStream_t st, const uint64_t *acc_cnt, const uint64_t *base, uint64_t *running, uint32_t limit,
                         uint32_t n_slots, uint64_t *keep, uint64_t *ev_off, uint64_t *totals) {
    hipLaunchKernelGGL(k_slot_plan, dim3(1), dim3(1024), 0, st, acc_cnt, base, running, limit, n_slots, keep, ev_off, totals);
}

void pg_launch_kept_meta(hipStream_t st, const uint32_t *skey, const uint32_t *sval, const uint32_t *m_ptr, uint64_t n_upper,
                         const uint32_t *slot_start, const uint64_t *keep, const uint64_t *ev_off, const PgDevBatch &B,
                         const PgWalkParams &W, const PgWalkOut &O, const PgKeptOut &K) {
    if (!n_upper) return;
    hipLaunchKernelGGL(k_kept_meta, dim3((uint32_t)((n_upper + 255) / 256)), dim3(256), 0, st, skey, sval, m_ptr, slot_start, keep,
                       ev_off, B, W, O, K);
}

void pg_launch_scan_u32_u64(hipStream_t st, const uint32_t *in, uint64_t n, uint64_t *out, uint64_t *scratch) {
    const uint32_t nb = (uint32_t)((n + SCAN_CHUNK - 1) / SCAN_CHUNK);
    const uint32_t nbl = nb ? nb : 1;
    hipLaunchKernelGGL(k_scan_partials, dim3(nbl), dim3(256), 0, st, in, n, (const uint64_t *)nullptr, scratch);
    hipLaunchKernelGGL(k_scan_partials_scan, dim3(1), dim3(256), 0, st, scratch, nbl);
    hipLaunchKernelGGL(k_scan_apply, dim3(nbl), dim3(256), 0, st, in, n, (const uint64_t *)nullptr, (const uint64_t *)scratch, out);
}

void pg_launch_read_stats(hipStream_t st, const PgDevBatch &B, double pa_min, double pa_max, const uint8_t *read_needed,
                          void *plan_buf, double *med, double *mad, int32_t *status, int32_t *err) {
    if (B.n_reads == 0) return;
    PgReadPlan *plan = reinterpret_cast<PgReadPlan *>(plan_buf); // 16 bytes per read
    hipLaunchKernelGGL(k_read_plan, dim3((B.n_reads + 255) / 256), dim3(256), 0, st, B, pa_min, pa_max, (const int32_t *)status, plan);
    hipLaunchKernelGGL(k_read_stats, dim3(B.n_reads), dim3(256), 0, st, B, (const PgReadPlan *)plan, read_needed, med, mad, status, err);
}

void pg_launch_gather(hipStream_t st, const PgDevBatch &B, uint64_t n_kept, const uint32_t *ev_len, const uint32_t *ev_read,
                      const uint32_t *ev_start, const uint64_t *samp_off, int scaling, double pa_min, double pa_max,
                      const double *med, const double *mad, double *samples) {
    if (n_kept == 0) return;
    uint64_t blocks = (n_kept + 3) / 4;
    if (blocks > 256ull * 32) blocks = 256ull * 32;
    hipLaunchKernelGGL(k_gather, dim3((uint32_t)blocks), dim3(256), 0, st, B, n_kept, ev_len, ev_read, ev_start, samp_off, scaling,
                       pa_min, pa_max, med, mad, samples);
}
