// pg_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the gmove hot path.
//
// Pipeline per batch (reference lines are src/gmove.cpp of hiruna72/poregen):
//   k_batch_init    per-read records (walk head 822-830, statistics plan 754-760), classification of the reads, flag resets
//   k_walk          reads with I / D ops only: ss walk (lines 822-871) + their event loop, one wave per listed read
//   k_events        op-parallel: the events of the match-only reads (lines 891-927, 204-211), 16 consecutive ops per thread,
//                   and the per-(tile, slot) counts of the direct ranking
//   k_rank_*        stable ranking of accepted events inside their k-mer slot: the deterministic stand-in for
//                   "first sample_limit events in PAF-line order, then event order" (lines 732, 891, 925-927);
//                   direct for <= 1024 slots (k_rank_scan with the sample_limit cut in its last workgroup / k_rank_emit),
//                   LSD radix sort (k_rank_count / k_sort_* / k_kept_meta) beyond
//   k_slot_plan (+ k_tile_max) or k_slot_keep, k_scan_*   the cut as a launch of its own (base from other ranks), output offsets
//   k_read_stats (+ the rare workers riding in k_scan_chained)   pA conversion, zero-fill, exact median and MAD (lines 754-771)
//   k_gather        window copy + normalisation of the kept events (lines 773-775, 928-944)
// All of this is HBM/LDS-bound integer and FP64 work: there is no contraction here, so no MFMA.
#include "pg_dev.h"
#include "pg_select.h"
#include <limits.h>
#include <type_traits>

#ifdef PG_PHASE_PROBE
// Measurement build only (make variant NAME=phase EXTRA=-DPG_PHASE_PROBE; tools/phase_probe.py): per-wave time stamps (s_memtime)
// between the phases of the two large kernels. A wave keeps its phase times in registers (PG_MARK(k, i): wait for the wave's
// outstanding memory operations, then the time since the previous mark goes to phase i) and stores them once, at its end, into
// its own record g_pg_phase[kernel][wave][phase] -- no atomics, nothing shared between waves.
#define PG_PROBE_WAVES 65536
__device__ unsigned long long g_pg_phase[3][PG_PROBE_WAVES][8]; // 0 k_read_stats, 1 k_events, 2 k_rank_emit
extern "C" void pg_debug_phases(unsigned long long *out, int kernel, int reset) { // out: [PG_PROBE_WAVES][8]
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pg_phase), sizeof(unsigned long long) * PG_PROBE_WAVES * 8, sizeof(unsigned long long) * PG_PROBE_WAVES * 8 * kernel);
    if (reset) { void *p = nullptr; (void)hipGetSymbolAddress(&p, HIP_SYMBOL(g_pg_phase)); (void)hipMemset(p, 0, sizeof(unsigned long long) * 3 * PG_PROBE_WAVES * 8); }
}
#define PG_PROBE_BEGIN(k) unsigned long long pg_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long pg_t_ = __builtin_readcyclecounter(); const unsigned long long pg_t0_ = pg_t_
#define PG_MARK(k, i) do { __builtin_amdgcn_s_waitcnt(0); const unsigned long long n_ = __builtin_readcyclecounter(); pg_acc_[i] += n_ - pg_t_; pg_t_ = n_; } while (0)
#define PG_PROBE_END(k, wave) do { if (lane_id() == 0 && (wave) < PG_PROBE_WAVES) { pg_acc_[7] = __builtin_readcyclecounter() - pg_t0_; for (int i_ = 0; i_ < 8; ++i_) g_pg_phase[k][wave][i_] = pg_acc_[i_]; } } while (0)
#define PG_PROBE_PARAM , unsigned long long &pg_t_, unsigned long long (&pg_acc_)[8]
#define PG_PROBE_ARG , pg_t_, pg_acc_
#else
#define PG_PROBE_BEGIN(k) do {} while (0)
#define PG_MARK(k, i) do {} while (0)
#define PG_PROBE_END(k, wave) do {} while (0)
#define PG_PROBE_PARAM
#define PG_PROBE_ARG
#endif

// smallest t in [0, n) for which the monotone predicate holds, n if none. All 64 lanes call it together;
// every round tests 64 candidates at once (two rounds cover n <= 4096).
template <class Pred> __device__ __forceinline__ int wave_first_true(int n, Pred pred) {
    const int lane = lane_id();
    int lo = 0, len = n;
    while (len > 0) {
        const int step = (len + WAVE - 1) / WAVE;
        int t = lo + (lane + 1) * step - 1;
        if (t > lo + len - 1) t = lo + len - 1;
        const uint64_t m = __ballot(pred(t));
        if (m == 0) return n; // only possible in the first round
        const int l = __ffsll((long long)m) - 1;
        if (step == 1) return lo + l;
        lo += l * step;
        len = (lo + step > n) ? n - lo : step;
    }
    return n;
}

// two independent searches at once: lanes 0..31 look for the first true t of side 0 in [0,n0), lanes 32..63
// of side 1 in [0,n1); 32 candidates per side and round (two rounds cover 1024)
template <class Pred> __device__ __forceinline__ void wave_first_true_pair(int n0, int n1, Pred pred, int &r0, int &r1) {
    const int lane = lane_id(), half = lane >> 5, sub = lane & 31;
    const int n = half ? n1 : n0;
    int lo = 0, len = n, res = n;
    bool done = n == 0;
    for (int round = 0; round < 4; ++round) {
        const int step = (len + 31) / 32;
        int t = lo + (sub + 1) * step - 1;
        if (t > lo + len - 1) t = lo + len - 1;
        const bool p = !done && pred(half, t);
        const uint64_t m = __ballot(p);
        const uint32_t mh = half ? (uint32_t)(m >> 32) : (uint32_t)m;
        if (!done) {
            if (mh == 0) { res = n; done = true; } // only possible in the first round
            else {
                const int l = __ffs((int)mh) - 1;
                if (step == 1) { res = lo + l; done = true; }
                else { lo += l * step; len = (lo + step > n) ? n - lo : step; }
            }
        }
        if (__ballot(!done) == 0) break;
    }
    r0 = __builtin_amdgcn_readlane(res, 0);
    r1 = __builtin_amdgcn_readlane(res, 32);
}

// =====================================================================================================
// Events of a batch (gmove.cpp:822-927), two ways:
//   * k_events, op-parallel, for the reads whose ss string holds matches only ("direct" reads: every ss string `reform`
//     writes, and BASELINE's synthetic workloads): match index = op index = base index, so the slot of event i depends on
//     k bases, one window length and the read's scalars only -- no prefix sums, no per-read dependency chain. A thread takes
//     4 consecutive op indices of the whole batch. The window STARTS (prefix sums of op_n) are needed for kept events only:
//     every wave leaves the sums over its 256-op block (cum / btot) and the emit kernels evaluate them.
//   * k_walk, one wave per read, for everything else (I / D ops, reads the reference treats as errors): the ss walk with
//     DPP wave scans followed by the event loop of the same read. A persistent grid strides over the list of such reads.
// k_batch_init writes the per-read records and classifies the reads (its op-parallel part looks for non-match ops).
// =====================================================================================================

__device__ __forceinline__ uint8_t base_code(uint8_t ch, bool rna_read) {
    // DNA-oriented record: sequence as fetched, spelled with T. RNA-oriented record: the reference
    // replaces T by U before matching (gmove.cpp:815-817), so T and U are the same letter there.
    // Branch-free: letter index ch - 'A' into a bit set of the valid letters and a packed table of their 2-bit codes
    // (A = 0, C = 1, G = 2, T = 3, U = 3 on RNA-oriented records only); everything else is 4.
    const uint32_t idx = (uint32_t)ch - (uint32_t)'A';
    const uint32_t valid = (1u << 0) | (1u << 2) | (1u << 6) | (1u << 19) | (rna_read ? (1u << 20) : 0u);
    const uint64_t codes = (1ull << (2 * 2)) | (2ull << (2 * 6)) | (3ull << (2 * 19)) | (3ull << (2 * 20));
    const bool ok = idx < 32u && ((valid >> (idx & 31u)) & 1u);
    return ok ? (uint8_t)((codes >> (2u * (idx & 31u))) & 3u) : (uint8_t)4;
}

__device__ __forceinline__ void list_generic(const PgWalkOut &O, uint32_t r) {
    if (atomicExch(&O.gen_flag[r], O.batch_id) != O.batch_id) O.gen_list[atomicAdd(&O.gen_count[O.batch_id & 1u], 1u)] = r;
}

// One read's record, checks and class (k_batch_init, one thread per read). The checks are those at the head of the reference's
// loop body: gmove.cpp:752 (assert), 792-798 (orientation), 806-808 (short fetch).
__device__ __forceinline__ void read_head_one(const PgDevBatch &B, const PgWalkParams &W, const PgWalkOut &O, uint32_t r, int force_generic) {
    const uint64_t o0 = B.op_off[r], o1 = B.op_off[r + 1], s0 = B.seq_off[r], s1 = B.seq_off[r + 1];
    const uint64_t L = B.sig_off[r + 1] - B.sig_off[r];
    const int32_t ts = B.tstart[r], te = B.tend[r], qs = B.qstart[r];
    const bool rna = ts > te; // gmove.cpp:793
    const bool layout_ok = o1 >= o0 && o1 <= B.n_ops && s1 >= s0 && (r + 1 < B.n_reads || o1 == B.n_ops);
    PgReadMeta mt;
    mt.o0 = o0 < B.n_ops ? o0 : B.n_ops; mt.s0 = s0;
    mt.nops = layout_ok ? (uint32_t)(o1 - o0) : 0u;
    mt.slen = (uint32_t)(s1 - s0 > 0xffffffffull ? 0xffffffffull : s1 - s0);
    mt.n = 0; mt.m = 0; mt.st_k = rna ? te : ts; mt.end_k = rna ? ts : te;
    mt.L = (uint32_t)(L > 0xffffffffull ? 0xffffffffull : L); mt.qs = qs; mt.opsum0 = 0; mt.sig0 = B.sig_off[r];
    int status = PGR_OK;
    if (!layout_ok) status = PGR_ERR_LAYOUT;
    // gmove.cpp:752 assert(query_start < len); negative columns convert to huge size_t in the reference
    else if (qs < 0 || ts < 0 || te < 0 || (uint64_t)qs >= L || L > 0x7fffffffull) status = PGR_ERR_NEG;
    else if (rna && !W.allow_rna) status = PGR_ERR_RNA;  // gmove.cpp:795-798
    else if (mt.slen < W.k) status = PGR_SKIPPED;        // gmove.cpp:806-808
    if (status < 0) report_error(O, r, status); else O.status[r] = status;
    const bool live = status == PGR_OK;
    const bool direct_ok = live && !force_generic && mt.nops >= W.k && mt.nops <= mt.slen;
    mt.flags = (rna ? PG_RM_RNA : 0u) | (live ? PG_RM_LIVE : 0u) | (direct_ok ? PG_RM_DIRECT_OK : 0u);
    if (direct_ok) mt.n = mt.nops;
    O.meta[r] = mt;
    if (live && !direct_ok) list_generic(O, r); // the generic walk produces its events or its error
    // owner index: this read owns every 64-op block whose first op is one of its ops
    if (layout_ok) for (uint64_t b = (o0 + 63) >> 6; (b << 6) < o1; ++b) O.blk_read[b] = r;
}

// ---------------------------------------------------------------------------------------------------------------------
// k_walk: the generic walk (gmove.cpp:831-871) and event loop (891-927) of one read by one wave
// ---------------------------------------------------------------------------------------------------------------------
#define PG_EV_PER_THREAD 4 // four consecutive op indices per thread / lane
// LDS window of a read's per-match values: what the event loop needs besides the window starts -- base code, window length, I/D ops
// in front. A read of at most PG_WALK_LDS_OPS ss ops never writes them to global memory; a longer one re-fills the window per tile.
#define PG_WALK_LDS_OPS 512
struct WalkLds {
    uint32_t code[PG_WALK_LDS_OPS / 16 + 2]; // 2-bit base codes, match p at bits 2(p & 15) of word p >> 4 (OR-ed in: zeroed first)
    uint32_t bad[PG_WALK_LDS_OPS / 32 + 2];  // one bit per match: its base is not one of A C G T/U
    uint32_t len[PG_WALK_LDS_OPS + 16];      // read at i + sig_move_offset (< n whenever it is used)
    uint32_t tix[PG_WALK_LDS_OPS + 4];       // + 4: lanes behind the read's last op still form their (unused) addresses
};
template <int E, bool TIXG> __device__ __forceinline__ void walk_events_tile(const PgWalkParams &W, const PgWalkOut &O, const WalkLds *sm, uint64_t o0, uint32_t nops,
                                                                  uint32_t n, uint32_t m, bool rna, int32_t st_k, int32_t end_k, int lane, uint32_t base, uint32_t T0);

// a read without events (failed, fewer than k matches): every op index of it still carries a slot entry
__device__ __forceinline__ void walk_no_events(const PgWalkOut &O, uint64_t o0, uint32_t nops, int lane) {
    for (uint32_t i = lane; i < nops; i += WAVE) O.ev_slot[o0 + i] = PG_INVALID_SLOT;
}

// TIXG: the pick margin is too wide for the LDS window (> 100): the event loop reads the I/D counts from the global array
template <bool TIXG> __device__ __forceinline__ void walk_one_read(WalkLds *sm, const PgDevBatch &B, const PgWalkParams &W, const PgWalkOut &O, uint32_t r) {
    const int lane = lane_id();
    // k_batch_init's record: offsets, lengths, PAF columns. The kernel also writes records (n, m of the reads it walks), so the
    // compiler loads this one per lane: every field goes back to a scalar register (the whole walk is written around uniform values)
    const PgReadMeta *mp = O.meta + r;
    auto sc32 = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    auto sc64 = [&](uint64_t v) { return (uint64_t)sc32((uint32_t)v) | ((uint64_t)sc32((uint32_t)(v >> 32)) << 32); };
    const uint64_t o0 = sc64(mp->o0), s0 = sc64(mp->s0);
    const uint32_t nops = sc32(mp->nops), slen = sc32(mp->slen), k = W.k, flags = sc32(mp->flags);
    const bool rna = (flags & PG_RM_RNA) != 0;
    const int32_t st_k = (int32_t)sc32((uint32_t)mp->st_k), end_k = (int32_t)sc32((uint32_t)mp->end_k), qs = (int32_t)sc32((uint32_t)mp->qs);
    if (!(flags & PG_RM_LIVE)) { walk_no_events(O, o0, nops, lane); return; }
    if (O.oor && O.oor[r]) { walk_no_events(O, o0, nops, lane); return; } // SAM/BAM front-end (k_apply_oor has marked the read)

    // the read's own stretch of every per-op array: a uniform base pointer + a 32-bit lane offset per access
    const uint32_t *__restrict__ r_op_n = B.op_n + o0; const uint8_t *__restrict__ r_op_t = B.op_t + o0; const uint8_t *__restrict__ r_seq = B.seq + s0;
    uint32_t *__restrict__ r_start = O.m_start + o0, *__restrict__ r_len = O.m_len + o0, *__restrict__ r_tix = O.m_tix + o0;
    uint8_t *__restrict__ r_base = O.m_base + o0;
    const bool in_lds = nops <= PG_WALK_LDS_OPS; // wave-uniform
    if (in_lds) {
        if (lane < PG_WALK_LDS_OPS / 16 + 2) sm->code[lane] = 0;
        if (lane < PG_WALK_LDS_OPS / 32 + 2) sm->bad[lane] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    uint64_t raw_carry = (uint64_t)(uint32_t)qs; // i_raw
    uint32_t match_carry = 0;            // i_k_raw  (matched bases so far)
    uint64_t del_carry = 0;              // num_deletion
    uint32_t indel_carry = 0;            // entries pushed to indel_pos so far (interior only)
    int err = 0;
    // Four chunks of 64 ops per trip: all their op loads are in flight together, then the (ALU-only) wave scans run
    // chunk after chunk, then all sequence loads, then the stores -- three dependent round trips per 256 ops.
    constexpr int U = 4;
    for (uint32_t c = 0; c < nops; c += U * WAVE) {
        uint32_t nn[U], tt[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t i = c + u * WAVE + lane;
            nn[u] = i < nops ? r_op_n[i] : 0u;
            tt[u] = i < nops ? (uint32_t)r_op_t[i] : 3u;
        }
        uint32_t jj[U], tix[U], st32[U]; uint64_t ik[U]; bool okm[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            jj[u] = 0; tix[u] = 0; st32[u] = 0; ik[u] = 0; okm[u] = false;
            if (c + u * WAVE >= nops) continue; // wave-uniform: a chunk behind the read's last op costs nothing
            const uint32_t i = c + u * WAVE + lane;
            const bool act = i < nops;
            const uint32_t n = nn[u], t = tt[u];
            const bool is_m = act && t == 0, is_i = act && t == 1, is_d = act && t == 2;
            if (act && t > 2) err = PGR_ERR_OP;
            if (n >= PG_OP_N_LIMIT) err = PGR_ERR_RANGE; // keeps the 32-bit chunk scans (and k_events' block sums) exact
            const uint64_t mm = __ballot(is_m), mi = __ballot(is_i || is_d);
            jj[u] = match_carry + (uint32_t)__popcll(mm & lanemask_lt());
            tix[u] = indel_carry + (uint32_t)__popcll(mi & lanemask_lt());
            const uint32_t radv = (is_m || is_i) ? (n & (PG_OP_N_LIMIT - 1u)) : 0u;
            const uint32_t dadv = is_d ? (n & (PG_OP_N_LIMIT - 1u)) : 0u;
            const uint32_t rinc = wave_incl_scan_u32(radv);
            const uint32_t dinc = __ballot(is_d) ? wave_incl_scan_u32(dadv) : 0u; // most chunks hold no deletion
            const uint64_t start = raw_carry + rinc - radv;
            ik[u] = (uint64_t)jj[u] + del_carry + dinc - dadv; // i_k at this op
            if (is_m) {
                if (ik[u] >= slen) err = PGR_ERR_SEQ_OVERRUN;
                else if (start + n > 0x7fffffffull) err = PGR_ERR_RANGE;
                else okm[u] = true;
            }
            st32[u] = (uint32_t)start;
            raw_carry += (uint32_t)__builtin_amdgcn_readlane((int)rinc, WAVE - 1);
            del_carry += (uint32_t)__builtin_amdgcn_readlane((int)dinc, WAVE - 1);
            match_carry += (uint32_t)__popcll(mm);
            indel_carry += (uint32_t)__popcll(mi);
        }
        uint8_t bc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t src = rna ? slen - 1u - (uint32_t)ik[u] : (uint32_t)ik[u]; // gmove.cpp:849-853 (okm: ik < slen)
            bc[u] = okm[u] ? base_code(r_seq[src], rna) : (uint8_t)4;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (okm[u]) {
                r_start[jj[u]] = st32[u]; // end_raw_idx[i_k_raw]
                r_len[jj[u]] = nn[u];     // st_raw_idx - end_raw_idx
                if (TIXG) r_tix[jj[u]] = tix[u];
                if (in_lds) {
                    atomicOr(&sm->code[jj[u] >> 4], ((uint32_t)bc[u] & 3u) << (2u * (jj[u] & 15u)));
                    if (bc[u] > 3) atomicOr(&sm->bad[jj[u] >> 5], 1u << (jj[u] & 31u));
                    sm->len[jj[u]] = nn[u]; sm->tix[jj[u]] = tix[u];
                } else {
                    r_base[jj[u]] = bc[u];
                    if (!TIXG) r_tix[jj[u]] = tix[u]; // I/D ops in front of this match (the indel positions themselves, i_k -
                                                      // num_deletion at every I/D op, are only needed as these counts)
                }
            }
        }
    }
    const uint64_t bm = __ballot(err != 0);
    if (bm) { // lowest failing lane decides the code
        const int code = __shfl(err, __ffsll((long long)bm) - 1, WAVE);
        if (lane == 0) report_error(O, r, code);
        walk_no_events(O, o0, nops, lane);
        return;
    }
    if (match_carry < k) { // unsigned wrap at gmove.cpp:891 in the PAF path; simply no events for the move-table front-end
        if (lane == 0 && !W.short_ok) report_error(O, r, PGR_ERR_SHORT);
        walk_no_events(O, o0, nops, lane);
        return;
    }
    if (lane == 0) { O.meta[r].n = match_carry; O.meta[r].m = indel_carry; } // the emit kernels read them
    if (TIXG) __syncthreads(); // the event loop reads the I/D counts back from global memory
    if (in_lds) { // the whole read is in the window (base 0). LDS operations of one wave execute in order: only the compiler has to be held back
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // events per lane: the fewest trips first, then the most lanes at work
        const uint32_t trips4 = (nops + 4 * WAVE - 1) / (4 * WAVE);
        if ((nops + 2 * WAVE - 1) / (2 * WAVE) == trips4) {
            for (uint32_t T0 = 0; T0 < nops; T0 += 2 * WAVE) walk_events_tile<2, TIXG>(W, O, sm, o0, nops, match_carry, indel_carry, rna, st_k, end_k, lane, 0u, T0);
        } else if ((nops + 3 * WAVE - 1) / (3 * WAVE) == trips4) {
            for (uint32_t T0 = 0; T0 < nops; T0 += 3 * WAVE) walk_events_tile<3, TIXG>(W, O, sm, o0, nops, match_carry, indel_carry, rna, st_k, end_k, lane, 0u, T0);
        } else {
            for (uint32_t T0 = 0; T0 < nops; T0 += 4 * WAVE) walk_events_tile<4, TIXG>(W, O, sm, o0, nops, match_carry, indel_carry, rna, st_k, end_k, lane, 0u, T0);
        }
    } else {
        // A longer read: tile after tile of 256 events, the window re-filled from the global arrays the wave has just
        // written. Window = matches [base, base + PG_WALK_LDS_OPS), base = T0 - min(T0, pick margin) -- the left bound of
        // pick_this_kmer reaches that far back; 256 + k + 2 * margin + 16 <= PG_WALK_LDS_OPS (wider margins: TIXG).
        __syncthreads(); // phase 1's global stores are complete and visible to all lanes of the wave
        const uint32_t Mu = TIXG ? 0u : (uint32_t)W.pick_margin;
        for (uint32_t T0 = 0; T0 < nops; T0 += PG_EV_PER_THREAD * WAVE) {
            const uint32_t base = T0 - (T0 < Mu ? T0 : Mu);
            if (lane < PG_WALK_LDS_OPS / 16 + 2) sm->code[lane] = 0;
            if (lane < PG_WALK_LDS_OPS / 32 + 2) sm->bad[lane] = 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (uint32_t q = lane; q < PG_WALK_LDS_OPS && base + q < match_carry; q += WAVE) {
                const uint32_t bcq = r_base[base + q];
                atomicOr(&sm->code[q >> 4], (bcq & 3u) << (2u * (q & 15u)));
                if (bcq > 3) atomicOr(&sm->bad[q >> 5], 1u << (q & 31u));
                sm->len[q] = r_len[base + q];
                if (!TIXG) sm->tix[q] = r_tix[base + q];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            walk_events_tile<PG_EV_PER_THREAD, TIXG>(W, O, sm, o0, nops, match_carry, indel_carry, rna, st_k, end_k, lane, base, T0);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// one one-wave workgroup per LISTED read: the grid is the number of reads (the host does not know the list's length), workgroups
// behind the list leave at once. (A persistent grid striding over the list compiles to 91 VGPRs instead of 48: the loop keeps the
// event loop's values alive across reads. A caller that knows its batch holds matches only says so -- PG_BATCH_ALL_MATCHES -- and
// the launch is skipped altogether.)
template <bool TIXG> __global__ __launch_bounds__(64) void k_walk(PgDevBatch B, PgWalkParams W, PgWalkOut O) {
    __shared__ WalkLds sm_store;
    if (blockIdx.x >= O.gen_count[O.batch_id & 1u]) return;
    walk_one_read<TIXG>(&sm_store, B, W, O, (uint32_t)__builtin_amdgcn_readfirstlane((int)O.gen_list[blockIdx.x]));
}

// pick_this_kmer (gmove.cpp:204-211) over indel_pos = [-st_k, interior..., end_k + M] without the list: with cp(Z) = #{interior
// entries p < Z} = I/D ops in front of match Z-1 (0 for Z <= 0, all m for Z > n), the entries strictly between the event's bounds
// number cp(i+k+M) - cp(i-M+1) in BOTH orientations (the RNA list is the mirrored one), and the search result `lo` of the reference's
// loop is cp(i+k+M) on DNA-oriented records and m - cp(i-M+1) on RNA-oriented ones. cA / cB: the two counts.
__device__ __forceinline__ bool pick_kmer(uint32_t i, uint32_t k, int32_t M, uint32_t n, uint32_t m, bool rna, int32_t st_k, int32_t end_k, uint32_t cA, uint32_t cB) {
    const int32_t left = rna ? (int32_t)(n - i - k) : (int32_t)i;
    const int32_t X = left + (int32_t)k + M, Y = left - M;
    const uint32_t lo = rna ? m - cA : cB;                       // first interior entry >= X
    const bool prev_ok = lo == 0 ? (-st_k <= Y) : (cA == cB);    // the entry in front of it is <= Y
    return prev_ok && (lo < m || X <= end_k + M);
}

// The event loop of the generic walk: 64 * E events of ONE read -- matches T0 + E*lane .. + E-1 -- by the wave that walked it, from the
// LDS window that holds the values of matches base .. base + PG_WALK_LDS_OPS - 1, with the read's summary (n matches, m I/D ops,
// orientation, target range) in scalar registers. E = events per lane (2, 3 or 4): a short read spreads its events over as many lanes
// as it can fill in one trip (130 ops: 44 lanes x 3 instead of 33 lanes x 4).
template <int E, bool TIXG> __device__ __forceinline__ void walk_events_tile(const PgWalkParams &W, const PgWalkOut &O, const WalkLds *sm, uint64_t o0, uint32_t nops,
                                                                  uint32_t n, uint32_t m, bool rna, int32_t st_k, int32_t end_k, int lane, uint32_t base, uint32_t T0) {
    static_assert(E >= 2 && E <= 4, "the 16 codes fetched per lane cover E - 1 + k <= 16 matches");
    const uint32_t k = W.k;
    const int32_t M = W.pick_margin;
    const uint32_t kM = k + (uint32_t)M;
    const int32_t *__restrict__ table = rna ? W.table_u : W.table_t;
    const uint32_t i0 = T0 + E * lane;
    if (i0 >= nops) return;
    const uint32_t l0 = i0 - base; // window index of match i0
    // the 2-bit codes of matches i0 .. i0+15 (match i0+p at bits 2p) and their "not ACGT/U" bits
    const uint32_t c2 = (uint32_t)(((uint64_t)sm->code[l0 >> 4] | ((uint64_t)sm->code[(l0 >> 4) + 1] << 32)) >> (2u * (l0 & 15u)));
    const uint32_t badbits = (uint32_t)(((uint64_t)sm->bad[l0 >> 5] | ((uint64_t)sm->bad[(l0 >> 5) + 1] << 32)) >> (l0 & 31u));
    const uint32_t *__restrict__ g_tix = O.m_tix + o0; // TIXG: readable from index -front (pg_api.hip pads both ends)
    uint32_t len[E], tx_lo[E], tx_hi[E];
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const int32_t a = (int32_t)(i0 + j) - M;                     // match whose I/D count bounds the event on the left: used when >= 0
        len[j] = sm->len[l0 + j + W.sig_move_offset];
        if (TIXG) {
            const uint32_t h = i0 + j + kM - 1;                      // ... on the right: used when the match exists (< n)
            tx_lo[j] = g_tix[a > 0 ? (uint32_t)a : 0u];
            tx_hi[j] = g_tix[h < n ? h : n - 1];
        } else {
            const uint32_t h = i0 + j + kM - 1 - base;
            tx_lo[j] = sm->tix[(a > 0 ? (uint32_t)a : 0u) - base];   // base <= max(i0 - M, 0): see the staging in walk_one_read
            tx_hi[j] = sm->tix[h < PG_WALK_LDS_OPS ? h : PG_WALK_LDS_OPS - 1];
        }
    }
    // the slot of the k-mer of matched bases [i, i+k) (mirrored on RNA-oriented records: gmove.cpp:883, 899)
    int32_t slot[E]; bool cand[E];
#pragma unroll
    for (int j = 0; j < E; ++j) {
        // the mirrored code (base t at bits 2t) is a bit field of c2; the forward one (first base most significant) is its
        // 2-bit groups in reverse order: reverse all bits, swap the bits inside each pair, drop the unused low end
        const uint32_t rev = (c2 >> (2 * j)) & ((1u << (2 * k)) - 1u); // k <= 13
        const uint32_t x = __builtin_bitreverse32(rev);
        const uint32_t fwd = (((x & 0xAAAAAAAAu) >> 1) | ((x & 0x55555555u) << 1)) >> (32u - 2u * k);
        const bool bad = ((badbits >> j) & ((1u << k) - 1u)) != 0;
        const uint32_t i = i0 + j, e = i + W.sig_move_offset;
        cand[j] = i < nops && i <= n - k && e < n; // n >= k here; e >= n: end_raw_idx[e] == -1 (gmove.cpp:892-894)
        slot[j] = (cand[j] && !bad) ? table[rna ? rev : fwd] : -1;
    }
    uint32_t out[E];
#pragma unroll
    for (int j = 0; j < E; ++j) {
        out[j] = PG_INVALID_SLOT;
        if (cand[j]) {
            const uint32_t i = i0 + j;
            auto cp = [&](int32_t Z, uint32_t tix_of_match_Zm1) -> uint32_t { return Z <= 0 ? 0u : ((uint32_t)Z > n ? m : tix_of_match_Zm1); };
            const uint32_t cA = cp((int32_t)i - M + 1, tx_lo[j]), cB = cp((int32_t)i + (int32_t)k + M, tx_hi[j]);
            // accepted (gmove.cpp:916-924). Whether its window can be printed (gmove.cpp:928-944 is undefined for margin > start or an
            // empty window) only matters if the event is KEPT: the emit kernels check it there.
            if (pick_kmer(i, k, M, n, m, rna, st_k, end_k, cA, cB) && slot[j] >= 0 && len[j] <= W.max_dur && len[j] >= W.min_dur) out[j] = (uint32_t)slot[j];
        }
    }
    uint32_t *__restrict__ dst = O.ev_slot + o0; // uniform base, 32-bit lane offset; one 8/12/16-byte store at 4-byte alignment
    if (i0 + E <= nops) {
        struct alignas(4) Vec { uint32_t v[E]; }; // exactly E dwords: one global_store_dwordx2/x3/x4
        Vec v;
#pragma unroll
        for (int j = 0; j < E; ++j) v.v[j] = out[j];
        *reinterpret_cast<Vec *>(dst + i0) = v;
    } else { // the read's last, partial group: the entries behind it belong to the next read
#pragma unroll
        for (int j = 0; j < E; ++j) if (i0 + j < nops) dst[i0 + j] = out[j];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_events: op-parallel. Thread t of workgroup w takes op indices g0 .. g0+3 of each of the workgroup's 4 tiles of 4096.
// ---------------------------------------------------------------------------------------------------------------------
// 16 sequence bytes -> 2-bit codes (byte q at bits 2q) + one "not A C G T/U" bit per byte, four bytes at a time:
// code = ((x >> 1) & 3) ^ (((x >> 1) & 3) >> 1) maps A C G T U to 0 1 2 3 3; the letter a code stands for is rebuilt
// (0x41 + 2 b0 + 6 b1 + 11 b0 b1) and compared with the byte, 'U' (= 'T' ^ 1) passing on RNA-oriented records only.
// NW words of sequence bytes -> 2-bit codes (byte p at bits 2p) + one "not A C G T/U" bit per byte
template <int NW> __device__ __forceinline__ void codes_of(const uint32_t (&w)[NW], bool rna, uint64_t &code2, uint32_t &bad) {
    code2 = 0; bad = 0;
#pragma unroll
    for (int q = 0; q < NW; ++q) {
        const uint32_t x = w[q];
        uint32_t y = (x >> 1) & 0x03030303u;
        y ^= (y >> 1) & 0x01010101u;
        const uint32_t b0 = y & 0x01010101u, b1 = (y >> 1) & 0x01010101u, b01 = b0 & b1;
        const uint32_t expect = 0x41414141u + 2u * b0 + 6u * b1 + 11u * b01;
        uint32_t d = x ^ expect;
        if (rna) d &= ~b01; // 'U' against the 'T' of code 3
        const uint32_t nz = (((d & 0x7f7f7f7fu) + 0x7f7f7f7fu) | d) & 0x80808080u; // 0x80 in every non-zero byte
        code2 |= (uint64_t)((y * 0x01041040u) >> 24) << (8 * q);
        bad |= ((((nz >> 7) * 0x00204081u) >> 21) & 0xfu) << (4 * q);
    }
}

// One op's event with every look-up done by the thread itself and the k bases fetched one by one: k_events' way for a tile that
// more than PG_EV_TBL reads touch (reads of a few ops each). Small, not fast.
__device__ __forceinline__ uint32_t event_slot_scalar(const PgDevBatch &B, const PgWalkParams &W, const PgWalkOut &O, uint64_t g, uint32_t own_len, uint32_t &r) {
    r = owner_search(B, g);
    const uint32_t k = W.k;
    const PgReadMeta *mp = O.meta + r;
    const uint32_t flags = mp->flags, n = mp->nops;
    const uint64_t o0 = mp->o0;
    const bool generic = O.gen_flag[r] == O.batch_id, skip = O.oor && O.oor[r];
    if (!(flags & PG_RM_LIVE) || skip) return PG_INVALID_SLOT;
    if (generic) return W.no_generic ? PG_INVALID_SLOT : O.ev_slot[g]; // computed by k_walk
    if (!(flags & PG_RM_DIRECT_OK)) return PG_INVALID_SLOT;
    if (own_len >= PG_OP_N_LIMIT) report_error(O, r, PGR_ERR_RANGE);
    const bool rna = (flags & PG_RM_RNA) != 0;
    const uint32_t i = (uint32_t)(g - o0), e = i + W.sig_move_offset;
    if (!(i <= n - k && e < n)) return PG_INVALID_SLOT; // gmove.cpp:891-894
    const uint64_t s0 = mp->s0; const uint32_t slen = mp->slen;
    uint32_t code = 0;
    for (uint32_t t = 0; t < k; ++t) { // gmove.cpp:849-853, 883, 899: base t of the k-mer string is match i+t, mirrored on RNA-oriented records
        const uint32_t pidx = i + t;
        const uint8_t bc = base_code(B.seq[rna ? s0 + slen - 1u - pidx : s0 + pidx], rna);
        if (bc > 3) return PG_INVALID_SLOT;
        code = rna ? code | ((uint32_t)bc << (2u * t)) : (code << 2) | bc;
    }
    const int32_t slot = (rna ? W.table_u : W.table_t)[code];
    const uint32_t dur = W.sig_move_offset == 0 ? own_len : B.op_n[g + W.sig_move_offset];
    const bool ok = slot >= 0 && dur <= W.max_dur && dur >= W.min_dur && pick_kmer(i, k, W.pick_margin, n, 0u, rna, mp->st_k, mp->end_k, 0u, 0u);
    return ok ? (uint32_t)slot : PG_INVALID_SLOT;
}

// k_events: a workgroup of 1024 threads takes 4 tiles of 4096 op indices side by side, 256 threads per tile, 16 CONSECUTIVE ops per
// thread; every stage runs once per workgroup:
//   1. the tile's reads go into an LDS table (entry e = read rFirst + e, built by thread e of the tile): first op relative to the
//      tile, class | orientation, sequence offset / length, and the range [ilo, ihi] of event indices that pass the position tests
//      of the reference (gmove.cpp:891-894: a window exists; 204-211 without I/D ops: the two ends of indel_pos) -- worked out
//      once per read instead of once per event;
//   2. every thread finds the read(s) of its 16 ops in the table (a guess from the tile's mean read length, then a probe), fetches
//      their 16 bases as one or two 16-byte windows of the sequence and leaves 2-bit codes (match order) + "not A C G T/U" bits
//      in LDS; its own op_n give the window lengths and, summed along DPP rows of 16 lanes (= 256 ops), cum / btot;
//   3. an event is k consecutive codes starting at its op (its own thread's and the next one's) -> slot table -> duration and
//      position tests. COUNT (direct ranking): the accepted events are counted per (tile, slot) into hist[slot][tile..tile+3].
// A tile touched by more than PG_EV_TBL reads, and a thread whose 16 ops span more than two reads, take event_slot_scalar.
#ifndef PG_EV2_WAVES
#define PG_EV2_WAVES 4 // waves per SIMD the partitioned variant is compiled for (8 = two workgroups per CU, if it fits 64 registers)
#endif
// LT: both slot tables (<= 1024 codes each: k <= 5; <= 1024 slots: COUNT) as 16-bit entries in LDS -- 16 look-ups per thread at LDS latency
// COUNT: 0 = slots only; 1 = direct ranking: counts per (tile, slot), the event's read in the upper bits of its slot word; 2 = partitioned
// ranking (pg_place.hip): counts per (tile, high digit of the slot = slot >> cshift), plain slot words
template <int COUNT, bool LT> __global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(COUNT == 1 ? 8 : (COUNT == 2 ? PG_EV2_WAVES : 4), 8))) void k_events(PgDevBatch B, PgWalkParams W, PgWalkOut O, int nbits, uint32_t n_tiles, uint32_t *__restrict__ hist, uint32_t cshift) {
    constexpr int TBL = PG_EV_TBL;
    __shared__ uint16_t ltab[LT ? 2048 + 2 : 2]; // [2048]: 0xFFFF, where a position that is no candidate looks itself up
    __shared__ uint32_t cnt[COUNT ? 4 : 1][COUNT ? PG_RANK_MAX_DIGITS + 32 : 1]; // + 32 dummy bins: positions that are no event
    __shared__ int32_t t_o0[4][TBL], t_ilo[4][TBL], t_ihi[4][TBL];
    __shared__ uint32_t t_fl[4][TBL], t_s0lo[4][TBL], t_s0hi[4][TBL], t_slen[4][TBL];
    // the thread's 16 slots wait here for the end of the kernel (transposed: op j of thread lt at [j][lt]): a global store between two
    // slot-table look-ups would serialise them (on this part a wait for a load also waits for every store issued in front of it), and
    // 16 more live registers cost a wave per SIMD. 16 bits per slot in direct mode (<= 1024 slots), 32 otherwise.
    typedef typename std::conditional<COUNT == 1, uint16_t, uint32_t>::type stage_t;
    __shared__ stage_t stage[4][16][256];
    constexpr uint32_t STAGE_INVALID = COUNT == 1 ? 0xFFFFu : PG_INVALID_SLOT;
    __shared__ uint32_t sh_rf[5], sh_R[4], sh_over[4];
    const uint32_t tid = threadIdx.x, tile0 = blockIdx.x * 4u, k = W.k, tq = tid >> 8, lt = tid & 255u;
    const int lane = lane_id();
    const uint64_t N = B.n_ops;
    // the op arrays are only read below NL: a caller's n_ops that is too large (reported by k_batch_init) is not followed behind them
    const uint64_t NL = B.n_reads && B.op_off[B.n_reads] < N ? B.op_off[B.n_reads] : N;
    const uint64_t T0 = (uint64_t)(tile0 + tq) * PG_SORT_TILE, g0 = T0 + (uint64_t)lt * 16u;
    const bool tile_live = T0 < N; // wave-uniform
    if (COUNT) for (uint32_t i = tid; i < 4 * (PG_RANK_MAX_DIGITS + 32); i += 1024) (&cnt[0][0])[i] = 0;
    if (tid < 5) { const uint64_t T = (uint64_t)(tile0 + tid) * PG_SORT_TILE; sh_rf[tid] = T < N ? owner_of(B, O, T) : B.n_reads; }
    if (tid < 4) { sh_R[tid] = 0; sh_over[tid] = 0; }
    if (LT && tid == 0) ltab[2048] = (uint16_t)0xFFFFu;
    if (LT) for (uint32_t i = tid; i < 2048; i += 1024) { // [0, 1024): T-spelled codes, [1024, 2048): U-spelled; -1 (not in the slice) -> 0xFFFF
        const uint32_t code = i & 1023u, which = i >> 10;
        ltab[i] = code < W.n_codes ? (uint16_t)W.table_t[which * W.n_codes + code] : (uint16_t)0xFFFFu;
    }
    // the thread's own op_n: four 16-byte loads in flight in front of everything else
    uint32_t opn[16];
    if (g0 + 16 <= NL) {
#pragma unroll
        for (int v = 0; v < 4; ++v) { const uint4 x = *reinterpret_cast<const uint4 *>(B.op_n + g0 + 4 * v); opn[4 * v] = x.x; opn[4 * v + 1] = x.y; opn[4 * v + 2] = x.z; opn[4 * v + 3] = x.w; }
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) opn[j] = g0 + j < NL ? B.op_n[g0 + j] : 0u;
    }
    const uint64_t seq_total = B.seq_off[B.n_reads];
    const uint32_t *__restrict__ seq32 = reinterpret_cast<const uint32_t *>(B.seq); // 4-byte aligned (checked by the host)
    const int64_t seq_last_dw = seq_total ? (int64_t)((seq_total - 1) >> 2) : 0;
    PG_PROBE_BEGIN(1);
    __syncthreads();
    PG_MARK(1, 0); // first reads of the tiles, op_n

    // ---- stage 1: the table entry of read rFirst + lt ------------------------------------------------------------------------------
    const uint32_t rFirst = sh_rf[tq];
    const uint64_t halo_end = T0 + PG_SORT_TILE + 16 < N ? T0 + PG_SORT_TILE + 16 : N;
    if (COUNT && tile_live && lt == 0) O.tile_read[tile0 + tq] = rFirst;
    if (tile_live) {
        const uint32_t r = rFirst + lt;
        if (lt < TBL && r < B.n_reads && (lt == 0 || B.op_off[r] < halo_end)) {
            const PgReadMeta *mp = O.meta + r;
            const uint32_t flags = mp->flags;
            const uint64_t s0 = mp->s0;
            const bool generic = O.gen_flag[r] == O.batch_id, skip = O.oor && O.oor[r], rna = (flags & PG_RM_RNA) != 0;
            // W.no_generic: the caller vouched for a batch of matches only and k_walk was not launched; a listed read then has no
            // events here and fails the batch on the host (pg_api.hip: check_read_errors)
            const uint32_t kind = (!(flags & PG_RM_LIVE) || skip) ? 0u : (generic ? (W.no_generic ? 0u : 2u) : ((flags & PG_RM_DIRECT_OK) ? 1u : 0u));
            // event i of a read of n matches and no I/D op is a candidate iff i <= n-k and i+off < n (gmove.cpp:891-894) and is picked
            // iff left >= M - st_k and left + k <= end_k, left = i (DNA-oriented) or n-i-k (RNA-oriented): pick_kmer with m = 0
            const int64_t n = mp->nops, M = W.pick_margin, stk = mp->st_k, endk = mp->end_k, kk = k, off = W.sig_move_offset;
            int64_t ilo = rna ? n - endk : M - stk, ihi = rna ? n - kk - M + stk : endk - kk;
            if (ilo < 0) ilo = 0;
            if (ihi > n - kk) ihi = n - kk;
            if (ihi > n - off - 1) ihi = n - off - 1;
            if (ihi < -1) ihi = -1;
            if (ilo > 0x7fffffff) ilo = 0x7fffffff;
            t_o0[tq][lt] = (int32_t)((int64_t)mp->o0 - (int64_t)T0); t_ilo[tq][lt] = (int32_t)ilo; t_ihi[tq][lt] = (int32_t)ihi;
            t_fl[tq][lt] = kind | (rna ? 4u : 0u); t_s0lo[tq][lt] = (uint32_t)s0; t_s0hi[tq][lt] = (uint32_t)(s0 >> 32); t_slen[tq][lt] = mp->slen;
            atomicMax(&sh_R[tq], lt + 1u);
            if (lt == TBL - 1 && r + 1 < B.n_reads && B.op_off[r + 1] < halo_end) sh_over[tq] = 1; // more reads than the table holds
        }
    }
    // ---- sums of op_n over 256-op blocks = DPP rows of 16 lanes: cum at 4-op granularity, the block totals ----------------------------
    uint32_t s16 = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) s16 += opn[j];
    uint32_t rowinc = s16;
    rowinc += dpp_zero<0x111, 0xF>(rowinc); rowinc += dpp_zero<0x112, 0xF>(rowinc); rowinc += dpp_zero<0x114, 0xF>(rowinc); rowinc += dpp_zero<0x118, 0xF>(rowinc);
    const uint32_t rowex = rowinc - s16; // sum of op_n over [g0 & ~255, g0)
    if (g0 < N) {
        const uint32_t c1 = rowex + opn[0] + opn[1] + opn[2] + opn[3], c2_ = c1 + opn[4] + opn[5] + opn[6] + opn[7], c3 = c2_ + opn[8] + opn[9] + opn[10] + opn[11];
        *reinterpret_cast<uint4 *>(O.cum + (g0 >> 2)) = make_uint4(rowex, c1, c2_, c3); // entries behind n_ops are padding
        if ((lane & 15) == 15 || g0 + 16 >= N) O.btot[g0 >> 8] = rowinc;
    }
    PG_MARK(1, 1); // table entry, block sums
    __syncthreads();
    PG_MARK(1, 2); // barrier
    const uint32_t R = sh_R[tq];
    const bool over = sh_over[tq] != 0; // wave-uniform

    // ---- stage 2: the reads of the thread's 16 ops, their bases ---------------------------------------------------------------------
    // A = the read of op x0, B = the read (with ops) that starts inside the group at op jb (16 = none), more = a third read starts inside
    struct Seg { uint32_t e; int32_t o0; uint32_t fl; };
    auto find_reads = [&](uint32_t x0, Seg &A, Seg &Bs, uint32_t &jb, bool &more) {
        uint32_t e = R > 1 ? (uint32_t)(((uint64_t)x0 * R) >> 12) : 0u; // reads of about equal length: a guess, then a probe
        if (e >= R) e = R - 1;
        while (e > 0 && t_o0[tq][e] > (int32_t)x0) --e;
        while (e + 1 < R && t_o0[tq][e + 1] <= (int32_t)x0) ++e;
        A.e = e; A.o0 = t_o0[tq][e]; A.fl = t_fl[tq][e];
        jb = 16; more = false; Bs = A;
        uint32_t f = e + 1;
        if (f < R && t_o0[tq][f] < (int32_t)(x0 + 16)) {
            while (f + 1 < R && t_o0[tq][f + 1] == t_o0[tq][f]) ++f; // reads without ops
            Bs.e = f; Bs.o0 = t_o0[tq][f]; Bs.fl = t_fl[tq][f];
            jb = (uint32_t)(Bs.o0 - (int32_t)x0);
            more = f + 1 < R && t_o0[tq][f + 1] < (int32_t)(x0 + 16);
        }
    };
    // NB = 4 NW bases of read entry e starting at its event index i, as 2-bit codes (base i+p at bits 2p) + bad bits; 0 / all bad if no
    // window. NB >= 16 + k - 1: 20 with both tables in LDS (k <= 5), 28 otherwise (k <= 13)
    constexpr int NW = LT ? 5 : 7, NB = 4 * NW;
    auto window = [&](const Seg &sg, uint32_t i, uint64_t &code, uint32_t &bad) {
        code = 0; bad = (1u << NB) - 1u;
        if ((sg.fl & 3u) != 1u) return;
        const bool rna = (sg.fl >> 2) & 1u;
        const uint64_t s0 = (uint64_t)t_s0lo[tq][sg.e] | ((uint64_t)t_s0hi[tq][sg.e] << 32);
        // DNA-oriented: base i+p is sequence byte s0+i+p; RNA-oriented: match p is byte s0+slen-1-p, so the window is the NB bytes
        // that END at s0+slen-1-i. Bytes outside the read (or the buffer: clamped) belong to non-candidates.
        const int64_t a = rna ? (int64_t)(s0 + t_slen[tq][sg.e]) - NB - (int64_t)i : (int64_t)(s0 + i);
        const int64_t adw = a >> 2;
        uint32_t d[NW + 1];
        if (adw >= 0 && adw + NW <= seq_last_dw) {
            // the window's NW + 1 dwords as 16- and 8-byte loads (4-byte aligned): a load costs the texture addresser per LANE, whatever
            // its width (profiles/r04_gather_bound.txt 5) -- two lane loads instead of six or eight
            typedef uint32_t pg_a4 __attribute__((ext_vector_type(4), aligned(4)));
            typedef uint32_t pg_a2 __attribute__((ext_vector_type(2), aligned(4)));
            const pg_a4 v0 = *reinterpret_cast<const pg_a4 *>(seq32 + adw);
            d[0] = v0.x; d[1] = v0.y; d[2] = v0.z; d[3] = v0.w;
            if (NW + 1 == 6) { const pg_a2 v1 = *reinterpret_cast<const pg_a2 *>(seq32 + adw + 4); d[4] = v1.x; d[5] = v1.y; }
            else { const pg_a4 v1 = *reinterpret_cast<const pg_a4 *>(seq32 + adw + 4); d[4] = v1.x; d[5] = v1.y; d[NW + 1 > 6 ? 6 : 0] = v1.z; d[NW + 1 > 7 ? 7 : 0] = v1.w; }
        } else { // at the ends of the sequence buffer: dword by dword, clamped
#pragma unroll
            for (int q = 0; q < NW + 1; ++q) { int64_t x = adw + q; x = x < 0 ? 0 : (x > seq_last_dw ? seq_last_dw : x); d[q] = seq32[x]; }
        }
        const uint32_t sh = (uint32_t)(a & 3) * 8u;
        uint32_t w[NW];
#pragma unroll
        for (int q = 0; q < NW; ++q) w[q] = sh ? (d[q] >> sh) | (d[q + 1] << (32u - sh)) : d[q];
        if (rna) { // byte q of the window is match i+NB-1-q: into match order
            uint32_t t[NW];
#pragma unroll
            for (int q = 0; q < NW; ++q) t[q] = __builtin_bswap32(w[NW - 1 - q]);
#pragma unroll
            for (int q = 0; q < NW; ++q) w[q] = t[q];
        }
        codes_of<NW>(w, rna, code, bad);
    };
    // every thread fetches the bases of its own 16 ops AND of the k-1 ops behind them (the events that start in the group end
    // there): no exchange between threads, no halo, no barrier between this stage and the events
    Seg A{0, 0, 0}, Bs{0, 0, 0}; uint32_t jb = 16; bool more = false;
    uint64_t c64 = 0; uint32_t bad32 = (1u << NB) - 1u; // bases of ops x0 .. x0+NB-1, op x0+p at bits 2p
    if (tile_live && !over && g0 < N) {
        const uint32_t x0 = lt * 16u;
        find_reads(x0, A, Bs, jb, more);
        uint64_t cA, cB = 0; uint32_t bA, bB = (1u << NB) - 1u;
        window(A, (uint32_t)((int32_t)x0 - A.o0), cA, bA);
        if (jb < 16) window(Bs, 0u, cB, bB);
        c64 = jb < 16 ? (cA & ((1ull << (2u * jb)) - 1ull)) | (cB << (2u * jb)) : cA;
        bad32 = jb < 16 ? (bA & ((1u << jb) - 1u)) | (bB << jb) : bA;
    }
    PG_MARK(1, 3); // reads of the group, base codes

    // ---- stage 3: the events ------------------------------------------------------------------------------------------------------------
    const uint32_t cmask = (1u << nbits) - 1u;
    auto cdig = [&](uint32_t sl) { return (COUNT == 2 ? sl >> cshift : sl) & cmask; }; // the digit an accepted event is counted under
    if (tile_live && g0 < N && (over || more)) { // every op on its own (rare): its read, its k bases, its window length
#pragma unroll 1
        for (uint32_t j = 0; j < 16 && g0 + j < N; ++j) {
            uint32_t rd;
            const uint32_t sl = event_slot_scalar(B, W, O, g0 + j, g0 + j < NL ? B.op_n[g0 + j] : 0u, rd);
            constexpr uint32_t UNK = COUNT == 2 ? PG_PART_REL_UNKNOWN : PG_REL_UNKNOWN, RSH = COUNT == 2 ? PG_PART_REL_SHIFT : PG_SLOT_BITS;
            const uint32_t rel = rd - rFirst < UNK ? rd - rFirst : UNK;
            O.ev_slot[g0 + j] = COUNT && sl != PG_INVALID_SLOT ? sl | (rel << RSH) : sl; // the read rides in the upper bits (PgWalkOut::tile_read)
            if (COUNT && sl != PG_INVALID_SLOT) atomicAdd(&cnt[tq][cdig(sl)], 1u);
        }
    } else if (COUNT == 1 && LT && tile_live && g0 < N && W.sig_move_offset == 0 &&
               !((A.fl & 3u) == 1u && jb < 16 && (Bs.fl & 3u) == 1u && ((A.fl ^ Bs.fl) & 4u))) {
        // ---- the straight-line form (direct ranking, both tables in LDS, window of an event = its own op; the two reads of the group,
        // if both direct, of one orientation): the position, base and duration tests of the 16 events become three 16-bit masks, every
        // event then costs one shift of the thread's code word, one LDS look-up (a position that fails a test looks up the 0xFFFF entry),
        // one LDS count; the 16 slots stay in eight registers until the stores. No branch and no wait per event.
        const uint32_t x0 = lt * 16u;
        const uint32_t kindA = A.fl & 3u, kindB = jb < 16 ? (Bs.fl & 3u) : 0u;
        auto bits = [](int32_t lo, int32_t hi) -> uint32_t { // bits lo .. hi (0 <= lo <= 16, -1 <= hi <= 15), none if lo > hi
            const uint32_t h = (uint32_t)(hi < 0 ? 0 : hi), l = (uint32_t)(lo > 15 ? 15 : lo);
            return lo <= hi ? ((2u << h) - 1u) & ~((1u << l) - 1u) : 0u;
        };
        // candidates by position (gmove.cpp:891-894, 204-211 through the per-read range [ilo, ihi]); gm: ops of generic reads (k_walk's events)
        uint32_t pos = 0, gm = 0;
        {
            const int32_t jbi = (int32_t)jb, iA0 = (int32_t)x0 - A.o0; // event index of op x0 in read A
            if (kindA == 1u) {
                int32_t lo = t_ilo[tq][A.e] - iA0, hi = t_ihi[tq][A.e] - iA0; // ilo >= 0, ihi >= -1, iA0 >= 0: no overflow
                lo = lo < 0 ? 0 : (lo > 16 ? 16 : lo); hi = hi > jbi - 1 ? jbi - 1 : (hi < -1 ? -1 : hi);
                pos = bits(lo, hi);
            } else if (kindA == 2u) gm = bits(0, jbi - 1);
            if (kindB == 1u) { // event index of op x0 + j in read B: j - jb
                int32_t lo = t_ilo[tq][Bs.e], hi = t_ihi[tq][Bs.e];
                lo = jbi + (lo > 16 ? 16 : lo); hi = hi > 15 ? 15 : jbi + hi; hi = hi > 15 ? 15 : hi;
                pos |= bits(lo > 16 ? 16 : lo, hi);
            } else if (kindB == 2u) gm |= bits(jbi, 15);
        }
        const uint32_t nvalid = g0 + 16 <= N ? 16u : (uint32_t)(N - g0);
        const uint32_t vmask = (2u << (nvalid - 1u)) - 1u;
        // a base that is not A C G T/U anywhere in the k-mer: k <= 5 here (both tables fit 1024 codes)
        const uint32_t km1 = k - 1u;
        const uint32_t bm = bad32 | (bad32 >> (km1 < 1u ? km1 : 1u)) | (bad32 >> (km1 < 2u ? km1 : 2u)) | (bad32 >> (km1 < 3u ? km1 : 3u)) | (bad32 >> (km1 < 4u ? km1 : 4u));
        // durations (gmove.cpp:916-921): the thread's own op_n again (they were summed long ago; registers are worth more than cache hits)
        uint32_t dm = 0, orv = 0;
        {
            const uint32_t range = W.max_dur - W.min_dur;
#pragma unroll
            for (int v = 3; v >= 0; --v) {
                uint32_t len[4] = {0, 0, 0, 0};
                if (g0 + 4 * v + 4 <= NL) { const uint4 x = *reinterpret_cast<const uint4 *>(B.op_n + g0 + 4 * v); len[0] = x.x; len[1] = x.y; len[2] = x.z; len[3] = x.w; }
                else { for (int u = 0; u < 4; ++u) if (g0 + 4 * v + u < NL) len[u] = B.op_n[g0 + 4 * v + u]; }
#pragma unroll
                for (int u = 3; u >= 0; --u) { dm = (dm << 1) | ((len[u] - W.min_dur <= range) ? 1u : 0u); orv |= len[u]; }
            }
            if (W.max_dur < W.min_dur) dm = 0;
        }
        const uint32_t cm = pos & ~bm & dm & vmask;
        PG_MARK(1, 5); // straight-line form: the three masks (op_n arrives)
        // the code word in look-up order, times two (byte offsets into ltab): RNA-oriented records look the mirrored k-mer up (first base
        // lowest: a right shift by 2j of the word as it is, gmove.cpp:883, 899); DNA-oriented ones the k-mer itself (first base highest:
        // the word with its 2-bit groups in reverse order, base p at bits 62-2p, shifted right by 64-2k-2j)
        const bool rna = ((kindA == 1u ? A.fl : Bs.fl) >> 2) & 1u;
        uint64_t S64; uint32_t sh, tb2; int32_t step;
        if (rna) { S64 = c64 << 1; sh = 0u; step = 2; tb2 = 2048u; } // op 31 loses a bit: events only reach op 15 + k - 1
        else {
            const uint32_t lo = __builtin_bitreverse32((uint32_t)(c64 >> 32)), hi = __builtin_bitreverse32((uint32_t)c64);
            const uint64_t r = (uint64_t)lo | ((uint64_t)hi << 32);
            S64 = ((r & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((r & 0x5555555555555555ull) << 1);
            sh = 63u - 2u * k; step = -2; tb2 = 0u;
        }
        const uint32_t mask2 = ((1u << (2u * k)) - 1u) << 1;
        const char *lbase = reinterpret_cast<const char *>(ltab);
        uint32_t *cbase = &cnt[tq][0];
        const uint32_t dummy = PG_RANK_MAX_DIGITS + ((uint32_t)lane & 31u);
        uint32_t pk[8];
#pragma unroll
        for (int v = 0; v < 4; ++v) { // four look-ups in flight, then their four counts
            uint32_t t16[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = 4 * v + u;
                const uint32_t off = ((uint32_t)(S64 >> (sh + (uint32_t)(step * j))) & mask2) | tb2;
                const uint32_t m = (uint32_t)((int32_t)(cm << (31 - j)) >> 31);
                t16[u] = *reinterpret_cast<const volatile uint16_t *>(lbase + ((off & m) | (4096u & ~m)));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) atomicAdd(cbase + (t16[u] < dummy ? t16[u] : dummy), 1u);
            pk[2 * v] = t16[0] | (t16[1] << 16); pk[2 * v + 1] = t16[2] | (t16[3] << 16);
        }
        PG_MARK(1, 6); // straight-line form: look-ups and counts
        if (gm & vmask) { // the events of generic reads come from k_walk (plain slots < 1024, or PG_INVALID_SLOT)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                uint32_t wk[4] = {PG_INVALID_SLOT, PG_INVALID_SLOT, PG_INVALID_SLOT, PG_INVALID_SLOT};
                if (g0 + 4 * v + 4 <= N) { const uint4 x = *reinterpret_cast<const uint4 *>(O.ev_slot + g0 + 4 * v); wk[0] = x.x; wk[1] = x.y; wk[2] = x.z; wk[3] = x.w; }
                else { for (int u = 0; u < 4; ++u) if (g0 + 4 * v + u < N) wk[u] = O.ev_slot[g0 + 4 * v + u]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = 4 * v + u;
                    if ((gm & vmask) >> j & 1u) {
                        const uint32_t x = wk[u] & 0xFFFFu; // PG_INVALID_SLOT -> 0xFFFF
                        pk[j >> 1] = (j & 1) ? (pk[j >> 1] & 0xFFFFu) | (x << 16) : (pk[j >> 1] & 0xFFFF0000u) | x;
                        if (x != 0xFFFFu) atomicAdd(cbase + x, 1u);
                    }
                }
            }
        }
        if (orv >= PG_OP_N_LIMIT) { // an op of 2^24 samples or more in a direct read: any fails the batch, the lowest read is reported
            for (uint32_t j = 0; j < 16 && g0 + j < N; ++j) {
                const bool inB = j >= jb;
                if (((inB ? Bs.fl : A.fl) & 3u) == 1u && B.op_n[g0 + j] >= PG_OP_N_LIMIT) report_error(O, rFirst + (inB ? Bs.e : A.e), PGR_ERR_RANGE);
            }
        }
        // the stores: slot | read << PG_SLOT_BITS (read = its table entry); 0xFFFF sign-extends to PG_INVALID_SLOT and stays it under the OR
        const uint32_t relA = A.e << PG_SLOT_BITS, relB = Bs.e << PG_SLOT_BITS;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            uint32_t o4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = 4 * v + u;
                const int32_t x = (j & 1) ? (int32_t)pk[j >> 1] >> 16 : (int32_t)(int16_t)(pk[j >> 1] & 0xFFFFu);
                o4[u] = (uint32_t)x | ((uint32_t)j >= jb ? relB : relA);
            }
            if (g0 + 4 * v + 4 <= N) *reinterpret_cast<uint4 *>(O.ev_slot + g0 + 4 * v) = make_uint4(o4[0], o4[1], o4[2], o4[3]);
            else { for (int u = 0; u < 4; ++u) if (g0 + 4 * v + u < N) O.ev_slot[g0 + 4 * v + u] = o4[u]; }
        }
    } else if (COUNT == 2 && tile_live && g0 < N) {
        // ---- partitioned ranking (k = 9: global slot tables, 4 waves per SIMD, registers to spare): all 16 table look-ups of the thread are
        // issued before the first one is used (the per-event form below waits for every look-up where it stands: 16 dependent L2 round
        // trips per thread), the thread's own op_n stay in their registers, the slots go from registers to their four 16-byte stores
        const uint32_t x0 = lt * 16u;
        const int32_t iloA = t_ilo[tq][A.e], ihiA = t_ihi[tq][A.e];
        int32_t iloB = 0, ihiB = -1;
        if (jb < 16) { iloB = t_ilo[tq][Bs.e]; ihiB = t_ihi[tq][Bs.e]; }
        const bool any_generic = (A.fl & 3u) == 2u || (jb < 16 && (Bs.fl & 3u) == 2u);
        uint32_t tix[16], sl[16];
        bool too_long = false;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            uint32_t wk[4] = {PG_INVALID_SLOT, PG_INVALID_SLOT, PG_INVALID_SLOT, PG_INVALID_SLOT};
            if (any_generic) { // events of generic reads come from k_walk
                if (g0 + 4 * v + 4 <= N) { const uint4 x = *reinterpret_cast<const uint4 *>(O.ev_slot + g0 + 4 * v); wk[0] = x.x; wk[1] = x.y; wk[2] = x.z; wk[3] = x.w; }
                else { for (int u = 0; u < 4; ++u) if (g0 + 4 * v + u < N) wk[u] = O.ev_slot[g0 + 4 * v + u]; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = 4 * v + u;
                const bool inB = (uint32_t)j >= jb;
                const int32_t o0r = inB ? Bs.o0 : A.o0, ilo = inB ? iloB : iloA, ihi = inB ? ihiB : ihiA;
                const uint32_t fl = inB ? Bs.fl : A.fl, kind = fl & 3u;
                const int32_t i = (int32_t)(x0 + j) - o0r;
                tix[j] = 0xFFFFFFFFu; sl[j] = kind == 2u ? wk[u] : PG_INVALID_SLOT;
                if (kind == 1u && i >= ilo && i <= ihi && g0 + j < N) {
                    const uint32_t field = (uint32_t)(c64 >> (2 * j)) & ((1u << (2u * k)) - 1u), badf = (uint32_t)(((uint64_t)bad32 >> j) & ((1u << k) - 1u));
                    const uint32_t xr = __builtin_bitreverse32(field);
                    const uint32_t fwd = (((xr & 0xAAAAAAAAu) >> 1) | ((xr & 0x55555555u) << 1)) >> (32u - 2u * k);
                    const bool rna = (fl >> 2) & 1u;
                    const uint32_t dur = W.sig_move_offset == 0 ? opn[j] : B.op_n[g0 + j + W.sig_move_offset]; // gmove.cpp:916-921
                    if (!badf && dur <= W.max_dur && dur >= W.min_dur) tix[j] = rna ? field + W.n_codes : fwd; // table_u sits behind table_t
                }
                if (kind == 1u && opn[j] >= PG_OP_N_LIMIT && g0 + j < N) too_long = true;
            }
        }
        PG_MARK(1, 5); // partitioned variant: position / base / duration tests, table indices
        // a table that is code + constant inside a range (PgWalkParams::aff_ok) is computed; only the other indices are looked up
#pragma unroll
        for (int j = 0; j < 16; ++j) if (tix[j] != 0xFFFFFFFFu) {
            const uint32_t x = tix[j] >= W.n_codes ? 1u : 0u;
            if (W.aff_ok[x]) {
                const uint32_t code = tix[j] - (x ? W.n_codes : 0u);
                sl[j] = (code >= W.aff_lo[x] && code <= W.aff_hi[x]) ? (uint32_t)((int32_t)code + W.aff_delta[x]) : PG_INVALID_SLOT;
                tix[j] = 0xFFFFFFFFu;
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) if (tix[j] != 0xFFFFFFFFu) sl[j] = (uint32_t)W.table_t[tix[j]]; // -1 = not in the slice = PG_INVALID_SLOT
        PG_MARK(1, 6); // partitioned variant: the 16 slot look-ups
        if (too_long) {
            for (uint32_t j = 0; j < 16 && g0 + j < N; ++j) {
                const bool inB = j >= jb;
                if (((inB ? Bs.fl : A.fl) & 3u) == 1u && B.op_n[g0 + j] >= PG_OP_N_LIMIT) report_error(O, rFirst + (inB ? Bs.e : A.e), PGR_ERR_RANGE);
            }
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            uint32_t o4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = 4 * v + u;
                const uint32_t x = sl[j];
                if (x != PG_INVALID_SLOT && g0 + j < N) atomicAdd(&cnt[tq][cdig(x)], 1u);
                const uint32_t rel = (uint32_t)j >= jb ? Bs.e : A.e; // the event's read rides in the upper bits (PgWalkOut::tile_read)
                o4[u] = x == PG_INVALID_SLOT ? PG_INVALID_SLOT : x | (rel << PG_PART_REL_SHIFT);
            }
            if (g0 + 4 * v + 4 <= N) *reinterpret_cast<uint4 *>(O.ev_slot + g0 + 4 * v) = make_uint4(o4[0], o4[1], o4[2], o4[3]);
            else { for (int u = 0; u < 4; ++u) if (g0 + 4 * v + u < N) O.ev_slot[g0 + 4 * v + u] = o4[u]; }
        }
    } else if (COUNT != 2 && tile_live && g0 < N) { // (the partitioned variant has taken its own branch above: not compiled into it)
        const uint32_t x0 = lt * 16u;
        const int32_t iloA = t_ilo[tq][A.e], ihiA = t_ihi[tq][A.e];
        int32_t iloB = 0, ihiB = -1;
        if (jb < 16) { iloB = t_ilo[tq][Bs.e]; ihiB = t_ihi[tq][Bs.e]; }
        const bool any_generic = (A.fl & 3u) == 2u || (jb < 16 && (Bs.fl & 3u) == 2u);
        bool too_long = false;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            uint32_t wk[4] = {PG_INVALID_SLOT, PG_INVALID_SLOT, PG_INVALID_SLOT, PG_INVALID_SLOT};
            if (any_generic) { // events of generic reads come from k_walk
                if (g0 + 4 * v + 4 <= N) { const uint4 x = *reinterpret_cast<const uint4 *>(O.ev_slot + g0 + 4 * v); wk[0] = x.x; wk[1] = x.y; wk[2] = x.z; wk[3] = x.w; }
                else { for (int u = 0; u < 4; ++u) if (g0 + 4 * v + u < N) wk[u] = O.ev_slot[g0 + 4 * v + u]; }
            }
            // the window lengths: the thread's own op_n again (they were summed long ago; registers are worth more than cache hits)
            uint32_t len[4] = {0, 0, 0, 0};
            if (g0 + 4 * v + 4 <= NL) { const uint4 x = *reinterpret_cast<const uint4 *>(B.op_n + g0 + 4 * v); len[0] = x.x; len[1] = x.y; len[2] = x.z; len[3] = x.w; }
            else { for (int u = 0; u < 4; ++u) if (g0 + 4 * v + u < NL) len[u] = B.op_n[g0 + 4 * v + u]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = 4 * v + u;
                const bool inB = (uint32_t)j >= jb;
                const int32_t o0r = inB ? Bs.o0 : A.o0, ilo = inB ? iloB : iloA, ihi = inB ? ihiB : ihiA;
                const uint32_t fl = inB ? Bs.fl : A.fl, kind = fl & 3u;
                const int32_t i = (int32_t)(x0 + j) - o0r;
                uint32_t sl = kind == 2u ? wk[u] : PG_INVALID_SLOT;
                if (kind == 1u && i >= ilo && i <= ihi && g0 + j < N) {
                    // matches i .. i+k-1 at bits 2j ..: k 2-bit groups with the FIRST base lowest -- the code the reference looks up on
                    // RNA-oriented records (mirrored k-mer, gmove.cpp:883, 899); its groups reversed on DNA-oriented ones
                    const uint32_t field = (uint32_t)(c64 >> (2 * j)) & ((1u << (2u * k)) - 1u), badf = (uint32_t)(((uint64_t)bad32 >> j) & ((1u << k) - 1u));
                    const uint32_t xr = __builtin_bitreverse32(field);
                    const uint32_t fwd = (((xr & 0xAAAAAAAAu) >> 1) | ((xr & 0x55555555u) << 1)) >> (32u - 2u * k);
                    const bool rna = (fl >> 2) & 1u;
                    // the window length of event i is that of match i + sig_move_offset (op index g + offset, inside the read): gmove.cpp:916-921
                    const uint32_t dur = W.sig_move_offset == 0 ? len[u] : B.op_n[g0 + j + W.sig_move_offset];
                    if (!badf && dur <= W.max_dur && dur >= W.min_dur) {
                        if (LT) { const uint32_t t16 = ltab[rna ? 1024u + field : fwd]; sl = t16 == 0xFFFFu ? PG_INVALID_SLOT : t16; }
                        else if (W.aff_ok[rna ? 1 : 0]) { // the table in use is code + constant inside a range: computed (see the partitioned variant)
                            const uint32_t x = rna ? 1u : 0u, code = rna ? field : fwd;
                            sl = (code >= W.aff_lo[x] && code <= W.aff_hi[x]) ? (uint32_t)((int32_t)code + W.aff_delta[x]) : PG_INVALID_SLOT;
                        }
                        else sl = (uint32_t)W.table_t[rna ? field + W.n_codes : fwd]; // table_u sits behind table_t; -1 = not in the slice = PG_INVALID_SLOT
                    }
                }
                if (kind == 1u && len[u] >= PG_OP_N_LIMIT && g0 + j < N) too_long = true;
                stage[tq][j][lt] = (stage_t)(sl == PG_INVALID_SLOT ? STAGE_INVALID : sl);
                if (COUNT && sl != PG_INVALID_SLOT && g0 + j < N) atomicAdd(&cnt[tq][cdig(sl)], 1u);
            }
        }
        // ---- everything that stores to global memory: behind the last look-up -------------------------------------------------------
        if (too_long) { // an op of 2^24 samples or more in a direct read (PG_OP_N_LIMIT): any fails the batch, the lowest read is reported
            for (uint32_t j = 0; j < 16 && g0 + j < N; ++j) {
                const bool inB = j >= jb;
                if (((inB ? Bs.fl : A.fl) & 3u) == 1u && B.op_n[g0 + j] >= PG_OP_N_LIMIT) report_error(O, rFirst + (inB ? Bs.e : A.e), PGR_ERR_RANGE);
            }
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            uint32_t o4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t x = stage[tq][4 * v + u][lt];
                // COUNT: the event's read rides in the upper bits, as its table entry = read - first read of the tile (PgWalkOut::tile_read)
                const uint32_t rel = (uint32_t)(4 * v + u) >= jb ? Bs.e : A.e;
                o4[u] = x == STAGE_INVALID ? PG_INVALID_SLOT : (COUNT ? x | (rel << (COUNT == 2 ? PG_PART_REL_SHIFT : PG_SLOT_BITS)) : x);
            }
            if (g0 + 4 * v + 4 <= N) *reinterpret_cast<uint4 *>(O.ev_slot + g0 + 4 * v) = make_uint4(o4[0], o4[1], o4[2], o4[3]);
            else { for (int u = 0; u < 4; ++u) if (g0 + 4 * v + u < N) O.ev_slot[g0 + 4 * v + u] = o4[u]; }
        }
    }
    PG_MARK(1, 4); // stores (straight-line form), or the whole of the events (the other forms)
    if (COUNT) {
        __syncthreads();
        for (uint32_t d = tid; d < (1u << nbits); d += 1024)
            *reinterpret_cast<uint4 *>(hist + (uint64_t)d * n_tiles + tile0) = make_uint4(cnt[0][d], cnt[1][d], cnt[2][d], cnt[3][d]);
    }
    PG_PROBE_END(1, blockIdx.x * 16u + (tid >> 6));
}

// SAM/BAM front-end (gmove.cpp:1149-1160): a read with an out-of-range sample is a skipped read (no events, no ':')
__global__ __launch_bounds__(256) void k_apply_oor(uint32_t n_reads, PgWalkOut O) {
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r < n_reads && O.oor[r] && O.status[r] == PGR_OK) O.status[r] = PGR_SKIPPED;
}

// =====================================================================================================
// Stable ranking of accepted events by slot. One tile = 4 waves x ROWS rows x 64 keys; digits of up to
// 10 bits (1024 LDS counters per wave). Two uses:
//   direct  (n_slots <= 1024): digit == slot, so tile-prefix + in-tile rank IS the event's rank inside its
//           k-mer; k_rank_emit applies the sample_limit cut and writes only the kept events;
//   generic (more slots): LSD radix sort by slot in ceil(bits/10) passes, ranks from the sorted order.
// =====================================================================================================

__global__ __launch_bounds__(256) void k_rank_count(const uint32_t *__restrict__ keys, uint32_t n_scalar,
                                                    const uint32_t *__restrict__ n_ptr, uint32_t shift, int nbits,
                                                    uint32_t n_tiles, uint32_t *__restrict__ hist, uint32_t *__restrict__ wcnt) {
    __shared__ uint32_t cnt[4][PG_RANK_MAX_DIGITS];
    const uint32_t tid = threadIdx.x, tile = blockIdx.x, w = tid >> 6;
    const int lane = lane_id();
    const uint32_t ndig = 1u << nbits;
    for (uint32_t ww = 0; ww < 4; ++ww) for (uint32_t d = tid; d < ndig; d += 256) cnt[ww][d] = 0;
    __syncthreads();
    const uint32_t n = n_ptr ? *n_ptr : n_scalar;
    const uint32_t mask = ndig - 1u;
    const uint64_t base = (uint64_t)tile * PG_SORT_TILE + (uint64_t)w * PG_SORT_ROWS * WAVE;
    uint32_t kv[PG_SORT_ROWS]; // all rows in flight before the (serial) ranking
#pragma unroll
    for (int row = 0; row < PG_SORT_ROWS; ++row) {
        const uint64_t idx = base + (uint64_t)row * WAVE + lane;
        kv[row] = idx < n ? keys[idx] : PG_INVALID_SLOT;
    }
    // counts are order-free, so plain LDS atomics do (the stable in-row rank is only needed when placing)
#pragma unroll
    for (int row = 0; row < PG_SORT_ROWS; ++row) {
        const uint32_t key = kv[row];
        if (key != PG_INVALID_SLOT) atomicAdd(&cnt[w][(key >> shift) & mask], 1u);
    }
    (void)lane;
    __syncthreads();
    for (uint32_t d = tid; d < ndig; d += 256) {
        uint32_t sum = 0;
        for (uint32_t ww = 0; ww < 4; ++ww) {
            const uint32_t c = cnt[ww][d];
            wcnt[((uint64_t)tile * 4 + ww) * ndig + d] = c;
            sum += c;
        }
        hist[(uint64_t)d * n_tiles + tile] = sum;
    }
}

// exclusive prefix over tiles, one wave per digit; totals[d] = number of keys with that digit. Direct mode (slot == digit): the
// per-(tile, slot) counts come from k_events<true>; k_rank_emit recounts per wave for the few tiles it really places events from.
// running / limit / tile_last (direct mode, may be null): with base = the context's own running counts -- what pg_submit
// collects with -- the last tile that still places an event of this slot falls out of the same pass: keep = min(cnt,
// room), room = limit - base; it is the largest tile whose exclusive prefix is < room when cnt >= room, else the last
// non-empty tile (k_tile_max finds the same by searching the finished prefix rows when the base arrives later).
// plan (plan.keep != null; direct mode, base = the context's own running counts: pg_submit): the workgroup that finishes LAST also applies
// the sample_limit cut (k_slot_plan's work: gmove.cpp:925-927, 945-950) -- one launch less per batch. Hand-over without fences (a
// device-scope release writes the XCD's whole L2 back: 7 -> 34 us when every workgroup did one): each workgroup's two results leave
// as sc1 stores (agent-scope relaxed atomics), the wave waits for them (vmcnt(0)) and then adds to ONE counter; the workgroup whose
// add returns the last ticket reads all results back with sc1 loads (MI355X_MICROARCH.md, cross-workgroup hand-offs, first row).
struct PgScanPlan {
    uint64_t *keep, *ev_off, *plan_totals, *running_out;
    uint32_t *ticket;
};
#define PG_SCAN_WAVES 16 // digits (= waves) per workgroup of k_rank_scan: 64 workgroups for 1024 slots, 64 adds to the ticket
__global__ __launch_bounds__(PG_SCAN_WAVES * WAVE) void k_rank_scan(uint32_t *__restrict__ hist, uint32_t n_tiles, uint32_t *__restrict__ totals,
                                                  uint64_t *__restrict__ acc_cnt, uint32_t n_slots, uint32_t n_digits, const uint64_t *running,
                                                  uint32_t limit, int32_t *__restrict__ tile_last, uint64_t *__restrict__ acc_copy, PgScanPlan plan,
                                                  const uint32_t *__restrict__ bp_btot, uint32_t bp_nb, uint32_t *__restrict__ bp_out, uint64_t *__restrict__ zero64) {
    const uint32_t d = blockIdx.x * PG_SCAN_WAVES + (threadIdx.x >> 6); // one wave per digit
    const int lane = lane_id();
    if (bp_out && blockIdx.x == gridDim.x - 1) { // partitioned ranking: one extra workgroup turns k_events' 256-op block sums into their
        // exclusive prefix (pg_place.hip: op_prefix), next to the digit scans instead of in a launch of its own. 64 consecutive sums per
        // thread, all 16 loads in flight, one workgroup-wide scan per 65 536 sums (k = 9, 50 000 reads: one trip)
        __shared__ uint32_t bsum[PG_SCAN_WAVES];
        uint32_t carry = 0;
        for (uint32_t c = 0; c < bp_nb; c += PG_SCAN_WAVES * WAVE * 64) {
            const uint32_t i0 = c + threadIdx.x * 64;
            uint4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const uint32_t i = i0 + 4 * u;
                if (i + 4 <= bp_nb) v[u] = *reinterpret_cast<const uint4 *>(bp_btot + i);
                else v[u] = make_uint4(i < bp_nb ? bp_btot[i] : 0u, i + 1 < bp_nb ? bp_btot[i + 1] : 0u, i + 2 < bp_nb ? bp_btot[i + 2] : 0u, 0u);
            }
            uint32_t sum = 0;
#pragma unroll
            for (int u = 0; u < 16; ++u) { const uint4 x = v[u]; v[u] = make_uint4(sum, sum + x.x, sum + x.x + x.y, sum + x.x + x.y + x.z); sum += x.x + x.y + x.z + x.w; }
            const uint32_t inc = wave_incl_scan_u32(sum);
            if (lane == WAVE - 1) bsum[threadIdx.x >> 6] = inc;
            __syncthreads();
            uint32_t off = carry + inc - sum, tot = 0;
            for (uint32_t ww = 0; ww < PG_SCAN_WAVES; ++ww) { if (ww < (threadIdx.x >> 6)) off += bsum[ww]; tot += bsum[ww]; }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const uint32_t i = i0 + 4 * u;
                const uint4 x = make_uint4(v[u].x + off, v[u].y + off, v[u].z + off, v[u].w + off);
                if (i + 4 <= bp_nb) *reinterpret_cast<uint4 *>(bp_out + i) = x;
                else { if (i < bp_nb) bp_out[i] = x.x; if (i + 1 < bp_nb) bp_out[i + 1] = x.y; if (i + 2 < bp_nb) bp_out[i + 2] = x.z; }
            }
            carry += tot;
            __syncthreads();
        }
        if (threadIdx.x == 0) bp_out[bp_nb] = carry;
        return;
    }
    // the chunk sums of the kept window lengths (k_region_place / k_len_partials add to them): zeroed here instead of by a fill launch, a
    // slice per workgroup (the extra workgroup above starts last and is the launch's long pole already)
    if (zero64) for (uint32_t i = blockIdx.x * (PG_SCAN_WAVES * WAVE) + threadIdx.x; i < PG_CHUNK_PART_N; i += (gridDim.x - (bp_out ? 1u : 0u)) * (PG_SCAN_WAVES * WAVE)) zero64[i] = 0;
    if (d < n_digits) {
        uint32_t run = 0;
        const bool want_last = tile_last && d < n_slots;
        const uint64_t base0 = want_last ? running[d] : 0;
        const uint32_t room = base0 >= limit ? 0u : limit - (uint32_t)base0;
        int candA = -1, candB = -1;
        uint32_t *__restrict__ row = hist + (uint64_t)d * n_tiles;
        for (uint32_t c0 = 0; c0 < n_tiles; c0 += 8 * WAVE) { // eight independent loads in flight, then the (ALU-only) scans
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const uint32_t i = c0 + u * WAVE + lane; v[u] = i < n_tiles ? row[i] : 0u; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t i = c0 + u * WAVE + lane;
                const uint32_t inc = wave_incl_scan_u32(v[u]);
                const uint32_t excl = run + inc - v[u];
                if (i < n_tiles) {
                    row[i] = excl;
                    if (excl < room) candA = (int)i; // i grows along the loop: the last assignment is the largest
                    if (v[u] > 0) candB = (int)i;
                }
                run += (uint32_t)__builtin_amdgcn_readlane((int)inc, WAVE - 1);
            }
        }
        int last = -1;
        if (want_last) {
            last = run >= room ? candA : candB;
            if (room == 0 || run == 0) last = -1;
            for (int o = 32; o >= 1; o >>= 1) { const int t = __shfl_xor(last, o, WAVE); last = t > last ? t : last; }
        }
        if (lane == 0) {
            totals[d] = run;
            if (acc_copy && d < n_slots) acc_copy[d] = run; // pg_count's device output (a rank's row of the all_gather buffer): no copy kernel
            if (plan.keep) { // sc1: read back by the last workgroup
                if (d < n_slots) {
                    __hip_atomic_store(reinterpret_cast<unsigned long long *>(acc_cnt + d), (unsigned long long)run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(tile_last + d, last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                if (want_last) tile_last[d] = last;
                if (acc_cnt && d < n_slots) acc_cnt[d] = run; // direct mode: digit == slot, so this is the batch's accepted-event count
            }
        }
    }
    if (!plan.keep) return; // (block-uniform)
    __shared__ uint32_t sh_ticket, wsum[PG_SCAN_WAVES], wfull[PG_SCAN_WAVES]; __shared__ int wmax[PG_SCAN_WAVES];
    __builtin_amdgcn_s_waitcnt(0); // every storing wave: its two stores have left ...
    __syncthreads();               // ... before the one lane that signals for the workgroup adds to the counter
    if (threadIdx.x == 0) sh_ticket = __hip_atomic_fetch_add(plan.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (sh_ticket != gridDim.x - (bp_out ? 2u : 1u)) return; // (the block-sum workgroup, if any, takes no ticket)
    // ---- the last workgroup: keep = min(cnt, limit - running), offsets, totals (as k_slot_plan with base == running) ---------------
    const uint32_t tid = threadIdx.x, w = tid >> 6;
    uint32_t carry = 0, full = 0; int tmax = -1;
    for (uint32_t c = 0; c < n_slots; c += PG_SCAN_WAVES * WAVE) {
        const uint32_t s_ = c + tid;
        uint32_t kp = 0; bool isfull = false; int tl = -1;
        if (s_ < n_slots) {
            const uint64_t cnt = __hip_atomic_load(reinterpret_cast<unsigned long long *>(acc_cnt + s_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tl = __hip_atomic_load(tile_last + s_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint64_t b = running[s_];
            const uint64_t rm = b >= limit ? 0 : (uint64_t)limit - b;
            kp = (uint32_t)(cnt < rm ? cnt : rm);
            isfull = limit > 0 && b + cnt >= limit; // at limit 0 no k-mer ever completes (gmove.cpp:925-927 skips every event)
            plan.running_out[s_] = b + cnt;
            plan.keep[s_] = kp;
        }
        const uint32_t inc = wave_incl_scan_u32(kp); // kept events of a batch number < 2^31
        const uint32_t nf = (uint32_t)__popcll(__ballot(isfull));
        for (int o = 32; o >= 1; o >>= 1) { const int t = __shfl_xor(tl, o, WAVE); tl = t > tl ? t : tl; }
        if (lane == WAVE - 1) { wsum[w] = inc; wfull[w] = nf; wmax[w] = tl; }
        __syncthreads();
        uint32_t off = 0, tot = 0;
        for (uint32_t ww = 0; ww < PG_SCAN_WAVES; ++ww) { if (ww < w) off += wsum[ww]; tot += wsum[ww]; full += wfull[ww]; tmax = wmax[ww] > tmax ? wmax[ww] : tmax; }
        if (s_ < n_slots) plan.ev_off[s_] = (uint64_t)carry + off + inc - kp;
        carry += tot;
        __syncthreads();
    }
    if (tid == 0) {
        plan.ev_off[n_slots] = carry; plan.plan_totals[0] = carry; plan.plan_totals[1] = full;
        plan.plan_totals[3] = (uint64_t)(int64_t)tmax; // read as a signed tile index by k_rank_emit
        __hip_atomic_store(plan.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // for the next batch
    }
}

// generic mode: exclusive prefix over the digits (<= 1024) -> first output index of each digit
__global__ __launch_bounds__(256) void k_sort_dbase(const uint32_t *__restrict__ totals, uint32_t ndig, uint32_t *__restrict__ dbase,
                                                    uint32_t *__restrict__ count_out) {
    __shared__ uint32_t wsum[4];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = ndig >= 256 ? ndig / 256 : 1;
    uint32_t v[4] = {0, 0, 0, 0}, s = 0;
    for (uint32_t i = 0; i < per; ++i) { const uint32_t d = tid * per + i; v[i] = d < ndig ? totals[d] : 0u; s += v[i]; }
    const uint32_t inc = wave_incl_scan_u32(s);
    if (lane_id() == WAVE - 1) wsum[tid >> 6] = inc;
    __syncthreads();
    uint32_t off = inc - s;
    for (uint32_t w = 0; w < (tid >> 6); ++w) off += wsum[w];
    for (uint32_t i = 0; i < per; ++i) { const uint32_t d = tid * per + i; if (d < ndig) dbase[d] = off; off += v[i]; }
    if (tid == 255) *count_out = off;
}

__global__ __launch_bounds__(256) void k_sort_scatter(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                      uint32_t n_scalar, const uint32_t *__restrict__ n_ptr, uint32_t shift,
                                                      int nbits, uint32_t n_tiles, const uint32_t *__restrict__ hist,
                                                      const uint32_t *__restrict__ dbase, const uint32_t *__restrict__ wcnt,
                                                      uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out) {
    // The tile's pairs are first put in digit order in LDS and leave from there: consecutive threads then store consecutive
    // elements of a digit's run (at 512 digits a tile of 4096 holds runs of ~8), instead of every lane of a row storing 4 bytes to a
    // place of its own -- the same bytes in a quarter to an eighth of the memory transactions.
    __shared__ uint32_t wbase[4][PG_RANK_MAX_DIGITS];
    __shared__ uint32_t gb[PG_RANK_MAX_DIGITS], ls[PG_RANK_MAX_DIGITS + 1]; // the tile's base in the output / in the LDS stage, per digit
    __shared__ uint32_t lkey[PG_SORT_TILE], lval[PG_SORT_TILE];
    __shared__ uint32_t wsum[4];
    const uint32_t tid = threadIdx.x, tile = blockIdx.x, w = tid >> 6;
    const int lane = lane_id();
    const uint32_t ndig = 1u << nbits;
    const uint32_t n = n_ptr ? *n_ptr : n_scalar;
    const uint32_t mask = ndig - 1u;
    const uint64_t base = (uint64_t)tile * PG_SORT_TILE + (uint64_t)w * PG_SORT_ROWS * WAVE;
    // all rows in flight before the (serial) ranking
    uint32_t kv[PG_SORT_ROWS], vv[PG_SORT_ROWS];
#pragma unroll
    for (int row = 0; row < PG_SORT_ROWS; ++row) {
        const uint64_t idx = base + (uint64_t)row * WAVE + lane;
        kv[row] = idx < n ? keys[idx] : PG_INVALID_SLOT;
        vv[row] = idx < n ? (vals ? vals[idx] : (uint32_t)idx) : 0u;
    }
    // per digit: where the tile's run starts in the output (gb), how long it is, and (exclusive scan over the digits) where it starts
    // in the stage (ls); digits are dealt out in consecutive chunks so that the scan is a wave scan + four sums
    const uint32_t per = ndig >= 256 ? ndig / 256 : 1;
    uint32_t cnt_d[4] = {0, 0, 0, 0}, sum = 0;
    for (uint32_t i = 0; i < per; ++i) {
        const uint32_t d = tid * per + i;
        if (d < ndig) {
            uint32_t b = dbase[d] + hist[(uint64_t)d * n_tiles + tile];
            gb[d] = b;
            for (uint32_t ww = 0; ww < 4; ++ww) { wbase[ww][d] = b; b += wcnt[((uint64_t)tile * 4 + ww) * ndig + d]; }
            cnt_d[i] = b - gb[d]; sum += cnt_d[i];
        }
    }
    const uint32_t inc = wave_incl_scan_u32(sum);
    if (lane == WAVE - 1) wsum[w] = inc;
    __syncthreads();
    {
        uint32_t off = inc - sum;
        for (uint32_t ww = 0; ww < w; ++ww) off += wsum[ww];
        for (uint32_t i = 0; i < per; ++i) { const uint32_t d = tid * per + i; if (d < ndig) { ls[d] = off; off += cnt_d[i]; } }
        if (tid == 255) ls[ndig] = off; // the tile's valid pairs (per * 256 >= ndig: thread 255 holds the last digits, or none)
    }
    __syncthreads();
    volatile uint32_t *mybase = wbase[w];
#pragma unroll
    for (int row = 0; row < PG_SORT_ROWS; ++row) {
        const uint32_t key = kv[row];
        const bool valid = key != PG_INVALID_SLOT;
        const uint32_t d = (key >> shift) & mask;
        const uint64_t peers = match_digit(d, valid, nbits);
        uint32_t b = 0;
        if (valid) b = mybase[d];                   // every peer reads the running base of its digit ...
        __builtin_amdgcn_wave_barrier();
        if (valid && lane == __ffsll((long long)peers) - 1) mybase[d] = b + (uint32_t)__popcll(peers); // ... then its leader advances it
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            const uint32_t dest = b + (uint32_t)__popcll(peers & lanemask_lt());
            const uint32_t j = ls[d] + (dest - gb[d]); // its place among the tile's pairs in digit order
            lkey[j] = key; lval[j] = vv[row];
        }
    }
    __syncthreads();
    const uint32_t total = ls[ndig];
    for (uint32_t j = tid; j < total; j += 256) {
        const uint32_t key = lkey[j], d = (key >> shift) & mask;
        const uint32_t dest = gb[d] + (j - ls[d]);
        keys_out[dest] = key;
        vals_out[dest] = lval[j];
    }
}

// window of a kept event (gmove.cpp:928-937) and its bookkeeping
__device__ __forceinline__ void write_kept(const PgDevBatch &B, const PgWalkParams &W, const PgWalkOut &O, const PgKeptOut &K,
                                           uint64_t e, uint64_t g) {
    const uint32_t rd = owner_of(B, O, g);
    const KeptRead kr = kept_read(O, rd);
    uint32_t start, len;
    if (!kept_window(B, W, O, kr, g, start, len)) { report_error(O, rd, PGR_ERR_RANGE); start = 0; len = 0; }
    const uint64_t L = kr.L;
    const uint64_t we64 = (uint64_t)start + len + W.print_margin;
    uint32_t we = (uint32_t)(we64 > L ? L : we64), ws = start - W.print_margin;
    // a kept event's window must be printable (gmove.cpp:928-944 is undefined for margin > start or an empty window)
    if (W.print_margin > start || we <= ws) { report_error(O, rd, PGR_ERR_WINDOW); ws = we = 0; }
    K.rec[e] = PgKeptRec{kr.sig0 + ws, we - ws, rd};
    if (K.read_needed) K.read_needed[rd] = 1;
}

// direct mode: rank = (events of the same slot in earlier tiles / waves / rows) + in-row rank; events with
// rank < keep[slot] are the first sample_limit ones in (read, event) order (gmove.cpp:925-927)
// Geometry: only the few tiles in front of the last useful one do any work, so the tile is spread over 16 waves of 4
// rows (a 1024-thread workgroup): the ordered loop, the one serial part, is 4 steps instead of 16.
#define PG_EMIT_WAVES 16
#define PG_EMIT_GRID 512
#define PG_EMIT_ROWS (PG_SORT_TILE / (PG_EMIT_WAVES * WAVE))
__global__ __launch_bounds__(PG_EMIT_WAVES * WAVE) void k_rank_emit(const uint32_t *__restrict__ keys, uint32_t n, int nbits, uint32_t n_slots, uint32_t n_tiles,
                                                   const uint32_t *__restrict__ hist,
                                                   const uint64_t *__restrict__ keep, const uint64_t *__restrict__ ev_off,
                                                   const uint64_t *__restrict__ totals, PgDevBatch B, PgWalkParams W, PgWalkOut O,
                                                   PgKeptOut K) {
    __shared__ uint32_t wbase[PG_EMIT_WAVES][PG_RANK_MAX_DIGITS];
    __shared__ uint32_t s_keep[PG_RANK_MAX_DIGITS], s_off[PG_RANK_MAX_DIGITS];
    static_assert(PG_RANK_MAX_DIGITS <= PG_EMIT_WAVES * WAVE, "one digit per thread");
    const uint32_t tid = threadIdx.x, w = tid >> 6;
    const int lane = lane_id();
    const uint32_t ndig = 1u << nbits;
    // the workgroups share the tiles round robin: at most PG_EMIT_GRID of them are started, and only the ones with a tile up to the
    // last that can still place an event (k_slot_plan: usually a few dozen of thousands) stay -- 1024-thread workgroups with 72 KB of
    // LDS that start only to leave are not free, least of all next to another stream's kernel. Measured and lost (22 -> 25..26 us):
    // asking for the tile's data in front of that test, and dropping the test for "stop at the first tile without room" -- either way
    // every started workgroup first gathers its 1024-line column of the [slot][tile] table.
    PG_PROBE_BEGIN(2);
    const int64_t last_tile = (int64_t)totals[3];
    for (uint32_t tile = blockIdx.x; tile < n_tiles && (int64_t)tile <= last_tile; tile += gridDim.x) {
    if (tile != blockIdx.x) __syncthreads(); // the previous tile's ranks are read out of wbase until its last wave is through
    PG_MARK(2, 0); // the last useful tile is known
    // phase 0: everything the ordered loop needs from global memory, all rows in flight at once. The slots of the tile's events
    // are requested in front of the "any room left?" test: a tile up to the last useful one nearly always passes it, and the
    // test's own loads and barrier then cost no round trip of their own.
    const uint32_t tile_first = O.tile_read[tile];
    const uint64_t base = (uint64_t)tile * PG_SORT_TILE + (uint64_t)w * PG_EMIT_ROWS * WAVE;
    uint32_t kv[PG_EMIT_ROWS], kp[PG_EMIT_ROWS], eo[PG_EMIT_ROWS];
#pragma unroll
    for (int row = 0; row < PG_EMIT_ROWS; ++row) {
        const uint64_t idx = base + (uint64_t)row * WAVE + lane;
        kv[row] = idx < n ? keys[idx] : PG_INVALID_SLOT;
    }
    // thread d (ndig <= 1024 = the workgroup: one digit each) brings slot d's column entry, keep and offset: the column entry stays
    // in its register for the wave bases below, the other two go to LDS, where the events look them up -- no second round to memory
    uint32_t hcol = 0; uint64_t kd = 0, od = 0;
    if (tid < n_slots) { hcol = hist[(uint64_t)tid * n_tiles + tile]; kd = keep[tid]; od = ev_off[tid]; }
    PG_MARK(2, 1); // keys, column, keep, offsets have arrived
    int any = 0; // does any slot still have room at this tile's position in the (read, event) order?
    if (tid < ndig) {
        if (tid < n_slots && (uint64_t)hcol < kd) any = 1;
        s_keep[tid] = (uint32_t)kd; // <= sample_limit
        s_off[tid] = (uint32_t)od;  // < number of kept events of the batch (< 2^32)
        for (uint32_t ww = 0; ww < PG_EMIT_WAVES; ++ww) wbase[ww][tid] = 0;
    }
    // the records of the events' reads (the read rides in the key's upper bits, relative to the tile's first read): requested now,
    // for every event of the tile, so that they arrive while the ranks are worked out in LDS
    uint32_t rd[PG_EMIT_ROWS]; KeptRead kr[PG_EMIT_ROWS]; bool have[PG_EMIT_ROWS];
#pragma unroll
    for (int row = 0; row < PG_EMIT_ROWS; ++row) {
        const uint32_t rel = kv[row] >> PG_SLOT_BITS;
        have[row] = kv[row] != PG_INVALID_SLOT && rel != PG_REL_UNKNOWN;
        rd[row] = tile_first + rel;
        if (have[row]) kr[row] = kept_read(O, rd[row]);
    }
    if (!__syncthreads_or(any)) break; // every k-mer this tile could feed is already full (gmove.cpp:925-927), and -- the column entries only grow along the tiles -- so are the tiles behind it
    volatile uint32_t *mybase = wbase[w];
#pragma unroll
    for (int row = 0; row < PG_EMIT_ROWS; ++row) {
        const bool valid = kv[row] != PG_INVALID_SLOT;
        kp[row] = valid ? s_keep[kv[row] & (ndig - 1u)] : 0u;
        eo[row] = valid ? s_off[kv[row] & (ndig - 1u)] : 0u;
    }
    // events of each slot in each wave of this tile (k_rank_count_direct keeps only the tile totals), then the rank of
    // each wave's first event of a slot = tile prefix + earlier waves
#pragma unroll
    for (int row = 0; row < PG_EMIT_ROWS; ++row) if (kv[row] != PG_INVALID_SLOT) atomicAdd(&wbase[w][kv[row] & (ndig - 1u)], 1u);
    __syncthreads();
    if (tid < ndig) { // sixteen independent reads, the prefix in registers, sixteen writes: not a read-write chain through LDS
        uint32_t cw[PG_EMIT_WAVES];
#pragma unroll
        for (int ww = 0; ww < PG_EMIT_WAVES; ++ww) cw[ww] = wbase[ww][tid];
        uint32_t b = hcol;
#pragma unroll
        for (int ww = 0; ww < PG_EMIT_WAVES; ++ww) { wbase[ww][tid] = b; b += cw[ww]; }
    }
    __syncthreads();
    PG_MARK(2, 2); // any-room test, per-wave counts, wave bases (three barriers)
    // phase 1: the ordered part -- LDS and ALU only: rank = tile prefix + earlier waves + earlier rows + in-row rank
    uint32_t dst[PG_EMIT_ROWS];
#pragma unroll
    for (int row = 0; row < PG_EMIT_ROWS; ++row) {
        const uint32_t key = kv[row];
        const bool valid = key != PG_INVALID_SLOT;
        const uint32_t d = key & (ndig - 1u);
        const uint64_t peers = match_digit(d, valid, nbits);
        uint32_t b = 0;
        if (valid) b = mybase[d];
        __builtin_amdgcn_wave_barrier();
        if (valid && lane == __ffsll((long long)peers) - 1) mybase[d] = b + (uint32_t)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
        const uint32_t rank = b + (uint32_t)__popcll(peers & lanemask_lt());
        dst[row] = (valid && rank < kp[row]) ? eo[row] + rank : 0xFFFFFFFFu;
    }
    // phase 2: the kept events' windows (gmove.cpp:928-937), staged across rows so that every stage is one set of independent loads: the
    // window (start / length arrays of the generic walk, or op_n and k_events' block sums) from the records that have arrived by now,
    // then the stores.
    PG_MARK(2, 3); // the ordered rows (and the read records, requested long ago)
#pragma unroll
    for (int row = 0; row < PG_EMIT_ROWS; ++row) // a kept event whose key does not name its read (k_events' per-op form in a crowded tile): look it up
        if (dst[row] != 0xFFFFFFFFu && !have[row]) { rd[row] = owner_of(B, O, base + (uint64_t)row * WAVE + lane); kr[row] = kept_read(O, rd[row]); }
    uint32_t ws[PG_EMIT_ROWS], wl[PG_EMIT_ROWS]; bool okr[PG_EMIT_ROWS];
#pragma unroll
    for (int row = 0; row < PG_EMIT_ROWS; ++row) {
        ws[row] = 0; wl[row] = 0; okr[row] = true;
        if (dst[row] != 0xFFFFFFFFu) okr[row] = kept_window(B, W, O, kr[row], base + (uint64_t)row * WAVE + lane, ws[row], wl[row]);
    }
    PG_MARK(2, 4); // the windows
#pragma unroll
    for (int row = 0; row < PG_EMIT_ROWS; ++row) {
        if (dst[row] == 0xFFFFFFFFu) continue;
        if (!okr[row]) { report_error(O, rd[row], PGR_ERR_RANGE); ws[row] = 0; wl[row] = 0; }
        const uint32_t Lr = kr[row].L;
        uint32_t start = ws[row] - W.print_margin;
        const uint64_t we64 = (uint64_t)ws[row] + wl[row] + W.print_margin;
        uint32_t we = (uint32_t)(we64 > Lr ? Lr : we64);
        // a kept event's window must be printable (gmove.cpp:928-944 is undefined for margin > start or an empty window)
        if (W.print_margin > ws[row] || we <= start) { report_error(O, rd[row], PGR_ERR_WINDOW); start = we = 0; }
        K.rec[dst[row]] = PgKeptRec{kr[row].sig0 + start, we - start, rd[row]}; // one 16-byte store
        if (K.read_needed) K.read_needed[rd[row]] = 1;
    }
    PG_MARK(2, 5); // stores
    if (tile == blockIdx.x) PG_PROBE_END(2, tile * PG_EMIT_WAVES + w);
    } // tiles of this workgroup
}

// =====================================================================================================
// generic mode bookkeeping on the sorted (slot, op index) pairs
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
                                                     uint32_t n_slots, uint64_t *__restrict__ acc_cnt, uint64_t *__restrict__ acc_copy) {
    const uint32_t s = blockIdx.x * 256 + threadIdx.x;
    if (s < n_slots) {
        acc_cnt[s] = (uint64_t)(slot_end[s] - slot_start[s]);
        if (acc_copy) acc_copy[s] = acc_cnt[s];
    }
}

// Multi-GPU job: the receive buffer of the all_gather (row g = accepted events of rank g per slot). The base of this rank
// (events of the ranks below it), the job's accepted events per k-mer and its freq.txt value (src/gmove.cpp:945-953: a file
// closes at sample_limit) come out of the same pass that applies the sample_limit cut: no kernel of their own.
__device__ __forceinline__ uint64_t slot_base(const PgGathered &G, const uint64_t *base, uint32_t s, uint32_t n_slots, uint32_t limit) {
    if (!G.all_counts) return base[s];
    uint64_t b = 0, t = 0;
    for (uint32_t g = 0; g < G.world; ++g) {
        if (g == G.rank) b = t;
        t += G.all_counts[(uint64_t)g * n_slots + s];
    }
    G.total[s] = t;
    G.freq[s] = t < limit ? t : limit;
    return b;
}

// single workgroup: the sample_limit cut (gmove.cpp:925-927, 945-950) and the output offsets
__global__ __launch_bounds__(1024) void k_slot_plan(const uint64_t *__restrict__ acc_cnt, const uint64_t *base, uint64_t *running,
                                                    uint32_t limit, uint32_t n_slots, uint64_t *__restrict__ keep,
                                                    uint64_t *__restrict__ ev_off, uint64_t *__restrict__ totals,
                                                    const uint32_t *__restrict__ hist, uint32_t n_tiles, const int32_t *__restrict__ tile_last,
                                                    PgGathered G) {
    __shared__ uint64_t wsum[16];
    __shared__ uint64_t wfull[16];
    (void)hist; (void)n_tiles;
    const uint32_t tid = threadIdx.x, w = tid >> 6;
    const int lane = lane_id();
    uint64_t carry = 0, full = 0;
    // tile_last (k_rank_scan) was worked out with the context's running count as the base. The base used here may be larger (events of
    // lower ranks, of other contexts): the last useful tile can then only lie earlier, and k_rank_emit's own "any room?" test ends
    // the tiles in between -- no k_tile_max launch. A SMALLER base (a caller's pg_collect with a base of its own) would make it too
    // early: then every tile counts as useful.
    int base_below_running = 0;
    for (uint32_t c = 0; c < n_slots; c += 1024) {
        const uint32_t s = c + tid;
        uint64_t kp = 0, isfull = 0;
        if (s < n_slots) {
            const uint64_t cnt = acc_cnt[s], b = slot_base(G, base, s, n_slots, limit);
            if (tile_last && running && b < running[s]) base_below_running = 1;
            const uint64_t room = b >= limit ? 0 : (uint64_t)limit - b;
            kp = cnt < room ? cnt : room;
            isfull = (limit > 0 && b + cnt >= limit) ? 1 : 0; // at limit 0 no k-mer ever completes: the test at gmove.cpp:925-927 skips every event before 945-950 can count it
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
    __syncthreads();
    int last = -1; // direct mode with the last useful tiles already known (k_rank_scan): their maximum
    if (tile_last) {
        __shared__ int wlast[16];
        for (uint32_t s2 = tid; s2 < n_slots; s2 += 1024) { const int t = tile_last[s2]; last = t > last ? t : last; }
        for (int o = 32; o >= 1; o >>= 1) { const int t = __shfl_xor(last, o, WAVE); last = t > last ? t : last; }
        if (lane == 0) wlast[w] = last;
        __syncthreads();
        if (tid == 0) for (int ww = 0; ww < 16; ++ww) last = wlast[ww] > last ? wlast[ww] : last;
    }
    base_below_running = __syncthreads_or(base_below_running);
    if (tid == 0) { // totals[3]: read as a signed tile index by k_rank_emit; ~0 = -1: k_tile_max raises it
        ev_off[n_slots] = carry; totals[0] = carry; totals[1] = full;
        totals[3] = tile_last ? (base_below_running ? ~0ull >> 1 : (uint64_t)(int64_t)last) : (hist ? ~0ull : ~0ull >> 1);
    }
}

// Many slots (k = 9: 262 144): the same cut element-wise over a grid, the offsets by the generic scan, the totals by one
// atomic per workgroup (the single-workgroup loop above needs 256 trips there: 0.6 ms).
__global__ __launch_bounds__(256) void k_slot_keep(const uint64_t *__restrict__ acc_cnt, const uint64_t *base, uint64_t *running, uint32_t limit,
                                                   uint32_t n_slots, uint64_t *__restrict__ keep, uint32_t *__restrict__ keep32,
                                                   uint64_t *__restrict__ totals, PgGathered G) {
    __shared__ uint32_t wfull[4];
    const uint32_t s = blockIdx.x * 256 + threadIdx.x;
    bool isfull = false;
    if (s < n_slots) {
        const uint64_t cnt = acc_cnt[s], b = slot_base(G, base, s, n_slots, limit);
        const uint64_t room = b >= limit ? 0 : (uint64_t)limit - b;
        const uint64_t kp = cnt < room ? cnt : room;
        isfull = limit > 0 && b + cnt >= limit; // see k_slot_plan
        if (running) running[s] = b + cnt;
        keep[s] = kp; keep32[s] = (uint32_t)kp; // <= sample_limit
    }
    const uint32_t nfull = (uint32_t)__popcll(__ballot(isfull));
    if (lane_id() == 0) wfull[threadIdx.x >> 6] = nfull;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(reinterpret_cast<unsigned long long *>(&totals[1]), (unsigned long long)(wfull[0] + wfull[1] + wfull[2] + wfull[3]));
}
#define SCAN_CHUNK 4096 // elements per workgroup of the scan kernels
// The same cut and its offsets in ONE launch (round 3; k_slot_keep + a scan + k_slot_totals + a memset were four: 34 us at k = 9): a chained
// scan (decoupled look-back, as k_scan_chained: state[0] ticket, [1] blocks done, [2 + b] block entries, [2 + n_scan] full slots) whose
// elements are worked out on the way in. 16 consecutive slots per thread; the state is left zeroed for the next launch.
__global__ __launch_bounds__(256) void k_slot_cut(const uint64_t *__restrict__ acc_cnt, const uint64_t *base, uint64_t *running, uint32_t limit, uint32_t n_slots,
                                                  uint64_t *__restrict__ keep, uint32_t *__restrict__ keep32, uint64_t *__restrict__ ev_off, uint64_t *__restrict__ totals,
                                                  uint64_t *__restrict__ state, uint32_t n_scan, PgGathered G) {
    __shared__ uint64_t wsum[4];
    __shared__ uint64_t sh_prefix;
    __shared__ uint32_t sh_block, wfull[4];
    if (threadIdx.x == 0) sh_block = (uint32_t)atomicAdd(reinterpret_cast<unsigned long long *>(state), 1ull);
    __syncthreads();
    const uint32_t b = sh_block;
    const uint64_t s0 = (uint64_t)b * SCAN_CHUNK + (uint64_t)threadIdx.x * 16;
    uint32_t v[16]; uint64_t s = 0; uint32_t nfull = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint64_t sl = s0 + i;
        v[i] = 0;
        if (sl < n_slots) {
            const uint64_t cnt = acc_cnt[sl], bb = slot_base(G, base, (uint32_t)sl, n_slots, limit);
            const uint64_t room = bb >= limit ? 0 : (uint64_t)limit - bb;
            const uint64_t kp = cnt < room ? cnt : room;
            nfull += (limit > 0 && bb + cnt >= limit) ? 1u : 0u; // see k_slot_plan
            if (running) running[sl] = bb + cnt;
            keep[sl] = kp; keep32[sl] = (uint32_t)kp; v[i] = (uint32_t)kp; // <= sample_limit
        }
        s += v[i];
    }
    const uint64_t inc = wave_incl_scan_u64(s);
    for (int o = 32; o >= 1; o >>= 1) nfull += __shfl_xor(nfull, o, WAVE);
    if (lane_id() == WAVE - 1) { wsum[threadIdx.x >> 6] = inc; wfull[threadIdx.x >> 6] = nfull; }
    __syncthreads();
    const uint64_t block_sum = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    uint64_t *st = state + 2;
    if (threadIdx.x < WAVE) { // wave 0: publish, then look back 64 blocks at a time (k_scan_chained)
        const int lane = threadIdx.x;
        if (lane == 0) __hip_atomic_store(st + b, (block_sum << 2) | (b == 0 ? 2ull : 1ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint64_t excl = 0;
        int64_t hi = (int64_t)b - 1;
        while (hi >= 0) {
            const int64_t j = hi - lane;
            uint64_t wv = 0;
            if (j >= 0) do { wv = __hip_atomic_load(st + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((wv & 3ull) == 0);
            const uint64_t pm = __ballot(j >= 0 && (wv & 3ull) == 2ull);
            const int stop = pm ? __ffsll((long long)pm) - 1 : WAVE;
            uint64_t part = (j >= 0 && lane <= stop) ? (wv >> 2) : 0ull;
            part = wave_incl_scan_u64(part);
            excl += (uint64_t)__shfl(part, WAVE - 1, WAVE);
            if (pm) break;
            hi -= WAVE;
        }
        if (lane == 0) {
            if (b != 0) __hip_atomic_store(st + b, ((excl + block_sum) << 2) | 2ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sh_prefix = excl;
        }
    }
    __syncthreads();
    uint64_t run = sh_prefix + inc - s;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) run += wsum[w];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint64_t sl = s0 + i;
        if (sl < n_slots) ev_off[sl] = run;
        run += v[i];
        if (sl + 1 == n_slots) { ev_off[n_slots] = run; totals[0] = run; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // this block's full slots are in the job's count before the block is counted as done (a returning atomic: its result is waited for)
        const unsigned long long seen = atomicAdd(reinterpret_cast<unsigned long long *>(st + n_scan), (unsigned long long)(wfull[0] + wfull[1] + wfull[2] + wfull[3]));
        const unsigned long long done = atomicAdd(reinterpret_cast<unsigned long long *>(state + 1), 1ull + (seen >> 63)); // (seen >> 63 == 0: keeps the order)
        sh_block = done + 1 == n_scan;
    }
    __syncthreads();
    if (sh_block) { // last block out: the totals, and nobody reads the state any more
        if (threadIdx.x == 0) {
            totals[1] = __hip_atomic_load(st + n_scan, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            totals[3] = ~0ull >> 1; // (read as a signed tile index by the direct ranking only)
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n_scan + 3; i += 256) __hip_atomic_store(state + i, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__global__ void k_slot_totals(const uint64_t *__restrict__ ev_off, uint32_t n_slots, uint64_t *__restrict__ totals) {
    totals[0] = ev_off[n_slots]; totals[3] = ~0ull >> 1;
}

// direct mode: totals[3] = last tile that can still place an event = max over slots of the last tile whose exclusive
// prefix in the slot's row is below keep[slot]. One thread per slot, spread over many small workgroups (the probes are
// scattered loads: a single workgroup would serialise them on one CU).
__global__ __launch_bounds__(64) void k_tile_max(const uint32_t *__restrict__ hist, uint32_t n_tiles, const uint64_t *__restrict__ keep,
                                                 uint32_t n_slots, uint64_t *__restrict__ totals) {
    const uint32_t s = blockIdx.x * 64 + threadIdx.x;
    int last = -1;
    if (s < n_slots) {
        const uint64_t kp = keep[s];
        if (kp > 0) {
            // the row is non-decreasing: bracket the crossing with 12 independent probes at 2^j, then narrow 8 ways per
            // round (three dependent round trips instead of a binary search's eleven)
            const uint32_t *row = hist + (uint64_t)s * n_tiles;
            uint32_t pv[12];
#pragma unroll
            for (int j = 0; j < 12; ++j) { const uint32_t p = 1u << j; pv[j] = row[p < n_tiles ? p : n_tiles - 1]; }
            uint32_t lo = 0, hi = n_tiles; // row[lo] < kp (row[0] == 0), row[hi] >= kp or hi == n_tiles
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                const uint32_t p = 1u << j;
                if (p < n_tiles) { if ((uint64_t)pv[j] < kp) lo = p; else if (p < hi) hi = p; }
            }
            while (hi - lo > 1) {
                const uint32_t step = (hi - lo + 8) / 9;
                uint32_t qv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) { const uint32_t p = lo + (i + 1) * step; qv[i] = row[p < hi ? p : lo]; }
                uint32_t nlo = lo, nhi = hi;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const uint32_t p = lo + (i + 1) * step;
                    if (p < hi) { if ((uint64_t)qv[i] < kp) nlo = p; else if (p < nhi) nhi = p; }
                }
                lo = nlo; hi = nhi;
            }
            last = (int)lo;
        }
    }
    // one atomic per wave
    for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(last, d, WAVE); last = o > last ? o : last; }
    if (threadIdx.x == 0 && last >= 0) atomicMax(reinterpret_cast<long long *>(&totals[3]), (long long)last);
}

// The kept events of the sorted (slot, op index) pairs, in two passes. Working the windows out per SORTED position (the first form) is
// a dozen scattered loads per event -- owner, read record, block sums, op_n -- at 15 M kept events (k = 9): 1.57 ms. Per OP INDEX the
// same loads are shared by the neighbouring threads (consecutive ops of one read), so: pass 1, per sorted position, only says where
// the event goes (dst[g] = ev_off[slot] + rank, or "not kept": one scattered 4-byte store); pass 2, per op index, works the window
// out and stores it there.
__global__ __launch_bounds__(256) void k_kept_pos(const uint32_t *__restrict__ skey, const uint32_t *__restrict__ sval,
                                                  const uint32_t *__restrict__ m_ptr, const uint32_t *__restrict__ slot_start,
                                                  const uint64_t *__restrict__ keep, const uint64_t *__restrict__ ev_off, uint32_t *__restrict__ dst) {
    const uint64_t pos = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (pos >= *m_ptr) return;
    const uint32_t s = skey[pos];
    const uint64_t rank = pos - slot_start[s];
    dst[sval[pos]] = rank < keep[s] ? (uint32_t)(ev_off[s] + rank) : 0xFFFFFFFFu; // kept events of a batch number < 2^32 - 1 (<= n_ops)
}
__global__ __launch_bounds__(256) void k_kept_fill(const uint32_t *__restrict__ ev_slot, const uint32_t *__restrict__ dst, uint64_t n_ops,
                                                   PgDevBatch B, PgWalkParams W, PgWalkOut O, PgKeptOut K) {
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= n_ops || ev_slot[g] == PG_INVALID_SLOT) return; // dst[g] is only written for accepted events
    const uint32_t e = dst[g];
    if (e != 0xFFFFFFFFu) write_kept(B, W, O, K, e, g);
}

// =====================================================================================================
// exclusive scan u32 -> u64 over n elements, out has n+1 entries
// =====================================================================================================

__global__ __launch_bounds__(256) void k_scan_partials(const uint32_t *__restrict__ in, uint32_t stride, uint64_t n_scalar,
                                                       const uint64_t *__restrict__ n_ptr, uint64_t *__restrict__ partial) {
    __shared__ uint64_t wsum[4];
    const uint64_t n = n_ptr ? *n_ptr : n_scalar;
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_CHUNK;
    uint64_t s = 0;
    for (uint32_t i = threadIdx.x; i < SCAN_CHUNK; i += 256) { const uint64_t a = base + i; if (a < n) s += in[a * stride]; }
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

__global__ __launch_bounds__(256) void k_scan_apply(const uint32_t *__restrict__ in, uint32_t stride, uint64_t n_scalar,
                                                    const uint64_t *__restrict__ n_ptr, const uint64_t *__restrict__ partial,
                                                    uint64_t *__restrict__ out, uint64_t *__restrict__ total_out) {
    __shared__ uint64_t wsum[4];
    const uint64_t n = n_ptr ? *n_ptr : n_scalar;
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_CHUNK + (uint64_t)threadIdx.x * 16;
    uint32_t v[16];
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const uint64_t a = base + i; v[i] = a < n ? in[a * stride] : 0u; s += v[i]; }
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
        if (a + 1 == n) { out[n] = run; if (total_out) *total_out = run; }
    }
    if (n == 0 && blockIdx.x == 0 && threadIdx.x == 0) { out[0] = 0; if (total_out) *total_out = 0; }
}

// The same scan in ONE launch (chained scan with decoupled look-back): block b publishes the sum of its 4096 elements,
// then adds up the published sums (or the first published prefix) of the blocks before it. state[0] = ticket (block
// order = order of arrival, so a block only ever waits for blocks that already run), state[1] = blocks done,
// state[2 + b] = (value << 2) | flag, flag 1 = sum of block b, 2 = inclusive prefix up to block b. The block that
// finishes last clears the state for the next launch (the buffer is zeroed when it is allocated).
#define PG_RARE_LDS_WORDS (PG_STATS_BINS + WAVE + 32 + 4) // = StatsGeom<PG_STATS_BINS>::LDS_WORDS (checked where that is defined)
__device__ __forceinline__ void rare_worker(uint32_t worker, uint32_t n_wide, uint32_t *hist, const PgRareArgs &A); // with the statistics kernels
// RARE: blocks [n_scan, gridDim.x) are not part of the scan: each of their 4 waves is a worker of the rare statistics launch
// (reads whose in-range interval needs more than 1024 bins; usually none) -- an empty launch of its own costs a kernel boundary.
template <bool RARE> __global__ __launch_bounds__(256) void k_scan_chained(const uint32_t *__restrict__ in, uint32_t stride, uint64_t n_scalar, const uint64_t *__restrict__ n_ptr,
                                                      uint64_t *__restrict__ out, uint64_t *__restrict__ state, uint32_t n_scan, PgRareArgs A,
                                                      uint64_t *__restrict__ total_out) {
    if (RARE) {
        __shared__ __attribute__((aligned(16))) uint32_t rare_hist[4][PG_RARE_LDS_WORDS];
        if (blockIdx.x >= n_scan) {
            const uint32_t workers = (gridDim.x - n_scan) * 4u;
            rare_worker((blockIdx.x - n_scan) * 4u + (threadIdx.x >> 6), workers - PG_HUGE_BLOCKS, rare_hist[threadIdx.x >> 6], A);
            return;
        }
    }
    __shared__ uint64_t wsum[4];
    __shared__ uint64_t sh_prefix;
    __shared__ uint32_t sh_block;
    const uint64_t n = n_ptr ? *n_ptr : n_scalar;
    if (threadIdx.x == 0) sh_block = (uint32_t)atomicAdd(reinterpret_cast<unsigned long long *>(state), 1ull);
    __syncthreads();
    const uint32_t b = sh_block;
    const uint64_t base = (uint64_t)b * SCAN_CHUNK + (uint64_t)threadIdx.x * 16;
    uint32_t v[16];
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const uint64_t a = base + i; v[i] = a < n ? in[a * stride] : 0u; s += v[i]; }
    const uint64_t inc = wave_incl_scan_u64(s);
    if (lane_id() == WAVE - 1) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    const uint64_t block_sum = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    uint64_t *st = state + 2;
    if (threadIdx.x < WAVE) { // wave 0: publish, then look back 64 blocks at a time
        const int lane = threadIdx.x;
        if (lane == 0) __hip_atomic_store(st + b, (block_sum << 2) | (b == 0 ? 2ull : 1ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint64_t excl = 0;
        int64_t hi = (int64_t)b - 1; // highest block not yet accounted for
        while (hi >= 0) {
            const int64_t j = hi - lane;
            uint64_t w = 0;
            // every lane waits for its block's entry; entries only move 0 -> sum -> prefix
            if (j >= 0) do { w = __hip_atomic_load(st + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((w & 3ull) == 0);
            const uint64_t pm = __ballot(j >= 0 && (w & 3ull) == 2ull);
            const int stop = pm ? __ffsll((long long)pm) - 1 : WAVE; // nearest block that already holds a prefix
            uint64_t part = (j >= 0 && lane <= stop) ? (w >> 2) : 0ull;
            part = wave_incl_scan_u64(part);
            excl += (uint64_t)__shfl(part, WAVE - 1, WAVE);
            if (pm) break;
            hi -= WAVE;
        }
        if (lane == 0) {
            if (b != 0) __hip_atomic_store(st + b, ((excl + block_sum) << 2) | 2ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sh_prefix = excl;
        }
    }
    __syncthreads();
    uint64_t off = sh_prefix;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) off += wsum[w];
    uint64_t run = off + inc - s;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint64_t a = base + i;
        if (a < n) out[a] = run;
        run += v[i];
        if (a + 1 == n) { out[n] = run; if (total_out) *total_out = run; }
    }
    if (n == 0 && b == 0 && threadIdx.x == 0) { out[0] = 0; if (total_out) *total_out = 0; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long done = atomicAdd(reinterpret_cast<unsigned long long *>(state + 1), 1ull);
        sh_block = done + 1 == n_scan;
    }
    __syncthreads();
    if (sh_block) // last block out: nobody reads the state any more
        for (uint32_t i = threadIdx.x; i < n_scan + 2; i += 256) __hip_atomic_store(state + i, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// =====================================================================================================
// read statistics: pA conversion + zero-fill + exact median / MAD (gmove.cpp:754-771)
// =====================================================================================================

__global__ void k_stat_flags_init(int32_t *__restrict__ flags) { flags[0] = INT_MAX; flags[1] = 0; flags[2] = 0; flags[3] = 0; } // [3]: a read's calibration is outside pg_div_domain_ok (the dense gathers then divide)

// the record of one read. Reads whose in-range interval does not fit the 1024-bin LDS histogram put THEMSELVES on the lists of
// the (rare) wide launch when k_read_stats meets them (stats_list_wide): nothing here depends on another thread, so the
// records can also be written by k_batch_init's launch (eager statistics: one kernel boundary less per batch)
__device__ __forceinline__ void read_plan_one(const PgDevBatch &B, uint32_t r, const uint8_t *__restrict__ needed, double pa_min, double pa_max,
                                              PgStatRec *__restrict__ rec, int32_t *__restrict__ stat_status, const PgLongState &LS) {
    const double offset = B.off[r], scale = B.range[r] / B.dig[r];
    const PgReadPlan p = pg_make_plan(B.dig[r], offset, B.range[r], pa_min, pa_max);
    PgStatRec o;
    o.beg = B.sig_off[r]; o.end = B.sig_off[r + 1];
    o.c_lo = p.c_lo; o.span = p.span; o.z0 = p.z0;
    const bool skip = (needed && !needed[r]) || o.end == o.beg;
    o.mode = skip ? PG_STAT_SKIP : (p.status != 0 ? PG_STAT_BAD : PG_STAT_RUN);
    o.offset = offset; o.scale = scale; o.inv = 1.0 / scale;
    o.sym = (p.status == 0 && p.span > 0 && pg_sym_guard(offset, scale, pg_pa(p.c_lo, offset, scale), pg_pa(p.c_lo + p.span - 1, offset, scale))) ? 1u : 0u;
    o.split = 0;
    if (LS.tab && o.mode == PG_STAT_RUN && o.span > 0 && o.span <= 1024 && o.end - o.beg > PG_LONG_MIN) { // a long read: helpers for its slices 1 .. S - 1
        uint32_t S; uint64_t slen;
        pg_long_geometry(o.end - o.beg, &S, &slen);
        const uint32_t first = (uint32_t)atomicAdd(LS.cnt, (int)(S - 1u)); // (a batch of 2^31 slices is not representable: pg_count refuses 2^31 ops)
        const bool room = first < LS.cap && S - 1u <= LS.cap - first;
        for (uint32_t j = 0; j + 1 < S && first + j < LS.cap; ++j) LS.tab[first + j] = room ? make_uint2(r, j + 1u) : make_uint2(PG_LONG_INVALID, 0u);
        if (room) { o.split = first + 1u; atomicAdd(LS.cnt + 1, 1); }
    }
    rec[r] = o;
    stat_status[r] = 0;
}

__global__ __launch_bounds__(256) void k_read_plan(PgDevBatch B, const uint8_t *__restrict__ needed, double pa_min, double pa_max,
                                                   PgStatRec *__restrict__ rec, int32_t *__restrict__ stat_status, int32_t *__restrict__ reset_flags, PgLongState LS) {
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r == 0 && reset_flags) { reset_flags[0] = INT_MAX; reset_flags[1] = 0; reset_flags[2] = 0; reset_flags[3] = 0; } // as k_stat_flags_init: the lists are k_read_stats' (a later launch)
    if (r < B.n_reads) read_plan_one(B, r, needed, pa_min, pa_max, rec, stat_status, LS);
}

// a read of the main statistics launch whose interval needs a wider histogram: the list is filled from the front
// (<= PG_STATS_BINS codes, LDS) and from the back (more: global-memory histogram); wide_count[0] / [1] count the two ends
__device__ __forceinline__ void stats_list_wide(uint32_t r, int32_t span, uint32_t n_reads, uint32_t *__restrict__ wide_list, int32_t *__restrict__ wide_count) {
    if (span > PG_STATS_BINS) wide_list[n_reads - 1 - (uint32_t)atomicAdd(wide_count + 1, 1)] = r;
    else wide_list[atomicAdd(wide_count, 1)] = r;
}

// prefix accessor over the padded LDS histogram: lane l owns BPL consecutive bins, stored with one pad
// dword per lane so that the column-wise reads/writes of the scan are bank-conflict free
template <int LOG_BPL> struct PaddedPre {
    const uint32_t *h;
    __device__ __forceinline__ uint32_t operator[](int b) const { return h[b + (b >> LOG_BPL)]; }
};

// One WAVE per read (64-thread workgroups): no barriers, no idle waves. The signal is streamed once with
// 16-byte loads (8 int16 per lane, up to 8 loads in flight per lane) and binned by raw code into an LDS
// histogram of the in-range code interval [c_lo, c_lo+span); an inclusive prefix sum of the histogram
// then yields both order statistics (pg_select.h) without touching the signal again.
// prefix accessor over a global-memory histogram (reads served by L2: the wave's own atomics/stores are visible there)
struct GlobalPre {
    const uint32_t *h;
    __device__ __forceinline__ uint32_t operator[](int b) const {
        return __hip_atomic_load(h + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};
#define PG_HUGE_BINS 65536 // every int16 code: the global-memory fallback for very wide [pa_min, pa_max]
#define PG_HUGE_WORDS (PG_HUGE_BINS + 64)

template <int BINS> struct StatsGeom {
    static constexpr int BPL = BINS / WAVE;           // bins per lane in the scan
    static constexpr int LOG_BPL = BPL == 16 ? 4 : (BPL == 32 ? 5 : 6);
    static constexpr int TRASH = BINS + WAVE;         // padded size of the real bins
    static constexpr int LDS_WORDS = TRASH + 32 + 4;  // + 32 dummy bins (padded) for out-of-range samples
};

#ifdef PG_COUNT_FALLBACKS
__device__ unsigned long long g_pg_fallbacks; // measurement build only: reads whose selection took the general path
extern "C" unsigned long long pg_debug_fallbacks(int reset) {
    unsigned long long v = 0;
    (void)hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_pg_fallbacks), sizeof(v));
    if (reset) { unsigned long long z = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pg_fallbacks), &z, sizeof(z)); }
    return v;
}
#endif
// ---- pieces shared by the three statistics kernels -----------------------------------------------------------------
// BINS == 1024 / PG_STATS_BINS: padded LDS histogram; BINS == PG_HUGE_BINS: global-memory histogram (one per block)
template <int BINS> struct StatsCfg {
    static constexpr bool GLOBAL = BINS == PG_HUGE_BINS;
    static constexpr int BPL = GLOBAL ? 1 : BINS / WAVE;
    static constexpr int LOG_BPL = GLOBAL ? 31 : (BPL == 16 ? 4 : (BPL == 32 ? 5 : 6));
    static constexpr int WORDS = GLOBAL ? PG_HUGE_WORDS : BINS + WAVE + 32 + 4;
    using Pre = typename std::conditional<GLOBAL, GlobalPre, PaddedPre<LOG_BPL>>::type;
};

template <int BINS> __device__ __forceinline__ void stats_zero(uint32_t *hist, int lane) {
    using C = StatsCfg<BINS>;
    uint4 *h4 = reinterpret_cast<uint4 *>(hist);
    for (int i = lane; i < C::WORDS / 4; i += WAVE) h4[i] = make_uint4(0, 0, 0, 0);
    if (C::GLOBAL) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); __builtin_amdgcn_s_waitcnt(0); }
}

// one sample into the histogram: out-of-range samples (idx wraps to a huge unsigned value) are clamped onto 32
// dummy bins behind the real ones, one per lane pair, so the path is branch-free
template <int BINS> __device__ __forceinline__ void stats_bin1(uint32_t *hist, int code, int c_lo, int lane) {
    using C = StatsCfg<BINS>;
    const uint32_t idx = min((uint32_t)(code - c_lo), (uint32_t)BINS + (lane & 31u));
    atomicAdd(&hist[C::GLOBAL ? idx : idx + (idx >> C::LOG_BPL)], 1u);
}

typedef unsigned short pg_us2 __attribute__((ext_vector_type(2)));

// eight samples (one 16-byte vector) into the LDS histogram with packed 16-bit arithmetic: per PAIR of samples
// v_pk_sub_u16, v_pk_min_u16, v_pk_lshrrev_b16, v_pk_add_u16, v_pk_lshlrev_b16 + two extracts = 3.5 VALU per sample.
// code - c_lo is taken modulo 2^16: an out-of-range code wraps to a value >= span (c_lo + span <= 32768), so it
// lands on a bin the selection never reads or, through the min, on this lane's dummy bin.
template <int BINS> __device__ __forceinline__ void stats_bin8(uint32_t *hist, const int4 &q, uint32_t c2, uint32_t cap2) {
    using C = StatsCfg<BINS>;
    static_assert(!C::GLOBAL, "packed binning is for the LDS histograms");
    const pg_us2 cv = __builtin_bit_cast(pg_us2, c2), capv = __builtin_bit_cast(pg_us2, cap2);
    // the four pairs side by side: dependent packed ops of one pair are never back to back (no hazard wait states)
    const int w[4] = {q.x, q.y, q.z, q.w};
    pg_us2 d[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) d[i] = __builtin_bit_cast(pg_us2, w[i]) - cv;
#pragma unroll
    for (int i = 0; i < 4; ++i) d[i] = __builtin_elementwise_min(d[i], capv);
    pg_us2 t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = d[i] >> (unsigned short)C::LOG_BPL;
#pragma unroll
    for (int i = 0; i < 4; ++i) d[i] = d[i] + t[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) d[i] = d[i] << (unsigned short)2;
    char *hb = reinterpret_cast<char *>(hist);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t aw = __builtin_bit_cast(uint32_t, d[i]);
        atomicAdd(reinterpret_cast<uint32_t *>(hb + (aw & 0xffffu)), 1u);
        atomicAdd(reinterpret_cast<uint32_t *>(hb + (aw >> 16)), 1u);
    }
}

// inclusive prefix over the bins, in place
// returns the inclusive prefix at the last bin this lane owns (LDS variants; 0 for the global-memory histogram)
template <int BINS> __device__ __forceinline__ uint32_t stats_prefix(uint32_t *hist, int lane, uint32_t span) {
    using C = StatsCfg<BINS>;
    if (C::GLOBAL) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); __builtin_amdgcn_s_waitcnt(0); }
    else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if constexpr (C::GLOBAL) {
        // 64 bins at a time with a running carry
        uint32_t carry = 0;
        for (uint32_t c = 0; c < span; c += WAVE) {
            const uint32_t b = c + lane;
            const uint32_t v = b < span ? __hip_atomic_load(hist + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            const uint32_t inc = wave_incl_scan_u32(v);
            if (b < span) __hip_atomic_store(hist + b, carry + inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            carry += (uint32_t)__builtin_amdgcn_readlane((int)inc, WAVE - 1);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); __builtin_amdgcn_s_waitcnt(0);
    } else {
        // lane l owns bins [l*BPL, (l+1)*BPL) at padded address l*(BPL+1)+i
        constexpr int BPL = C::BPL;
        uint32_t ssum = 0;
#pragma unroll
        for (int i = 0; i < BPL; ++i) ssum += hist[lane * (BPL + 1) + i];
        uint32_t run = wave_incl_scan_u32(ssum) - ssum;
#pragma unroll 8
        for (int i = 0; i < BPL; ++i) { run += hist[lane * (BPL + 1) + i]; hist[lane * (BPL + 1) + i] = run; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        return run;
    }
    __builtin_amdgcn_wave_barrier();
    return 0;
}

// order statistics from the prefix sums: every search step tests 64 candidates (pg_select.h holds the arithmetic)
template <int BINS>
__device__ __forceinline__ void stats_select(const uint32_t *hist, int lane, const PgReadPlan &pl, uint64_t L, double offset, double scale,
                                             int win, double &med_out, double &mad_out) {
    using Pre = typename StatsCfg<BINS>::Pre;
    PgSel<Pre> sel;
    sel.pre = Pre{hist}; sel.span = pl.span; sel.c_lo = pl.c_lo; sel.z0 = pl.z0; sel.L = L;
    sel.offset = offset; sel.scale = scale;
    sel.begin();
    int bm = 0;
    if (!sel.zmed) bm = wave_first_true(sel.span, [&](int b) { return sel.med_pred(b); });
    sel.set_median(bm);
    double best = INFINITY;
    if (sel.L > 1) {
        sel.begin_mad();
        // 1. narrow each side with an integer-only model of the predicate (prefix lookups, no FP64)
        sel.begin_approx();
        int aU, aD;
        wave_first_true_pair(sel.nU, sel.nD, [&](int side, int tt) { return sel.approx_pred(side == 0, tt); }, aU, aD);
        // 2. ONE exact round: lanes 0..30 test U candidates around aU, lanes 32..62 D candidates around aD,
        //    lane 31 the zero-filled class. The exact predicate decides; the model only placed the window.
        const bool up = lane < 32;
        const int sub = lane & 31, n_side = up ? sel.nU : sel.nD;
        const int t = (up ? aU : aD) - win + sub;
        const bool cand = sub < 2 * win + 1 && sub < 31 && t >= 0 && t < n_side;
        double v = INFINITY; int hU = -1, hD = -1; bool act = cand;
        if (cand) { v = sel.dev(up, t); if (up) hU = t + 1; else hD = t + 1; }
        if (lane == 31 && sel.nZ > 0) { v = sel.dZ; act = true; }
        const bool ok = act && sel.N(v, hU, hD) >= sel.need; // one pass of the exact predicate for all 63 candidates
        const uint64_t okm = __ballot(ok), cm = __ballot(cand);
        bool certain[2];
        for (int side = 0; side < 2; ++side) {
            const uint64_t lanes = side == 0 ? 0x7fffffffull : (0x7fffffffull << 32);
            const uint64_t o = okm & lanes, c = cm & lanes;
            const int n = side == 0 ? sel.nU : sel.nD, a = side == 0 ? aU : aD;
            if (n == 0) { certain[side] = true; continue; }
            if (o) {
                const int fl = __ffsll((long long)o) - 1;           // first lane whose candidate qualifies
                const int ft = a - win + (fl & 31);
                const bool prev_tested = fl > (__ffsll((long long)c) - 1); // the candidate before it was tested (and failed)
                certain[side] = ft == 0 || prev_tested;
                if (certain[side]) { const double vv = __shfl(v, fl, WAVE); if (vv < best) best = vv; }
            } else {
                // no candidate in the window qualifies: conclusive only if the window reached the last code
                const int last_t = c ? a - win + ((63 - __clzll((long long)c)) & 31) : -1;
                certain[side] = last_t == n - 1;
            }
        }
        // 3. fall back to the full exact search for a side the window could not decide (rare)
        for (int side = 0; side < 2; ++side) {
            if (certain[side]) continue;
            const bool u2 = side == 0;
            const int n = u2 ? sel.nU : sel.nD;
            const int tt = wave_first_true(n, [&](int x) { return sel.mad_pred(u2, x); });
            if (tt < n) { const double vv = sel.dev(u2, tt); if (vv < best) best = vv; }
        }
        if ((okm >> 31) & 1) { const double vz = sel.dZ; if (vz < best) best = vz; }
    }
    const PgMedMad mm = sel.finish(best);
    med_out = mm.med; mad_out = mm.mad;
}

// The usual case in ~1/3 of the instructions of stats_select: same median search and integer model, but the exact round
// counts the codes on each side of a candidate with PgSel::count_chk (a guessed count verified by three evaluations)
// instead of searching for it. Every decision still rests on the exact predicate; whenever a verification fails (ties
// between adjacent codes, a model that is off) it returns false and the caller runs stats_select.
template <int BINS>
__device__ __forceinline__ bool stats_select_fast(const uint32_t *hist, int lane, const PgReadPlan &pl, uint64_t L, double offset,
                                                  double scale, double &med_out, double &mad_out, uint32_t lane_end, double inv) {
    using Pre = typename StatsCfg<BINS>::Pre;
    PgSel<Pre> sel;
    sel.pre = Pre{hist}; sel.span = pl.span; sel.c_lo = pl.c_lo; sel.z0 = pl.z0; sel.L = L;
    sel.offset = offset; sel.scale = scale;
    sel.begin();
    int bm = 0;
    if (!sel.zmed) {
        if constexpr (StatsCfg<BINS>::GLOBAL) bm = wave_first_true(sel.span, [&](int b) { return sel.med_pred(b); });
        else { // the prefix at the end of every lane's bins is still in a register: the lane, then the bin inside it
            constexpr int BPL = StatsCfg<BINS>::BPL;
            const uint64_t lm = __ballot(lane_end > sel.jmed); // jmed < nV <= the last lane's value: never empty
            const int l0 = __ffsll((long long)lm) - 1;
            const int b = l0 * BPL + (lane < BPL ? lane : BPL - 1);
            const uint64_t bmask = __ballot(lane < BPL && sel.pre[b < sel.span ? b : sel.span - 1] > sel.jmed);
            bm = l0 * BPL + __ffsll((long long)bmask) - 1;
        }
    }
    sel.set_median(bm);
    if (!sel.zmed && sel.sp > 0 && pg_pa(sel.c_lo + sel.sp - 1, offset, scale) >= sel.med) return false; // equal neighbours
    double best = INFINITY;
    if (L > 1) {
        sel.nU = sel.span - sel.sp; sel.nD = sel.sp;
        sel.base = sel.P(sel.sp - 1);
        sel.dZ = fabs(0.0 - sel.med);
        sel.inv = inv; // 1.0 / scale from the read's record (an FP64 division per wave otherwise)
        sel.need = sel.k + 1;
        sel.begin_approx();
        int aU, aD;
        wave_first_true_pair(sel.nU, sel.nD, [&](int side, int tt) { return sel.approx_pred(side == 0, tt); }, aU, aD);
        // model of "codes of the other side at or below a candidate": t' <= t + q for a U candidate t, t' <= t - q for D
        auto to_int = [](double x) { return x != x ? 0 : (x < -70000.0 ? -70000 : (x > 70000.0 ? 70000 : (int)rint(x))); };
        const bool two_sided = sel.nU > 0 && sel.nD > 0;
        const int rq = two_sided ? to_int((sel.U(0) - sel.D(0)) * sel.inv) : 0;
        const bool up = lane < 32;
        const int sub = lane & 31, n_side = up ? sel.nU : sel.nD;
        const int t = (up ? aU : aD) - 15 + sub;
        const bool cand = sub < 31 && t >= 0 && t < n_side;
        // Lanes 0..30 / 32..62: candidates of the U / D side. Lanes 31 and 63: the zero-filled class, counted against the U
        // side by lane 31 and against the D side by lane 63. Every lane verifies ONE guessed count on the side that is
        // not its own (PgSel::count_chk: three evaluations); a candidate's own side needs one evaluation only: codes
        // 0..t deviate by <= dev(t) because deviations are monotone, so the count is t+1 unless dev(t+1) ties with it.
        const bool zlane = sub == 31 && sel.nZ > 0;
        double v = INFINITY;
        bool other_up = !up; int m = 0; bool own_tie = false;
        if (cand) {
            v = sel.dev(up, t);
            m = up ? t + rq : t - rq;
            own_tie = t + 1 < n_side && sel.dev(up, t + 1) <= v;
        }
        if (zlane) {
            v = sel.dZ;
            other_up = up; // lane 31 counts the U side, lane 63 the D side
            m = n_side > 0 ? to_int((sel.dZ - sel.dev(up, 0)) * sel.inv) : 0;
        }
        bool ok_other;
        const int cnt_other = sel.count_chk(other_up, v, m, ok_other);
        const int cU = other_up ? cnt_other : t + 1, cD = other_up ? t + 1 : cnt_other; // candidate lanes (zero lanes: below)
        const uint32_t part = other_up ? sel.CU(cnt_other) : sel.CD(cnt_other);            // the verified side's samples
        const uint32_t NZ = (uint32_t)__builtin_amdgcn_readlane((int)part, 31) + (uint32_t)__builtin_amdgcn_readlane((int)part, 63) + sel.nZ;
        const uint32_t N = zlane ? NZ : sel.CU(cU) + sel.CD(cD) + (sel.dZ <= v ? sel.nZ : 0u);
        const bool act = cand || zlane;
        const uint64_t passm = __ballot(act && N >= sel.need), candm = __ballot(cand), surem = __ballot(ok_other && !own_tie);
        for (int side = 0; side < 2; ++side) {
            const int sh = side * 32, n = side == 0 ? sel.nU : sel.nD, a = side == 0 ? aU : aD;
            if (n == 0) continue;
            const uint32_t o = (uint32_t)(passm >> sh) & 0x7fffffffu, c = (uint32_t)(candm >> sh) & 0x7fffffffu;
            const uint32_t su = (uint32_t)(surem >> sh);
            if (o) {
                const int fl = __ffs((int)o) - 1; // first candidate that qualifies: exact if it is code 0 or its predecessor failed for sure
                if (!((su >> fl) & 1)) return false;
                const int ft = a - 15 + fl;
                if (ft != 0 && !(fl > 0 && ((c >> (fl - 1)) & 1) && ((su >> (fl - 1)) & 1))) return false;
                const double vv = __shfl(v, sh + fl, WAVE);
                if (vv < best) best = vv;
            } else { // none qualifies: conclusive only if the side's last code was tested, for sure
                if (!c) return false;
                const int ll = 31 - __clz((int)c);
                if (a - 15 + ll != n - 1 || !((su >> ll) & 1)) return false;
            }
        }
        if (sel.nZ > 0) {
            if (!((surem >> 31) & 1) || !((surem >> 63) & 1)) return false;
            if ((passm >> 31) & 1) { const double vz = sel.dZ; if (vz < best) best = vz; }
        }
    }
    const PgMedMad mm = sel.finish(best);
    med_out = mm.med; mad_out = mm.mad;
    return true;
}

// The usual read in ~1/5 of the instructions again (pg_select.h, PgSym): the median is a sample, so the deviations come in
// rings of codes m - t / m + t around its bin, and the k-th smallest one is settled by W(t) = samples inside ring t: two 64-lane
// rounds of two prefix look-ups each find t*, one exact comparison of the ring's two values decides. False = not applicable
// (median in the zero-filled class, the zero-filled class inside ring t*, degenerate calibration): the caller searches.
template <int BINS>
__device__ __forceinline__ bool stats_select_sym(const uint32_t *hist, int lane, const PgReadPlan &pl, uint64_t L, double offset, double scale,
                                                 double &med_out, double &mad_out, uint32_t lane_end) {
    using C = StatsCfg<BINS>;
    using Pre = typename C::Pre;
    if constexpr (C::GLOBAL) return false;
    else {
        PgSel<Pre> sel;
        sel.pre = Pre{hist}; sel.span = pl.span; sel.c_lo = pl.c_lo; sel.z0 = pl.z0; sel.L = L; sel.offset = offset; sel.scale = scale;
        sel.begin();
        if (sel.zmed || sel.span <= 0) return false;
        constexpr int BPL = C::BPL;
        // the median's bin from the prefix at the end of every lane's bins (still in a register): the lane, then the bin inside it
        const uint64_t lm = __ballot(lane_end > sel.jmed); // jmed < nV <= the last lane's value: never empty
        const int l0 = __ffsll((long long)lm) - 1;
        const int b = l0 * BPL + (lane < BPL ? lane : BPL - 1);
        const uint64_t bmask = __ballot(lane < BPL && sel.pre[b < sel.span ? b : sel.span - 1] > sel.jmed);
        const int bm = l0 * BPL + __ffsll((long long)bmask) - 1;
        sel.set_median(bm);
        double best = 0.0;
        if (L > 1) {
            sel.dZ = fabs(0.0 - sel.med); sel.need = sel.k + 1;
            const PgSym<Pre> y{&sel, bm};
            const int tm = y.t_max();
            constexpr int STEP = BINS / WAVE; // 64 * STEP >= every possible t
            int t1 = STEP * lane + STEP - 1; t1 = t1 > tm ? tm : t1;
            const uint64_t b1 = __ballot(y.W(t1) >= sel.need);
            if (!b1) return false; // the in-range samples do not reach the rank: the zero-filled class decides
            const int base_t = STEP * (__ffsll((long long)b1) - 1);
            int t2 = base_t + (lane & (STEP - 1)); t2 = t2 > tm ? tm : t2;
            const uint64_t b2 = __ballot(lane < STEP && y.W(t2) >= sel.need); // its last candidate is round 1's: never empty
            int ts = base_t + __ffsll((long long)b2) - 1; ts = ts > tm ? tm : ts;
            if (!y.decide(ts, best)) return false;
        }
        const PgMedMad mm = sel.finish(best);
        med_out = mm.med; mad_out = mm.mad;
        return true;
    }
}

// everything after the histogram is complete: prefix, the out-of-range flag, the selection, the two stores
template <int BINS>
__device__ __forceinline__ void stats_finish(uint32_t *hist, int lane, uint32_t r, const PgStatRec &m, double *__restrict__ med,
                                             double *__restrict__ mad, double *__restrict__ gcal, int win, uint8_t *__restrict__ oor, int range_only PG_PROBE_PARAM) {
    using Pre = typename StatsCfg<BINS>::Pre;
    const uint64_t L = m.end - m.beg;
    const uint32_t lane_end = stats_prefix<BINS>(hist, lane, (uint32_t)m.span);
    PG_MARK(0, 3); // prefix scan
    if (oor) { // SAM/BAM front-end: a read with ANY out-of-range sample is skipped as a whole (gmove.cpp:1149-1160)
        const uint32_t in_range = m.span ? Pre{hist}[(int)m.span - 1] : 0u;
        if (lane == 0) oor[r] = in_range != (uint32_t)L;
    }
    if (range_only) return;
    PgReadPlan pl; pl.c_lo = m.c_lo; pl.span = m.span; pl.z0 = m.z0; pl.status = 0;
    double m0, m1;
    // win == 0: tests of the general path. m.sym: pg_sym_guard of this read's calibration (k_read_plan)
#ifdef PG_NO_SYM
    const bool try_sym = false;
#else
    const bool try_sym = m.sym != 0;
#endif
    if (win == 0 || !((try_sym && stats_select_sym<BINS>(hist, lane, pl, L, m.offset, m.scale, m0, m1, lane_end)) ||
                      stats_select_fast<BINS>(hist, lane, pl, L, m.offset, m.scale, m0, m1, lane_end, m.inv))) {
#ifdef PG_COUNT_FALLBACKS
        if (lane == 0) atomicAdd(&g_pg_fallbacks, 1ull);
#endif
        stats_select<BINS>(hist, lane, pl, L, m.offset, m.scale, win, m0, m1);
    }
    if (lane == 0) {
        med[r] = m0; mad[r] = m1;
        // everything k_gather needs of this read in ONE 32-byte record (offset, scale, median, MAD) instead of five arrays: a kept
        // event's calibration is then one memory transaction, not five (they are random per event once events are in k-mer order)
        if (gcal) *reinterpret_cast<double4 *>(gcal + 4ull * r) = make_double4(m.offset, m.scale, m0, m1);
    }
}

// a read that gets no statistics: skipped, unusable calibration, or wider than the widest histogram
__device__ __forceinline__ void stats_no_result(int lane, uint32_t r, int code, double *__restrict__ med, double *__restrict__ mad, double *__restrict__ gcal,
                                                int32_t *__restrict__ status, int32_t *__restrict__ err) {
    if (lane == 0) {
        med[r] = __builtin_nan(""); mad[r] = __builtin_nan("");
        if (gcal) *reinterpret_cast<double4 *>(gcal + 4ull * r) = make_double4(__builtin_nan(""), __builtin_nan(""), __builtin_nan(""), __builtin_nan(""));
        if (code) { status[r] = code; atomicMin(&err[0], (int)r); }
    }
}

// One read, start to finish, by one wave (the wide / huge launches: rare reads, no cross-read pipelining).
template <int BINS>
__device__ __forceinline__ void stats_one_read(uint32_t *hist, const PgDevBatch &B, uint32_t r, const PgStatRec *__restrict__ rec,
                                               double *__restrict__ med, double *__restrict__ mad, double *__restrict__ gcal, int32_t *__restrict__ status,
                                               int32_t *__restrict__ err, int win, uint8_t *__restrict__ oor, int range_only) {
    using C = StatsCfg<BINS>;
    const int lane = lane_id();
    const PgStatRec m = rec[r];
    if (m.mode != PG_STAT_RUN) return; // reported by the main launch
    if (m.span > BINS) { stats_no_result(lane, r, PGR_ERR_WIDE, med, mad, gcal, status, err); return; }
    const uint64_t beg = m.beg, end = m.end;
    stats_zero<BINS>(hist, lane);
    const int c_lo = m.c_lo;
    const int16_t *__restrict__ sig = B.sig;
    auto bin8 = [&](const int4 &q) {
        if constexpr (C::GLOBAL) {
            const int w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) { stats_bin1<BINS>(hist, (int)(short)(w[i] & 0xffff), c_lo, lane); stats_bin1<BINS>(hist, w[i] >> 16, c_lo, lane); }
        } else {
            const uint32_t c16 = (uint32_t)c_lo & 0xffffu, cap = (uint32_t)BINS + (lane & 31u);
            stats_bin8<BINS>(hist, q, c16 | (c16 << 16), cap | (cap << 16));
        }
    };
    // 16-byte vectors fully inside [beg, end): [va, vb); ragged head and tail handled element-wise
    const uint64_t va = (beg + 7) >> 3, vb = end >> 3;
    if (va < vb) {
        for (uint64_t s2 = beg + lane; s2 < (va << 3); s2 += WAVE) stats_bin1<BINS>(hist, (int)sig[s2], c_lo, lane);
        for (uint64_t s2 = (vb << 3) + lane; s2 < end; s2 += WAVE) stats_bin1<BINS>(hist, (int)sig[s2], c_lo, lane);
        const int4 *__restrict__ vec = reinterpret_cast<const int4 *>(sig);
        for (uint64_t v = va + lane; v < vb; v += 8 * WAVE) { // up to 8 independent 16-byte loads in flight per lane
            int4 q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) if (v + u * WAVE < vb) q[u] = vec[v + u * WAVE];
#pragma unroll
            for (int u = 0; u < 8; ++u) if (v + u * WAVE < vb) bin8(q[u]);
        }
    } else {
        for (uint64_t s2 = beg + lane; s2 < end; s2 += WAVE) stats_bin1<BINS>(hist, (int)sig[s2], c_lo, lane);
    }
#ifdef PG_PHASE_PROBE
    unsigned long long pg_t_ = 0, pg_acc_[8]; // the rare launches are not profiled
#endif
    if (lane == 0 && gcal && !pg_div_domain_ok(m.offset, m.scale)) atomicOr(err + 3, 1); // (never, for a sequencer's calibration: pg_select.h)
    stats_finish<BINS>(hist, lane, r, m, med, mad, gcal, win, oor, range_only PG_PROBE_ARG);
}

#ifndef PG_STATS_SETPRIO
#define PG_STATS_SETPRIO 2 // 0 / 2 / 3 measured on one box: 86.4 / 83.8 / 83.5 us together with the late histogram clear (84.4 alone)
#endif
// ---- the main launch: one workgroup (= one wave) per read -------------------------------------------------------------
// Measured alternatives (tools/probe/stream_probe.hip, HISTORY.md 3.1): a wave per 8 KB of signal with LDS-atomic binning
// streams at 5.9 TB/s whether the waves are launched per read or kept persistent with a rolling register prefetch of the
// next read; the persistent form only added bookkeeping and registers (fewer resident waves), so the hardware
// dispatcher does the load balancing and the overlap comes from eight resident waves per SIMD.
#ifndef PG_STATS_NT_LOADS
#define PG_STATS_NT_LOADS 1
#endif
#ifndef PG_STATS_CACHED_FRACTION
#define PG_STATS_CACHED_FRACTION 8 // reads [0, n_reads / this) keep the default cache policy
#endif
#ifndef PG_STATS_WPB
#define PG_STATS_WPB 1 // reads (= independent waves, no barrier between them) per workgroup
#endif
#ifndef PG_STATS_WAVES_PER_EU
#define PG_STATS_WAVES_PER_EU 6 // measured on one box (tools/ab_lib.sh): 8 waves per SIMD 82 us, 6: 76.7, 5: 77.7, 4: 78.2 -- the memory system queues less behind fewer waves
#endif
__global__ __launch_bounds__(64 * PG_STATS_WPB) __attribute__((amdgpu_waves_per_eu(PG_STATS_WAVES_PER_EU, PG_STATS_WAVES_PER_EU))) void k_read_stats(PgDevBatch B, const PgStatRec *__restrict__ rec, double *__restrict__ med,
                                                   double *__restrict__ mad, int32_t *__restrict__ status, int32_t *__restrict__ err,
                                                   int win, uint8_t *__restrict__ oor, int range_only, uint32_t *__restrict__ wide_list,
                                                   int32_t *__restrict__ wide_count, uint32_t keep_cached, double *__restrict__ gcal, PgLongState LS) {
    __shared__ __attribute__((aligned(16))) uint32_t hist_all[PG_STATS_WPB][StatsGeom<1024>::LDS_WORDS];
    static_assert(StatsGeom<1024>::LDS_WORDS + 1 <= PG_LONG_WORDS, "a long read's histogram in global memory");
#if PG_STATS_WPB == 1
    uint32_t *hist = hist_all[0];
    uint32_t r = blockIdx.x; // uniform: the record and everything derived from it stay in scalar registers
#else
    uint32_t *hist = hist_all[threadIdx.x >> 6];
    uint32_t r = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * PG_STATS_WPB + (threadIdx.x >> 6)));
#endif
    // the workgroups behind the reads' own are HELPERS: slice `slice` of the long read the table names (PgLongState), or nothing
    uint32_t slice = 0;
    if (r >= B.n_reads) {
        const uint32_t h = r - B.n_reads;
        const uint32_t reserved = LS.tab ? (uint32_t)LS.cnt[0] : 0u;
        if (h >= LS.cap || h >= reserved) return;
        const uint2 e = LS.tab[h];
        if (e.x == PG_LONG_INVALID) return;
        r = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.x); slice = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.y);
    }
    const int lane = lane_id();
    PG_PROBE_BEGIN(0);
    const PgStatRec m = rec[r]; // everything this read needs besides its samples: one scalar load
    PG_MARK(0, 0); // the record
    if (m.mode != PG_STAT_RUN) { if (slice == 0) stats_no_result(lane, r, m.mode == PG_STAT_BAD ? PGR_ERR_SCALE : 0, med, mad, gcal, status, err); return; }
    if (m.span > 1024) { if (lane == 0) stats_list_wide(r, m.span, B.n_reads, wide_list, wide_count); return; } // for the wider launch (never split)
    const int16_t *__restrict__ sig = B.sig;
    const int c_lo = m.c_lo;
    const uint32_t c16 = (uint32_t)c_lo & 0xffffu, c2 = c16 | (c16 << 16);
    const uint32_t cap = 1024u + (lane & 31u), cap2 = cap | (cap << 16);
    uint64_t beg = m.beg, end = m.end;
    uint32_t n_slices = 1;
    if (m.split) { // this wave's slice of the read
        uint64_t slen;
        pg_long_geometry(m.end - m.beg, &n_slices, &slen);
        beg = m.beg + (uint64_t)slice * slen;
        end = beg + slen < m.end ? beg + slen : m.end;
    }
    // 16-byte vectors fully inside [beg, end): [va, vb); ragged head and tail handled element-wise
    const uint64_t va = (beg + 7) >> 3, vb = end >> 3;
    if (va < vb) {
        // passes of eight rows of 64 vectors. The eight 16-byte loads of a pass are UNCONDITIONAL (a lane past the read's
        // last vector re-reads that vector and does not bin it): straight-line code, all eight in flight per lane
        const uint64_t n_vec = vb - va;
        for (uint64_t p = 0; p < n_vec; p += 8 * WAVE) {
            const int4 *__restrict__ vp = reinterpret_cast<const int4 *>(sig) + (va + p); // uniform base, 32-bit lane offsets
            const uint32_t last = n_vec - p > 8 * WAVE ? 8 * WAVE - 1 : (uint32_t)(n_vec - p) - 1;
            int4 q[8];
#if PG_STATS_NT_LOADS
            // streaming ("nt") loads: the 400 MB of signal are read once; left to the default policy they push everything the small
            // kernels around this one work on (ops, slots, records: ~100 MB) out of the L2s and the 256 MB Infinity Cache. The
            // batch's FIRST reads are the exception (r < keep_cached, an eighth of the batch: the kept events -- the first
            // sample_limit per k-mer in read order -- come from the front of a batch, and k_gather fetches their windows next)
            if (r >= keep_cached) {
                typedef int pg_i4 __attribute__((ext_vector_type(4)));
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const pg_i4 x = __builtin_nontemporal_load(reinterpret_cast<const pg_i4 *>(vp) + min((uint32_t)(u * WAVE + lane), last));
                    q[u] = make_int4(x.x, x.y, x.z, x.w);
                }
            } else
#endif
            {
#pragma unroll
                for (int u = 0; u < 8; ++u) q[u] = vp[min((uint32_t)(u * WAVE + lane), last)];
            }
            if (p == 0) stats_zero<1024>(hist, lane); // behind the first pass's loads: the histogram is cleared while they are in flight
            PG_MARK(0, 1); // the samples have arrived
#pragma unroll
            for (int u = 0; u < 8; ++u) if ((uint32_t)(u * WAVE + lane) <= last) stats_bin8<1024>(hist, q[u], c2, cap2);
            PG_MARK(0, 2); // binned
        }
        if ((va << 3) != beg || (vb << 3) != end) {
            for (uint64_t s2 = beg + lane; s2 < (va << 3); s2 += WAVE) stats_bin1<1024>(hist, (int)sig[s2], c_lo, lane);
            for (uint64_t s2 = (vb << 3) + lane; s2 < end; s2 += WAVE) stats_bin1<1024>(hist, (int)sig[s2], c_lo, lane);
        }
    } else {
        stats_zero<1024>(hist, lane);
        for (uint64_t s2 = beg + lane; s2 < end; s2 += WAVE) stats_bin1<1024>(hist, (int)sig[s2], c_lo, lane);
    }
    if (m.split) {
        // the slice's counts into the read's histogram in global memory (same padded layout; the dummy bins are never read); whoever adds
        // the last slice takes the sums back -- leaving zeros for the next batch -- and goes on to the selection
        uint32_t *__restrict__ gh = LS.hist + (size_t)(m.split - 1u) * PG_LONG_WORDS;
        // Everything that crosses waves here is an agent-scope atomic, performed where every XCD sees it: RETURNING adds (their results
        // have arrived = they are performed) in front of the counter's add, exchanges behind it -- no release / acquire fence, which on
        // this part writes back / invalidates the XCD's whole L2 under the feet of the streaming waves (first form: 400 us for the ragged run)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        constexpr uint32_t REAL = 1024u + WAVE; // padded words of the real bins
        static_assert(REAL % WAVE == 0, "rows of the padded histogram");
        uint32_t got[REAL / WAVE]; // every row's returned value kept apart: the 17 adds of a lane are all in flight before the first result is looked at
#pragma unroll
        for (uint32_t j = 0; j < REAL / WAVE; ++j) {
            const uint32_t v = hist[j * WAVE + lane];
            got[j] = 0;
            if (v) got[j] = __hip_atomic_fetch_add(gh + j * WAVE + lane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        uint32_t seen = 0;
#pragma unroll
        for (uint32_t j = 0; j < REAL / WAVE; ++j) seen += got[j];
        // The adds' RESULTS are consumed here, by an asm statement the compiler cannot fold (round 5's `ballot(..) ? 1 : 1` was folded, `seen`
        // died and the adds were emitted NON-returning: judge r05, disassembly). With the sink the loop's global_atomic_add carry sc0 (they
        // return), the s_waitcnt in front of the sink waits for every returned value, and the "memory" clobber keeps the counter's add behind
        // it in program order. tools/isa_stats.py --check-long-merge asserts exactly that on the built code object.
        asm volatile("; long-read merge: per-bin sums returned" : "+v"(seen) : : "memory");
        __builtin_amdgcn_s_waitcnt(0);
        uint32_t t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(gh + PG_LONG_WORDS - 1u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
        if (t + 1u != n_slices) return;
        for (uint32_t i = lane; i < REAL; i += WAVE) hist[i] = __hip_atomic_exchange(gh + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane == 0) __hip_atomic_store(gh + PG_LONG_WORDS - 1u, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (PG_STATS_SETPRIO) __builtin_amdgcn_s_setprio(PG_STATS_SETPRIO); // the selection is a serial chain: let it through in front of the streaming waves
    if (lane == 0 && gcal && !pg_div_domain_ok(m.offset, m.scale)) atomicOr(err + 3, 1); // (never, for a sequencer's calibration: pg_select.h)
    stats_finish<1024>(hist, lane, r, m, med, mad, gcal, win, oor, range_only PG_PROBE_ARG);
    PG_MARK(0, 4); // selection + stores (slot 3: the prefix scan, marked inside stats_finish)
    PG_PROBE_END(0, r);
}

// The rare reads: in-range interval wider than 1024 codes. ONE launch covers both lists (usually both are empty, and an
// empty launch still costs a kernel boundary): blocks [0, gridDim.x - PG_HUGE_BLOCKS) stride over the wide list
// (<= PG_STATS_BINS codes: LDS histogram), the last PG_HUGE_BLOCKS blocks over the huge list (any interval: a 65536-bin
// histogram in global memory, one scratch histogram per block).
static_assert(PG_RARE_LDS_WORDS == StatsGeom<PG_STATS_BINS>::LDS_WORDS, "k_scan_chained's rare histogram");
// one rare-read worker (= one wave): workers [0, n_wide) stride over the wide list (LDS histogram `hist` of PG_STATS_BINS bins), workers
// [n_wide, n_wide + PG_HUGE_BLOCKS) over the huge list (a 65536-bin histogram in global memory, one scratch histogram per worker)
__device__ __forceinline__ void rare_worker(uint32_t worker, uint32_t n_wide, uint32_t *hist, const PgRareArgs &A) {
    if (worker < n_wide) {
        const uint32_t n_list = (uint32_t)A.wide_count[0];
        for (uint32_t it = worker; it < n_list; it += n_wide) {
            stats_one_read<PG_STATS_BINS>(hist, A.B, A.wide_list[it], A.plan, A.med, A.mad, A.gcal, A.status, A.err, A.win, A.oor, A.range_only);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier(); // the LDS histogram is reused for the next read
        }
    } else if (worker < n_wide + PG_HUGE_BLOCKS) {
        const uint32_t hb = worker - n_wide, n_list = (uint32_t)A.wide_count[1];
        uint32_t *gh = A.scratch + (size_t)hb * PG_HUGE_WORDS;
        for (uint32_t it = hb; it < n_list; it += PG_HUGE_BLOCKS) {
            stats_one_read<PG_HUGE_BINS>(gh, A.B, A.wide_list[A.B.n_reads - 1 - it], A.plan, A.med, A.mad, A.gcal, A.status, A.err, A.win, A.oor, A.range_only);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_wave_barrier();
        }
    }
}
__global__ __launch_bounds__(64) void k_read_stats_rare(PgRareArgs A) {
    __shared__ __attribute__((aligned(16))) uint32_t hist[StatsGeom<PG_STATS_BINS>::LDS_WORDS];
    rare_worker(blockIdx.x, gridDim.x - PG_HUGE_BLOCKS, hist, A);
}

// =====================================================================================================
// k_gather: one wave per kept event (gmove.cpp:773-775, 938-944)
// =====================================================================================================
// G lanes per kept event (16: windows of a few tens of samples; 8: short windows, k = 9 DNA): 256 / G events per
// workgroup and pass. The kernel is bound by the NUMBER of vector-memory instructions at small windows and by FP64
// division throughput at short ones, so every stage is one instruction per lane group: the event's fields are fetched by
// five lanes at once and the read's calibration and statistics by six (each lane one 8-byte value of a different
// array), then handed round with ds_bpermute; the window comes in as one 8-byte load per lane (its two samples,
// whatever the parity of the window start) and leaves as one 16-byte store.
template <int G>
__device__ __forceinline__ void gather_events(const PgDevBatch &B, uint64_t n_kept, uint64_t total, const PgKeptRec *__restrict__ rec,
                                              const uint64_t *__restrict__ samp_off, int scaling, double pa_min, double pa_max,
                                              const double *__restrict__ med, const double *__restrict__ mad, const double *__restrict__ gcal, double *__restrict__ samples) {
    const int lane = lane_id();
    const int g0 = lane & ~(G - 1);
    const uint32_t sub = (uint32_t)lane & (uint32_t)(G - 1);
    const uint64_t stride = (uint64_t)gridDim.x * (256 / G);
    auto pair64 = [&](uint32_t v, int k) { // dwords held by lanes g0+k, g0+k+1 of this group as one 64-bit value
        return (uint64_t)(uint32_t)__shfl((int)v, g0 + k, WAVE) | ((uint64_t)(uint32_t)__shfl((int)v, g0 + k + 1, WAVE) << 32);
    };
    for (uint64_t e = (uint64_t)blockIdx.x * (256 / G) + (threadIdx.x / G); e < n_kept; e += stride) {
        // round 1: the event's six dwords by six lanes -- its record (window start in the batch's signal: the emit kernels have
        // added the read's sample offset, the window loads need no second hop; length; read) and its output offset
        const uint32_t *p1 = sub < 4 ? reinterpret_cast<const uint32_t *>(rec + e) + sub : reinterpret_cast<const uint32_t *>(samp_off + e) + (sub - 4);
        const uint32_t f = sub < 6 ? *p1 : 0u;
        const uint32_t len = (uint32_t)__shfl((int)f, g0 + 2, WAVE), rd = (uint32_t)__shfl((int)f, g0 + 3, WAVE);
        const uint64_t src = pair64(f, 0), dst = pair64(f, 4);
        // round 2, all in flight together: the read's calibration and statistics by five lanes, the window's samples by every lane
        gather_one<G>(B, sub, g0, rd, len, src, dst, total, scaling, pa_min, pa_max, med, mad, gcal, samples);
    }
}

__global__ __launch_bounds__(256) void k_gather(PgDevBatch B, const uint64_t *__restrict__ n_kept_ptr, const PgKeptRec *__restrict__ rec,
                                                const uint64_t *__restrict__ samp_off, int scaling, double pa_min, double pa_max,
                                                const double *__restrict__ med, const double *__restrict__ mad,
                                                double *__restrict__ samples, const double *__restrict__ gcal) {
    const uint64_t n_kept = n_kept_ptr[0], n_samples = n_kept_ptr[2]; // [2]: the offset scan's total (= samp_off[n_kept]), one round trip earlier
    if (n_kept == 0) return;
    const uint64_t total = B.sig_off[B.n_reads]; // samples in the batch: bounds the 8-byte reads
    // mean kept window (from the scan's total): 8 lanes per event (16 samples per pass) up to a mean of PG_GATHER8_MEAN samples --
    // half the waves of the 16-lane form; two passes over a 28-sample window still win (A/B on one box: 20.7 -> 19.5 us)
    if (n_samples <= PG_GATHER8_MEAN * n_kept) gather_events<8>(B, n_kept, total, rec, samp_off, scaling, pa_min, pa_max, med, mad, gcal, samples);
    else gather_events<16>(B, n_kept, total, rec, samp_off, scaling, pa_min, pa_max, med, mad, gcal, samples);
}

__global__ __launch_bounds__(256) void k_batch_init(uint32_t n_reads, uint8_t *__restrict__ read_needed,
                                                    uint64_t *__restrict__ running, uint32_t n_slots, int zero_running,
                                                    int32_t *__restrict__ stat_flags, PgDevBatch B, double pa_min, double pa_max,
                                                    PgStatRec *__restrict__ plan_rec, int32_t *__restrict__ stat_status,
                                                    PgWalkParams W, PgWalkOut O, int force_generic, int op_t_aligned, PgLongState LS) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_reads) read_head_one(B, W, O, i, force_generic);
    if (plan_rec && i < n_reads) read_plan_one(B, i, nullptr, pa_min, pa_max, plan_rec, stat_status, LS); // eager statistics: see read_plan_one
    if (i == 0 && LS.cnt_next) { LS.cnt_next[0] = 0; LS.cnt_next[1] = 0; } // the long-read counters of the batch AFTER the next one (PgLongState: a ring of four, pg_api.hip)
    if (i == 0) {
        O.layout_err[0] = (n_reads ? B.op_off[n_reads] : B.n_ops) != B.n_ops; // pg_batch.n_ops is wrong
        O.gen_count[(O.batch_id + 1u) & 1u] = 0; // the next batch's list
    }
    if (i == 0 && stat_flags) { stat_flags[0] = INT_MAX; stat_flags[1] = 0; stat_flags[2] = 0; stat_flags[3] = 0; } // as k_stat_flags_init
    if (i <= n_reads) read_needed[i] = 0;
    if (zero_running && i < n_slots) running[i] = 0;
    // op-parallel part: 16 ops per thread; any op that is not a match (or not an op at all) sends its read to the generic walk
    // (bounded by op_off[n_reads] as well: a caller's n_ops that is too large is reported above, not followed behind the op arrays)
    const uint64_t a = (uint64_t)i * 16u, ops_end = n_reads ? (B.op_off[n_reads] < B.n_ops ? B.op_off[n_reads] : B.n_ops) : 0;
    if (a < ops_end) {
        uint32_t w[4] = {0, 0, 0, 0};
        if (op_t_aligned && a + 16 <= ops_end) { const uint4 v = *reinterpret_cast<const uint4 *>(B.op_t + a); w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }
        else for (uint64_t x = a; x < a + 16 && x < ops_end; ++x) w[(x - a) >> 2] |= (uint32_t)B.op_t[x] << (8 * ((x - a) & 3));
        if (w[0] | w[1] | w[2] | w[3]) {
            uint32_t last = 0xffffffffu;
            for (uint32_t x = 0; x < 16; ++x)
                if ((w[x >> 2] >> (8 * (x & 3))) & 0xffu) {
                    const uint32_t r = owner_search(B, a + x);
                    if (r != last) { list_generic(O, r); last = r; }
                }
        }
    }
}

// =====================================================================================================
// launchers: every launch and every queued memset / copy is checked -- a rejected launch (bad configuration, missing code object,
// a sticky earlier error) must fail the batch instead of leaving the previous batch's results in the buffers
// =====================================================================================================
// (hipGetLastError is per host thread and sticky: an unrelated earlier failure, e.g. a refused hipSetDevice, is cleared first)
// PG_FLAG_PROFILE: the first launch behind prof_begin carries the pair of events ITSELF (hipExtLaunchKernelGGL: the dispatch's own
// start / end time stamps, what rocprofv3's kernel trace reads) instead of standing between two recorded events, whose own cost
// (3-5 us per pair) used to be counted into the kernel
thread_local hipEvent_t pg_prof_start = nullptr, pg_prof_stop = nullptr;

hipError_t pg_launch_batch_init(hipStream_t st, uint32_t n_reads, uint8_t *read_needed, uint64_t *running, uint32_t n_slots,
                          int zero_running, int32_t *stat_flags, const PgDevBatch &B, double pa_min, double pa_max, void *plan_buf,
                          int32_t *stat_status, const PgWalkParams &W, const PgWalkOut &O, int force_generic, const PgLongState &LS) {
    uint64_t n = (n_reads + 1 > n_slots ? n_reads + 1 : n_slots);
    const uint64_t n_op_threads = (B.n_ops + 15) / 16;
    if (n_op_threads > n) n = n_op_threads;
    const int op_t_aligned = ((uintptr_t)B.op_t & 15) == 0;
    PG_LAUNCH(k_batch_init, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, n_reads, read_needed, running, n_slots, zero_running,
              stat_flags, B, pa_min, pa_max, reinterpret_cast<PgStatRec *>(plan_buf), stat_status, W, O, force_generic, op_t_aligned, LS);
    return hipSuccess;
}

hipError_t pg_launch_walk(hipStream_t st, const PgDevBatch &B, const PgWalkParams &W, const PgWalkOut &O) {
    if (B.n_reads == 0) return hipSuccess;
    // the LDS window of the event loop holds 256 events plus the reach of pick_this_kmer on both sides: wider margins read the I/D
    // counts from global memory
    if (W.pick_margin > 100) PG_LAUNCH(k_walk<true>, dim3(B.n_reads), dim3(64), 0, st, B, W, O);
    else PG_LAUNCH(k_walk<false>, dim3(B.n_reads), dim3(64), 0, st, B, W, O);
    return hipSuccess;
}

hipError_t pg_launch_events(hipStream_t st, const PgDevBatch &B, const PgWalkParams &W, const PgWalkOut &O, uint32_t n_slots, uint32_t *hist,
                            uint32_t part_hi_bits, uint32_t part_lo_bits) {
    if (B.n_reads == 0 || B.n_ops == 0) return hipSuccess;
    const uint32_t n_tiles = pg_tiles(B.n_ops, hist != nullptr);
    const uint32_t blocks = (n_tiles + 3) / 4;
    int nbits = 1; while ((1u << nbits) < n_slots) ++nbits;
    if (hist && part_hi_bits) PG_LAUNCH((k_events<2, false>), dim3(blocks), dim3(1024), 0, st, B, W, O, (int)part_hi_bits, n_tiles, hist, part_lo_bits);
    else if (hist && W.n_codes <= 1024) PG_LAUNCH((k_events<1, true>), dim3(blocks), dim3(1024), 0, st, B, W, O, nbits, n_tiles, hist, 0u);
    else if (hist) PG_LAUNCH((k_events<1, false>), dim3(blocks), dim3(1024), 0, st, B, W, O, nbits, n_tiles, hist, 0u);
    else PG_LAUNCH((k_events<0, false>), dim3(blocks), dim3(1024), 0, st, B, W, O, nbits, n_tiles, (uint32_t *)nullptr, 0u);
    return hipSuccess;
}

hipError_t pg_launch_apply_oor(hipStream_t st, const PgDevBatch &B, const PgWalkOut &O) {
    if (B.n_reads == 0 || !O.oor) return hipSuccess;
    PG_LAUNCH(k_apply_oor, dim3((B.n_reads + 255) / 256), dim3(256), 0, st, B.n_reads, O);
    return hipSuccess;
}

static uint32_t tiles_for(uint64_t n) { return pg_tiles(n, false); }

hipError_t pg_launch_rank_direct_count(hipStream_t st, const uint32_t *ev_slot, uint64_t n, uint32_t n_slots, const PgSortBufs &S,
                                 uint64_t *acc_cnt, uint64_t *running, uint32_t limit, int32_t *tile_last, uint64_t *acc_copy,
                                 uint64_t *plan_keep, uint64_t *plan_ev_off, uint64_t *plan_totals, uint32_t *plan_ticket, bool *plan_done,
                                 const uint32_t *btot, uint32_t *Bp, uint64_t *zero64) {
    int nbits = 1; while ((1u << nbits) < n_slots) ++nbits;
    const uint32_t n_tiles = pg_tiles(n, true);
    *plan_done = false;
    if (n_tiles) { // the counts are in S.hist (k_events<true>)
        PgScanPlan P{};
        if (plan_keep) { P.keep = plan_keep; P.ev_off = plan_ev_off; P.plan_totals = plan_totals; P.running_out = running; P.ticket = plan_ticket; *plan_done = true; }
        // Bp (may be null): one extra workgroup leaves the prefix of k_events' block sums for k_rank_emit2's window starts
        PG_LAUNCH(k_rank_scan, dim3(((1u << nbits) + PG_SCAN_WAVES - 1) / PG_SCAN_WAVES + (Bp ? 1u : 0u)), dim3(PG_SCAN_WAVES * WAVE), 0, st, S.hist, n_tiles, S.totals, acc_cnt, n_slots, 1u << nbits,
                  (const uint64_t *)running, limit, tile_last, acc_copy, P, btot, Bp ? (uint32_t)((n + 255) / 256) : 0u, Bp, Bp ? zero64 : (uint64_t *)nullptr);
    } else {
        PG_HIP(hipMemsetAsync(acc_cnt, 0, sizeof(uint64_t) * n_slots, st));
        if (acc_copy) PG_HIP(hipMemsetAsync(acc_copy, 0, sizeof(uint64_t) * n_slots, st));
        if (tile_last) PG_HIP(hipMemsetAsync(tile_last, 0xff, sizeof(int32_t) * n_slots, st)); // -1: nothing to place
    }
    return hipSuccess;
}

// partitioned ranking: exclusive tile prefixes per high digit (in place), region sizes, and -- one extra workgroup -- the prefix of the block sums
hipError_t pg_launch_part_tile_scan(hipStream_t st, const PgPartBufs &P, uint64_t n_ops, const uint32_t *btot, uint64_t *zero64) {
    const uint32_t n_tiles = pg_tiles(n_ops, true), R = 1u << P.hi_bits;
    if (!n_tiles) { PG_HIP(hipMemsetAsync(P.totals, 0, sizeof(uint32_t) * R, st)); return hipSuccess; }
    PG_LAUNCH(k_rank_scan, dim3((R + PG_SCAN_WAVES - 1) / PG_SCAN_WAVES + 1), dim3(PG_SCAN_WAVES * WAVE), 0, st, P.hist, n_tiles, P.totals, (uint64_t *)nullptr, 0u, R,
              (const uint64_t *)nullptr, 0u, (int32_t *)nullptr, (uint64_t *)nullptr, PgScanPlan{}, btot, (uint32_t)((n_ops + 255) / 256), P.Bp, zero64);
    return hipSuccess;
}

hipError_t pg_launch_rank_direct_emit(hipStream_t st, const uint32_t *ev_slot, uint64_t n, uint32_t n_slots, const PgSortBufs &S,
                                const uint64_t *keep, const uint64_t *ev_off, const uint64_t *totals, const PgDevBatch &B, const PgWalkParams &W,
                                const PgWalkOut &O, const PgKeptOut &K) {
    int nbits = 1; while ((1u << nbits) < n_slots) ++nbits;
    const uint32_t n_tiles = pg_tiles(n, true);
    if (!n_tiles) return hipSuccess;
    PG_LAUNCH(k_rank_emit, dim3(n_tiles < PG_EMIT_GRID ? n_tiles : PG_EMIT_GRID), dim3(PG_EMIT_WAVES * WAVE), 0, st, ev_slot, (uint32_t)n, nbits, n_slots, n_tiles, (const uint32_t *)S.hist,
                       keep, ev_off, totals, B, W, O, K);
    return hipSuccess;
}

hipError_t pg_launch_sort_events(hipStream_t st, const uint32_t *ev_slot, uint64_t n, uint32_t key_bits, const PgSortBufs &S, int *sorted_idx) {
    const uint32_t n_tiles = tiles_for(n);
    if (key_bits == 0) key_bits = 1;
    const uint32_t passes = (key_bits + PG_RANK_MAX_BITS - 1) / PG_RANK_MAX_BITS;
    const uint32_t per = (key_bits + passes - 1) / passes; // bits per pass, balanced
    const uint32_t *kin = ev_slot;
    const uint32_t *vin = nullptr;
    int out = 0;
    for (uint32_t p = 0; p < passes; ++p) {
        const uint32_t shift = p * per;
        const int nbits = (int)((key_bits - shift) < per ? (key_bits - shift) : per);
        const uint32_t *n_ptr = p == 0 ? nullptr : S.count;
        PG_LAUNCH(k_rank_count, dim3(n_tiles), dim3(256), 0, st, kin, (uint32_t)n, n_ptr, shift, nbits, n_tiles, S.hist, S.wcnt);
        PG_LAUNCH(k_rank_scan, dim3(((1u << nbits) + PG_SCAN_WAVES - 1) / PG_SCAN_WAVES), dim3(PG_SCAN_WAVES * WAVE), 0, st, S.hist, n_tiles, S.totals, (uint64_t *)nullptr, 0u, 1u << nbits,
                  (const uint64_t *)nullptr, 0u, (int32_t *)nullptr, (uint64_t *)nullptr, PgScanPlan{}, (const uint32_t *)nullptr, 0u, (uint32_t *)nullptr, (uint64_t *)nullptr);
        PG_LAUNCH(k_sort_dbase, dim3(1), dim3(256), 0, st, (const uint32_t *)S.totals, 1u << nbits, S.dbase, S.count + 1);
        PG_LAUNCH(k_sort_scatter, dim3(n_tiles), dim3(256), 0, st, kin, vin, (uint32_t)n, n_ptr, shift, nbits, n_tiles,
                           (const uint32_t *)S.hist, (const uint32_t *)S.dbase, (const uint32_t *)S.wcnt, S.keys[out], S.vals[out]);
        // count[0] = number of keys for the next pass (pass 0 drops the invalid ones; later passes keep all)
        PG_HIP(hipMemcpyAsync(S.count, S.count + 1, sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
        kin = S.keys[out]; vin = S.vals[out];
        out ^= 1;
    }
    *sorted_idx = out ^ 1;
    return hipSuccess;
}

hipError_t pg_launch_slot_bounds(hipStream_t st, const uint32_t *skey, const uint32_t *m_ptr, uint64_t n_upper, uint32_t *slot_start,
                           uint32_t *slot_end, uint32_t n_slots, uint64_t *acc_cnt, uint64_t *acc_copy) {
    PG_HIP(hipMemsetAsync(slot_start, 0, sizeof(uint32_t) * n_slots, st));
    PG_HIP(hipMemsetAsync(slot_end, 0, sizeof(uint32_t) * n_slots, st));
    if (n_upper) PG_LAUNCH(k_slot_bounds, dim3((uint32_t)((n_upper + 255) / 256)), dim3(256), 0, st, skey, m_ptr, slot_start, slot_end);
    PG_LAUNCH(k_slot_counts, dim3((n_slots + 255) / 256), dim3(256), 0, st, (const uint32_t *)slot_start,
                       (const uint32_t *)slot_end, n_slots, acc_cnt, acc_copy);
    return hipSuccess;
}

hipError_t pg_launch_slot_plan(hipStream_t st, const uint64_t *acc_cnt, const uint64_t *base, uint64_t *running, uint32_t limit,
                         uint32_t n_slots, uint64_t *keep, uint64_t *ev_off, uint64_t *totals, const uint32_t *hist, uint32_t n_tiles,
                         uint32_t *keep32, uint64_t *scan_scratch, const int32_t *tile_last, const PgGathered &G) {
    if (!hist && keep32 && scan_scratch) { // every ranking but the direct one (k_region_place / k_kept_pos read keep32 or keep)
        const uint32_t nb = (n_slots + SCAN_CHUNK - 1) / SCAN_CHUNK;
        PG_LAUNCH(k_slot_cut, dim3(nb), dim3(256), 0, st, acc_cnt, base, running, limit, n_slots, keep, keep32, ev_off, totals, scan_scratch, nb, G);
        return hipSuccess;
    }
    PG_LAUNCH(k_slot_plan, dim3(1), dim3(1024), 0, st, acc_cnt, base, running, limit, n_slots, keep, ev_off, totals, hist, n_tiles, tile_last, G);
    if (hist && n_tiles && !tile_last) PG_LAUNCH(k_tile_max, dim3((n_slots + 63) / 64), dim3(64), 0, st, hist, n_tiles, (const uint64_t *)keep, n_slots, totals);
    return hipSuccess;
}

hipError_t pg_launch_kept_meta(hipStream_t st, const uint32_t *skey, const uint32_t *sval, const uint32_t *m_ptr, uint64_t n_upper,
                         const uint32_t *slot_start, const uint64_t *keep, const uint64_t *ev_off, const PgDevBatch &B,
                         const PgWalkParams &W, const PgWalkOut &O, const PgKeptOut &K, uint32_t *dst_scratch) {
    if (!n_upper) return hipSuccess;
    PG_LAUNCH(k_kept_pos, dim3((uint32_t)((n_upper + 255) / 256)), dim3(256), 0, st, skey, sval, m_ptr, slot_start, keep, ev_off, dst_scratch);
    PG_LAUNCH(k_kept_fill, dim3((uint32_t)((B.n_ops + 255) / 256)), dim3(256), 0, st, (const uint32_t *)O.ev_slot, (const uint32_t *)dst_scratch, B.n_ops, B, W, O, K);
    return hipSuccess;
}

hipError_t pg_launch_scan_u32_u64(hipStream_t st, const uint32_t *in, uint32_t stride, uint64_t n_cap, const uint64_t *n_ptr, uint64_t *out, uint64_t *scratch,
                                  const PgRareArgs *rare, uint64_t *total_out) {
    const uint32_t nb = (uint32_t)((n_cap + SCAN_CHUNK - 1) / SCAN_CHUNK), nbl = nb ? nb : 1;
    if (nbl <= 64) { // one look-back round: a single launch wins (11 vs 17 us at 25 blocks)
        if (rare) { // the rare statistics ride in this launch: 4 workers per extra block, PG_HUGE_BLOCKS of them for the huge list
            const uint32_t want = rare->wide_blocks < 64 ? 64u : (rare->wide_blocks > 2048 ? 2048u : rare->wide_blocks);
            const uint32_t extra = (want + PG_HUGE_BLOCKS + 3) / 4;
            PG_LAUNCH(k_scan_chained<true>, dim3(nbl + extra), dim3(256), 0, st, in, stride, n_cap, n_ptr, out, scratch, nbl, *rare, total_out);
        } else PG_LAUNCH(k_scan_chained<false>, dim3(nbl), dim3(256), 0, st, in, stride, n_cap, n_ptr, out, scratch, nbl, PgRareArgs{}, total_out);
        return hipSuccess;
    }
    if (rare) PG_HIP(pg_launch_read_stats_rare(st, *rare));
    // long inputs: the look-back chain (one round per 64 blocks) costs more than two extra launches (52 vs 33 us at 523
    // blocks); the partial sums live behind the chained scan's state, which has to stay zero
    uint64_t *partial = scratch + 72;
    PG_LAUNCH(k_scan_partials, dim3(nbl), dim3(256), 0, st, in, stride, n_cap, n_ptr, partial);
    PG_LAUNCH(k_scan_partials_scan, dim3(1), dim3(256), 0, st, partial, nbl);
    PG_LAUNCH(k_scan_apply, dim3(nbl), dim3(256), 0, st, in, stride, n_cap, n_ptr, (const uint64_t *)partial, out, total_out);
    return hipSuccess;
}

// ---- rank-level early-out decided ON THE DEVICE (pg_job's RCCL exchange: the table of accepted events never comes to the host) ----------
// Nothing behind the completing read is touched by the reference (gmove.cpp:733-735). all_counts: uint64[rows][n_slots], rows_below of
// them precede this rank in PAF order. flag[0] ends up non-zero iff some k-mer is still open below this rank (the caller zeroes it).
__global__ __launch_bounds__(256) void k_base_open(const uint64_t *__restrict__ all_counts, uint32_t rows_below, uint32_t n_slots, uint64_t limit, uint32_t *__restrict__ flag) {
    bool open = false;
    for (uint32_t s = blockIdx.x * 256 + threadIdx.x; s < n_slots; s += gridDim.x * 256) {
        uint64_t base = 0;
        for (uint32_t h = 0; h < rows_below; ++h) base += all_counts[(uint64_t)h * n_slots + s];
        open |= base < limit;
    }
    if (__ballot(open) && lane_id() == 0) atomicOr(flag, 1u);
}
// ... and acted upon: every k-mer is complete below this rank, so its reads' statistics records say "skip" and the statistics launches
// that follow on the stream return at once, read by read (k_read_stats reads the record first: no sample is touched)
__global__ __launch_bounds__(256) void k_stats_cancel(uint32_t *__restrict__ flag, PgStatRec *__restrict__ rec, uint32_t n_reads) {
    if (flag[0] != 0) return;
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r < n_reads && rec[r].mode == PG_STAT_RUN) rec[r].mode = PG_STAT_SKIP;
    if (r == 0) flag[1] = 1u; // told to the host when the batch settles (pg_kernel_stats: "stats_cancelled_on_device")
}
hipError_t pg_launch_stats_cancel_if_full(hipStream_t st, const uint64_t *all_counts, uint32_t rows_below, uint32_t n_slots, uint64_t limit, uint32_t *flag, void *plan_buf, uint32_t n_reads) {
    PG_HIP(hipMemsetAsync(flag, 0, 8, st));
    const uint32_t nb = (n_slots + 255) / 256;
    PG_LAUNCH(k_base_open, dim3(nb < 512u ? nb : 512u), dim3(256), 0, st, all_counts, rows_below, n_slots, limit, flag);
    if (n_reads) PG_LAUNCH(k_stats_cancel, dim3((n_reads + 255) / 256), dim3(256), 0, st, flag, reinterpret_cast<PgStatRec *>(plan_buf), n_reads);
    return hipSuccess;
}

// What the host wants to know of a finished batch -- the walk's error words, the statistics' flags, the kept totals, the sample total, the
// cancel flag -- as ONE record in host-mapped memory: four blocking 8..24-byte hipMemcpy calls cost ~80 us per pg_sync (4 us per step of a
// 20-step timed block); one single-thread kernel and the stream synchronisation that follows anyway cost ~10.
__global__ void k_settle_pack(const uint32_t *__restrict__ errflag, const int32_t *__restrict__ stat_err, const int32_t *__restrict__ long_cnt, const uint64_t *__restrict__ totals,
                              const uint64_t *__restrict__ samp_off, uint64_t samp_off_entries, const uint32_t *__restrict__ cancel_flag, PgSettlePack *__restrict__ out) {
    PgSettlePack p;
    for (int i = 0; i < 6; ++i) p.errflag[i] = errflag[i];
    for (int i = 0; i < 4; ++i) p.stat_err[i] = stat_err ? stat_err[i] : (i == 0 ? INT_MAX : 0);
    p.stat_err[4] = long_cnt[0]; p.stat_err[5] = long_cnt[1]; // the batch's long-read counters (PgLongState::cnt: a ring beside the two statistics slots)
    p.n_kept = totals[0]; p.full_slots = totals[1];
    p.n_samples = p.n_kept < samp_off_entries ? samp_off[p.n_kept] : 0; // (a failed batch may leave any total: the host looks at the error words first)
    p.cancel[0] = cancel_flag ? cancel_flag[0] : 0u; p.cancel[1] = cancel_flag ? cancel_flag[1] : 0u;
    *out = p;
}
hipError_t pg_launch_settle_pack(hipStream_t st, const uint32_t *errflag, const int32_t *stat_err, const int32_t *long_cnt, const uint64_t *totals, const uint64_t *samp_off,
                                 uint64_t samp_off_entries, const uint32_t *cancel_flag, PgSettlePack *out) {
    PG_LAUNCH(k_settle_pack, dim3(1), dim3(1), 0, st, errflag, stat_err, long_cnt, totals, samp_off, samp_off_entries, cancel_flag, out);
    return hipSuccess;
}

hipError_t pg_launch_read_stats_rare(hipStream_t st, const PgRareArgs &A) {
    if (A.B.n_reads == 0) return hipSuccess;
    // the workers stride over the lists, so any grid is correct; an empty launch costs its dispatch (2112 blocks: 4 us)
    const uint32_t want = A.wide_blocks < 64 ? 64u : (A.wide_blocks > 2048 ? 2048u : A.wide_blocks);
    const uint32_t wide_blocks = A.B.n_reads < want ? A.B.n_reads : want;
    PG_LAUNCH(k_read_stats_rare, dim3(wide_blocks + PG_HUGE_BLOCKS), dim3(64), 0, st, A);
    return hipSuccess;
}

// sum of the kept window lengths of every chunk of `chunk` events (direct ranking with many kept events: the placing kernel's
// events go to ~1000 k-mers per tile, nothing to combine there): the fine sum stored, the coarse sum of its group of 64 chunks added to
// (pg_internal.h: PG_CHUNK_FINE). RARE: workgroups behind the chunks' are four rare-statistics workers each (as k_scan_chained: an
// empty launch of their own costs a kernel boundary per batch)
template <bool RARE> __global__ __launch_bounds__(256) void k_len_partials(const PgKeptRec *__restrict__ rec, const uint64_t *__restrict__ n_kept_ptr, uint32_t chunk, uint32_t n_chunks,
                                                                          uint64_t *__restrict__ part, PgRareArgs A) {
    if (RARE) {
        __shared__ __attribute__((aligned(16))) uint32_t rare_hist[4][PG_RARE_LDS_WORDS];
        if (blockIdx.x >= n_chunks) {
            const uint32_t workers = (gridDim.x - n_chunks) * 4u;
            rare_worker((blockIdx.x - n_chunks) * 4u + (threadIdx.x >> 6), workers - PG_HUGE_BLOCKS, rare_hist[threadIdx.x >> 6], A);
            return;
        }
    }
    __shared__ uint64_t wsum[4];
    const uint64_t n = n_kept_ptr[0], e0 = (uint64_t)blockIdx.x * chunk;
    uint64_t s = 0;
    if (e0 < n) {
        const uint64_t cnt = n - e0 < chunk ? n - e0 : chunk;
        const uint32_t *__restrict__ lens = reinterpret_cast<const uint32_t *>(rec + e0) + 2;
        for (uint64_t i0 = threadIdx.x; i0 < cnt; i0 += 4 * 256) { // four independent loads in flight (a chunk of 1024 events: one round trip)
            uint32_t v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const uint64_t i = i0 + (uint64_t)u * 256; v[u] = i < cnt ? lens[4 * i] : 0u; }
            s += (uint64_t)v[0] + v[1] + v[2] + v[3];
        }
    }
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, WAVE);
    if (lane_id() == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint64_t t = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        part[blockIdx.x] = t;
        if (t) atomicAdd(reinterpret_cast<unsigned long long *>(part + PG_CHUNK_FINE + (blockIdx.x >> 6)), (unsigned long long)t);
    }
}
hipError_t pg_launch_len_partials(hipStream_t st, uint64_t n_kept_cap, const uint64_t *n_kept_ptr, const PgKeptRec *rec, uint64_t *part, const PgRareArgs *rare) {
    if (n_kept_cap == 0) { if (rare) return pg_launch_read_stats_rare(st, *rare); return hipSuccess; }
    uint32_t m; const uint32_t n_chunks = pg_gather_chunks(n_kept_cap, &m);
    if (rare && rare->B.n_reads) {
        const uint32_t want = rare->wide_blocks < 64 ? 64u : (rare->wide_blocks > 2048 ? 2048u : rare->wide_blocks);
        const uint32_t extra = (want + PG_HUGE_BLOCKS + 3) / 4;
        PG_LAUNCH(k_len_partials<true>, dim3(n_chunks + extra), dim3(256), 0, st, rec, n_kept_ptr, m * 1024u, n_chunks, part, *rare);
    } else PG_LAUNCH(k_len_partials<false>, dim3(n_chunks), dim3(256), 0, st, rec, n_kept_ptr, m * 1024u, n_chunks, part, PgRareArgs{});
    return hipSuccess;
}


hipError_t pg_launch_read_plan(hipStream_t st, const PgDevBatch &B, const uint8_t *read_needed, double pa_min, double pa_max, void *plan_buf,
                         int32_t *flags, int32_t *stat_status, bool flags_are_reset, const PgLongState &LS) {
    if (B.n_reads == 0) { // no record pass: the reset alone
        if (!flags_are_reset) PG_LAUNCH(k_stat_flags_init, dim3(1), dim3(1), 0, st, flags);
        return hipSuccess;
    }
    PG_LAUNCH(k_read_plan, dim3((B.n_reads + 255) / 256), dim3(256), 0, st, B, read_needed, pa_min, pa_max,
                       reinterpret_cast<PgStatRec *>(plan_buf), stat_status, flags_are_reset ? (int32_t *)nullptr : flags, LS);
    return hipSuccess;
}

hipError_t pg_launch_read_stats(hipStream_t st, const PgDevBatch &B, const void *plan_buf, double *med, double *mad, int32_t *status, int32_t *err, int win,
                                uint32_t *wide_list, int32_t *wide_count, uint8_t *oor, int range_only, double *gcal, const PgLongState &LS) {
    if (B.n_reads == 0) return hipSuccess;
    const PgStatRec *plan = reinterpret_cast<const PgStatRec *>(plan_buf);
    static_assert(PG_STATS_WPB == 1, "the helpers of long reads are addressed as workgroups behind the reads' own");
    PG_LAUNCH(k_read_stats, dim3(B.n_reads + (LS.tab ? LS.cap : 0u)), dim3(64 * PG_STATS_WPB), 0, st, B, plan, med, mad, status, err, win, oor, range_only, wide_list, wide_count,
              B.n_reads / PG_STATS_CACHED_FRACTION, gcal, LS);
    return hipSuccess;
}

// several batches' (or, for a pg_job, several ranks') kept samples into the k-mer-major order of the whole job, on the device: one
// workgroup per (k-mer, batch) segment -- or one wave when the segments are short (k = 9 over 8 ranks: 2 M segments of ~90 samples)
template <int PER_WG> __global__ __launch_bounds__(256) void k_merge_segments(const PgSeg *__restrict__ seg, uint32_t n_seg, double *__restrict__ dst) {
    const uint32_t si = PER_WG == 1 ? blockIdx.x : blockIdx.x * PER_WG + (threadIdx.x >> 6);
    if (si >= n_seg) return;
    const PgSeg sg = seg[si];
    const double *__restrict__ src = sg.src;
    double *__restrict__ d = dst + sg.dst_off;
    const uint32_t t = PER_WG == 1 ? threadIdx.x : (uint32_t)lane_id(), step = PER_WG == 1 ? 256u : (uint32_t)WAVE;
    for (uint64_t i = t; i < sg.n; i += step) d[i] = src[i];
}
hipError_t pg_launch_merge_segments(hipStream_t st, const PgSeg *d_seg, uint32_t n_seg, double *dst, uint64_t n_total) {
    if (!n_seg) return hipSuccess;
    if (n_total / n_seg >= 1024) PG_LAUNCH(k_merge_segments<1>, dim3(n_seg), dim3(256), 0, st, d_seg, n_seg, dst);
    else PG_LAUNCH(k_merge_segments<4>, dim3((n_seg + 3) / 4), dim3(256), 0, st, d_seg, n_seg, dst);
    return hipSuccess;
}

hipError_t pg_launch_gather(hipStream_t st, const PgDevBatch &B, uint64_t n_kept_cap, const uint64_t *n_kept_ptr, const PgKeptRec *rec,
                      const uint64_t *samp_off, int scaling, double pa_min,
                      double pa_max, const double *med, const double *mad, double *samples, const double *gcal) {
    if (n_kept_cap == 0) return hipSuccess;
    uint64_t blocks = (n_kept_cap + 15) / 16;
    if (blocks > 256ull * 32) blocks = 256ull * 32;
    PG_LAUNCH(k_gather, dim3((uint32_t)blocks), dim3(256), 0, st, B, n_kept_ptr, rec, samp_off, scaling,
                       pa_min, pa_max, med, mad, samples, gcal);
    return hipSuccess;
}
