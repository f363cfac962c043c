// pg_place.hip -- hand-written gfx950 kernels that PLACE the accepted events of a batch in the reference's output order (k-mer-major,
// inside a k-mer: PAF line, then event index; src/gmove.cpp:732, 891, 925-950) when there are more k-mers than the direct ranking of
// pg_kernels.hip holds (> 1024: k = 9 has 262 144), and the consumers whose work scales with the number of KEPT events.
//
// Partitioned ranking (1024 < slots <= 2^20; round 3). Round 2 sorted (slot, op index) pairs with two LSD radix passes and then needed
// three scattered 4..8-byte stores per kept event to bring window, length and read into k-mer order: 46 M write requests of 32 bytes at
// k = 9, 0.91 ms of a 2.85 ms step (profiles/r03_base_k9_traffic.json). Now every accepted event travels as ONE 16-byte element that
// already holds all a kept event needs -- worked out in source order, where the loads are shared by neighbouring ops:
//   k_events<2>       slots of all ops + per-(tile, high digit) counts                                   (pg_kernels.hip)
//   k_rank_scan       tile prefixes per high digit, region sizes; one extra workgroup: prefix of the 256-op block sums (Bp)
//   k_part_bases      regions = high digits, each padded to whole tiles of 4096 elements; per read: op-sum in front of its first op
//   k_part_scatter    pass A: elements {slot, window start, length, read} partitioned by high digit, stable, through an LDS stage
//                     in digit order (runs of ~8 elements = 128 bytes leave together)
//   k_region_count / k_region_scan   pass B counts: per (region tile, low digit), prefixes along each region -> accepted events per slot
//   (the sample_limit cut: k_slot_keep + offsets, pg_kernels.hip; with a base from other ranks in a multi-GPU job)
//   k_region_place    pass B: rank inside the slot = region-tile prefix + rank inside the tile; kept events (rank < keep) leave as
//                     PgKeptRec records at ev_off[slot] + rank, again in runs through the LDS stage
// Offsets + gather for many kept events (k_len_partials or the placing kernel's coarse sums, k_gather_chunks): the exclusive scan of
// the kept window lengths happens inside the gather's workgroups (chunks of 1024 events; a chunk's base = the coarse sums in front of
// its group of 128 chunks + the fine sums inside it: chunk_base), so the lengths are read once and the offsets written once, and no
// scan launch stands between the placing kernel and the gather.
#include "pg_dev.h"
#include "pg_select.h"
#include <type_traits>

// =====================================================================================================
// op-sum prefixes: window starts of direct reads without a walk
// =====================================================================================================
// sum of op_n over ALL ops of the batch in front of op x, modulo 2^32 (differences inside one read are exact: a read's samples fit
// 31 bits): Bp = exclusive prefix of k_events' 256-op block sums, cum = its in-block sums at 4-op granularity
__device__ __forceinline__ uint32_t op_prefix(const PgDevBatch &B, const PgWalkOut &O, const uint32_t *__restrict__ Bp, uint64_t x) {
    uint32_t s = Bp[x >> 8] + O.cum[x >> 2];
    const uint64_t y = x & ~3ull, last = B.n_ops ? B.n_ops - 1 : 0;
    const uint32_t a = B.op_n[y < last ? y : last], b = B.op_n[y + 1 < last ? y + 1 : last], c = B.op_n[y + 2 < last ? y + 2 : last]; // unconditional: in flight together
    const uint32_t m = (uint32_t)(x & 3);
    s += (m > 0 ? a : 0u) + (m > 1 ? b : 0u) + (m > 2 ? c : 0u);
    return s;
}

// what travels with an accepted event: y = window start in the batch's signal (low 32 bits) or the error code of an event whose
// window is not printable, z = length (24 bits; 0 = error) | high 8 bits of the start, w = read. A kept event's window must be
// printable (gmove.cpp:928-944 is undefined for margin > start or an empty window); whether it is only matters once the event is
// KEPT, so the verdict travels too and k_region_place reports it.
// have_row (sig_move_offset == 0): the caller holds op_n[g] and the sum of the ops in front of g inside its group of four (from its
// neighbouring lanes: no loads)
__device__ __forceinline__ uint4 event_element(const PgDevBatch &B, const PgWalkParams &W, const PgWalkOut &O, const uint32_t *__restrict__ Bp,
                                               uint32_t slot, uint64_t g, uint32_t rd, bool have_row, uint32_t row_len, uint32_t row_partial) {
    const PgReadMeta *mt = O.meta + rd;
    const uint2 lq = *reinterpret_cast<const uint2 *>(&mt->L);     // L, qs
    const uint4 fs = *reinterpret_cast<const uint4 *>(&mt->flags); // flags, opsum0, sig0
    const uint64_t sig0 = (uint64_t)fs.z | ((uint64_t)fs.w << 32);
    const uint64_t ge = g + W.sig_move_offset; // the event's window is that of match i + sig_move_offset
    uint32_t start, len; bool ok = true;
    if (fs.x & PG_RM_GENERIC) { start = O.m_start[ge]; len = O.m_len[ge]; }
    else {
        len = have_row ? row_len : B.op_n[ge];
        const uint32_t pre = have_row ? Bp[ge >> 8] + O.cum[ge >> 2] + row_partial : op_prefix(B, O, Bp, ge);
        const uint64_t st = (uint64_t)lq.y + (uint32_t)(pre - fs.y);
        start = (uint32_t)st;
        ok = st + len <= 0x7fffffffull;
    }
    int code = ok ? 0 : PGR_ERR_RANGE;
    const uint64_t we64 = (uint64_t)start + len + W.print_margin;
    const uint32_t we = (uint32_t)(we64 > lq.x ? lq.x : we64), ws = start - W.print_margin;
    if (ok && (W.print_margin > start || we <= ws)) code = PGR_ERR_WINDOW;
    const uint64_t src = sig0 + ws;
    const uint32_t wl = we - ws;
    if (!code && (wl >= (1u << 24) || (src >> 40))) code = PGR_ERR_RANGE; // (a window of 2^24 samples, a batch of 2^40: not representable here)
    if (code) return make_uint4(slot, (uint32_t)code, 0u, rd);
    return make_uint4(slot, (uint32_t)src, wl | ((uint32_t)(src >> 32) << 24), rd);
}

// the arithmetic of event_element for a direct read whose loads have been made by the caller (k_part_scatter requests the loads of all
// its rows together): lq = {L, qs}, fs = {flags, opsum0, sig0 lo, sig0 hi} of the read's record, pre = op sum in front of the event's op
__device__ __forceinline__ uint4 element_finish(const PgWalkParams &W, uint32_t slot, uint32_t rd, const uint2 &lq, const uint4 &fs, uint32_t len, uint32_t pre) {
    const uint64_t sig0 = (uint64_t)fs.z | ((uint64_t)fs.w << 32);
    const uint64_t st = (uint64_t)lq.y + (uint32_t)(pre - fs.y);
    const uint32_t start = (uint32_t)st;
    const bool ok = st + len <= 0x7fffffffull;
    int code = ok ? 0 : PGR_ERR_RANGE;
    const uint64_t we64 = (uint64_t)start + len + W.print_margin;
    const uint32_t we = (uint32_t)(we64 > lq.x ? lq.x : we64), ws = start - W.print_margin;
    if (ok && (W.print_margin > start || we <= ws)) code = PGR_ERR_WINDOW;
    const uint64_t src = sig0 + ws;
    const uint32_t wl = we - ws;
    if (!code && (wl >= (1u << 24) || (src >> 40))) code = PGR_ERR_RANGE;
    if (code) return make_uint4(slot, (uint32_t)code, 0u, rd);
    return make_uint4(slot, (uint32_t)src, wl | ((uint32_t)(src >> 32) << 24), rd);
}

// =====================================================================================================
// a tile of 4096 elements in stable digit order (512 threads: 8 waves x 8 rows of 64)
// =====================================================================================================
#define PG_PART_WAVES 8
#define PG_PART_THREADS (PG_PART_WAVES * WAVE)
#define PG_PART_ROWS (PG_SORT_TILE / PG_PART_THREADS)
static_assert(PG_PART_ROWS * PG_PART_THREADS == PG_SORT_TILE, "tile geometry");
#define PG_PART_TILES_PER_WG 4 // consecutive tiles of pass A per workgroup: a digit's four tile prefixes are ONE 16-byte load of the [digit][tile] table

struct PartLds {
    uint4 *stage;    // [PG_SORT_TILE] the tile's elements in digit order
    uint32_t *cnt;   // [PG_PART_WAVES][ndig / 2] per-wave counts, two 16-bit digits per word (a wave holds 512 elements, a tile 4096)
    uint32_t *ls;    // [ndig + 1] first stage position of every digit; [ndig] = elements of the tile
    uint32_t *aux;   // [2 * ndig]
    uint32_t *wsum;  // [PG_PART_WAVES]
};
static inline size_t part_lds_bytes(uint32_t ndig) {
    const uint32_t half = ndig / 2 ? ndig / 2 : 1;
    return (size_t)PG_SORT_TILE * 16 + ((size_t)PG_PART_WAVES * half + (ndig + 4) + 2 * ndig + 16) * 4;
}
__device__ __forceinline__ PartLds part_lds(uint32_t ndig) {
    extern __shared__ uint4 pg_part_smem[];
    const uint32_t half = ndig / 2 ? ndig / 2 : 1;
    PartLds L;
    L.stage = pg_part_smem;
    L.cnt = reinterpret_cast<uint32_t *>(pg_part_smem + PG_SORT_TILE);
    L.ls = L.cnt + PG_PART_WAVES * half;
    L.aux = L.ls + ndig + 4;
    L.wsum = L.aux + 2 * ndig;
    return L;
}

// dig / valid: the thread's PG_PART_ROWS elements (element of row r: w * 512 + r * 64 + lane of the tile, i.e. source order = wave,
// row, lane). cnt zeroed and visible (barrier) on entry. On exit (behind a barrier): j[r] = place of the element among the tile's valid
// elements in (digit, source order) order; ls as described above. One pass: the count and the rank inside the wave come from the
// same ordered walk over the rows (ballots find the lanes of equal digit; the lowest of them advances the wave's counter).
// the first half on its own: lrank[r] = elements of the same digit in front of the element inside its WAVE (rows in order, lanes in order);
// the wave's counters hold its totals per digit afterwards. Per row: every valid lane reads its digit's running count, adds one to it
// (LDS atomic, order-free), reads it again -- the difference is how many lanes of the row share the digit. Alone (the usual case:
// 64 lanes over 512..1024 digits): done. Otherwise the lanes of each shared digit find their order with one ballot per such digit
// (round 2's form paid nbits ballots and 64-bit selects for every row: these kernels are bound by instruction issue).
__device__ __forceinline__ void wave_digit_ranks(const uint32_t (&dig)[PG_PART_ROWS], const bool (&valid)[PG_PART_ROWS], int nbits, uint32_t *mycnt, uint32_t (&lrank)[PG_PART_ROWS]) {
    (void)nbits;
    volatile uint32_t *vc = reinterpret_cast<volatile uint32_t *>(mycnt);
#pragma unroll
    for (int r = 0; r < PG_PART_ROWS; ++r) {
        const uint32_t d = dig[r], sh = (d & 1u) * 16u;
        uint32_t b = 0, after = 0;
        if (valid[r]) b = (vc[d >> 1] >> sh) & 0xffffu;
        __builtin_amdgcn_wave_barrier();
        if (valid[r]) atomicAdd(&mycnt[d >> 1], 1u << sh);
        __builtin_amdgcn_wave_barrier();
        if (valid[r]) after = (vc[d >> 1] >> sh) & 0xffffu;
        __builtin_amdgcn_wave_barrier();
        uint32_t below = 0;
        uint64_t shared = __ballot(valid[r] && after - b > 1u); // lanes whose digit another lane of this row holds too
        while (shared) {
            const int l = __ffsll((long long)shared) - 1;
            const uint32_t dl = (uint32_t)__builtin_amdgcn_readlane((int)d, l);
            const uint64_t same = __ballot(valid[r] && d == dl);
            if (d == dl) below = (uint32_t)__popcll(same & lanemask_lt());
            shared &= ~same;
        }
        lrank[r] = b + below;
    }
}
__device__ __forceinline__ void tile_digit_order(const uint32_t (&dig)[PG_PART_ROWS], const bool (&valid)[PG_PART_ROWS], int nbits, uint32_t ndig,
                                                 const PartLds &L, uint32_t (&j)[PG_PART_ROWS]) {
    const uint32_t tid = threadIdx.x, w = tid >> 6, half = ndig / 2 ? ndig / 2 : 1;
    const int lane = lane_id();
    uint32_t lrank[PG_PART_ROWS];
    wave_digit_ranks(dig, valid, nbits, L.cnt + w * half, lrank);
    __syncthreads();
    // per digit: exclusive prefix over the waves (in place, both halves of a word at once: sums stay below 2^16), the tile's total
    uint32_t te = 0, to = 0;
    if (tid < half) {
        uint32_t run = 0;
#pragma unroll
        for (int ww = 0; ww < PG_PART_WAVES; ++ww) { const uint32_t word = L.cnt[ww * half + tid]; L.cnt[ww * half + tid] = run; run += word; }
        te = run & 0xffffu; to = run >> 16;
    }
    const uint32_t s = te + to, inc = wave_incl_scan_u32(s);
    if (lane == WAVE - 1) L.wsum[w] = inc;
    __syncthreads();
    uint32_t off = inc - s;
    for (uint32_t ww = 0; ww < w; ++ww) off += L.wsum[ww];
    if (tid < half) { L.ls[2 * tid] = off; if (2 * tid + 1 < ndig) L.ls[2 * tid + 1] = off + te; }
    if (tid == PG_PART_THREADS - 1) L.ls[ndig] = off + s; // threads behind the digits hold 0: the last thread's prefix is the total
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PG_PART_ROWS; ++r) {
        const uint32_t d = dig[r], sh = (d & 1u) * 16u;
        j[r] = valid[r] ? L.ls[d] + ((L.cnt[w * half + (d >> 1)] >> sh) & 0xffffu) + lrank[r] : 0u;
    }
}

// =====================================================================================================
// pass A
// =====================================================================================================
// workgroup 0: the regions. Region r = the accepted events whose slot has high digit r, padded to whole tiles, so that a tile of pass B
// lies in ONE region: rbase[r] = first element, tile_region[t] = region of tile t, n_tilesB = tiles in use.
// workgroups 1..: per read, the op-sum in front of its first op (op_prefix) and the "generic" flag folded into its record, so that an
// event's element costs ONE record load besides the owner look-up.
__global__ __launch_bounds__(1024) void k_part_bases(const uint32_t *__restrict__ totals, uint32_t R, uint32_t *__restrict__ rbase,
                                                     uint32_t *__restrict__ tile_region, uint32_t *__restrict__ n_tilesB, uint32_t tilesB_cap,
                                                     PgDevBatch B, PgWalkOut O, const uint32_t *__restrict__ Bp) {
    const uint32_t tid = threadIdx.x;
    if (blockIdx.x > 0) {
        const uint32_t r = (blockIdx.x - 1) * 1024 + tid;
        if (r < B.n_reads) {
            PgReadMeta *mt = O.meta + r;
            uint32_t flags = mt->flags;
            const bool gen = O.gen_flag[r] == O.batch_id;
            if (gen) flags |= PG_RM_GENERIC;
            mt->opsum0 = ((flags & PG_RM_DIRECT_OK) && !gen) ? op_prefix(B, O, Bp, mt->o0) : 0u;
            mt->flags = flags;
        }
        return;
    }
    __shared__ uint32_t sb[PG_RANK_MAX_DIGITS + 1], wsum[16];
    const uint32_t nt = tid < R ? (totals[tid] + PG_SORT_TILE - 1) / PG_SORT_TILE : 0u;
    const uint32_t inc = wave_incl_scan_u32(nt);
    if (lane_id() == WAVE - 1) wsum[tid >> 6] = inc;
    __syncthreads();
    uint32_t off = inc - nt, total = 0;
    for (uint32_t w = 0; w < 16; ++w) { if (w < (tid >> 6)) off += wsum[w]; total += wsum[w]; }
    if (tid < R) { sb[tid] = off; rbase[tid] = off * PG_SORT_TILE; }
    if (tid == 0) { sb[R] = total; rbase[R] = total * PG_SORT_TILE; *n_tilesB = total; }
    __syncthreads();
    for (uint32_t t = tid; t < total && t < tilesB_cap; t += 1024) { // the region of tile t: the last r with sb[r] <= t (empty regions share their base with the next one)
        uint32_t lo = 0, hi = R; // invariant: sb[lo] <= t < sb[hi]
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (sb[mid] <= t) lo = mid; else hi = mid; }
        tile_region[t] = lo;
    }
}

// pass A proper: the elements of PG_PART_TILES_PER_WG consecutive tiles of op indices go to their regions
__global__ __launch_bounds__(PG_PART_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_part_scatter(const uint32_t *__restrict__ ev_slot, uint32_t n, uint32_t shift, int nbits, uint32_t n_tiles_pad,
                                                                   const uint32_t *__restrict__ hist, const uint32_t *__restrict__ rbase,
                                                                   PgDevBatch B, PgWalkParams W, PgWalkOut O, const uint32_t *__restrict__ Bp, uint4 *__restrict__ elemA,
                                                                   uint16_t *__restrict__ loA) {
    const uint32_t ndig = 1u << nbits, tid = threadIdx.x, w = tid >> 6;
    const int lane = lane_id();
    const PartLds L = part_lds(ndig);
    const uint32_t tile0 = blockIdx.x * PG_PART_TILES_PER_WG;
    // the tile prefixes of digits tid and tid + 512 for the four tiles: [digit][tile] table, four tiles = one 16-byte load
    uint4 hc[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)}; uint32_t rb[2] = {0, 0};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const uint32_t d = tid + q * PG_PART_THREADS;
        if (d < ndig) { hc[q] = *reinterpret_cast<const uint4 *>(hist + (uint64_t)d * n_tiles_pad + tile0); rb[q] = rbase[d]; }
    }
    for (uint32_t k = 0; k < PG_PART_TILES_PER_WG; ++k) {
        const uint64_t T0 = (uint64_t)(tile0 + k) * PG_SORT_TILE;
        if (T0 >= n) break; // (block-uniform)
        if (k) __syncthreads(); // the previous tile's stage and tables are read until its last thread is through
        const uint32_t tile_first = O.tile_read[tile0 + k];
        for (uint32_t i = tid; i < PG_PART_WAVES * (ndig / 2 ? ndig / 2 : 1); i += PG_PART_THREADS) L.cnt[i] = 0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const uint32_t d = tid + q * PG_PART_THREADS;
            if (d < ndig) L.aux[d] = rb[q] + (k == 0 ? hc[q].x : (k == 1 ? hc[q].y : (k == 2 ? hc[q].z : hc[q].w)));
        }
        // the tile's slots, then -- all rows' loads in flight -- the elements of the accepted events
        uint32_t key[PG_PART_ROWS]; bool valid[PG_PART_ROWS]; uint32_t dig[PG_PART_ROWS];
#pragma unroll
        for (int r = 0; r < PG_PART_ROWS; ++r) {
            const uint64_t g = T0 + w * (PG_PART_ROWS * WAVE) + r * WAVE + lane;
            key[r] = g < n ? ev_slot[g] : PG_INVALID_SLOT;
        }
        // every lane's own op_n: the window length of its event, and -- handed along the lanes -- the ops in front of it inside its group of
        // four, which is what the 4-op granularity of k_events' in-block sums leaves to add (a row starts at a multiple of 64 ops)
        const bool have_row = W.sig_move_offset == 0;
        const uint64_t n_loads = B.n_reads && B.op_off[B.n_reads] < n ? B.op_off[B.n_reads] : n; // (a caller's n_ops that is too large is not followed behind the op arrays)
        uint32_t opn[PG_PART_ROWS];
#pragma unroll
        for (int r = 0; r < PG_PART_ROWS; ++r) {
            const uint64_t g = T0 + w * (PG_PART_ROWS * WAVE) + r * WAVE + lane;
            opn[r] = (have_row && g < n_loads) ? B.op_n[g] : 0u;
        }
#pragma unroll
        for (int r = 0; r < PG_PART_ROWS; ++r) { valid[r] = key[r] != PG_INVALID_SLOT; dig[r] = (key[r] >> shift) & (ndig - 1u); }
        __syncthreads();
        uint32_t j[PG_PART_ROWS];
        tile_digit_order(dig, valid, nbits, ndig, L, j);
        // The elements, straight into the stage at their place. The loads an element needs -- two of its read's record, two of the op-sum
        // tables -- are requested for FOUR rows together with no branch in between (the first form asked for them row by row inside
        // `if (valid)`: sixteen to twenty-four dependent round trips per tile, most of the 26 us a tile took); four, not eight, and behind
        // the ranking rather than in front of it, so that the kernel stays within 128 registers (two workgroups per CU).
        // the read of row r's event: its entry in the tile's table rides in the slot word's upper bits (k_events<2>), else a look-up
        // (rare: a loop of loads); a lane without an event reads the tile's first record and drops the result
        auto read_of = [&](int r) {
            const uint64_t g = T0 + w * (PG_PART_ROWS * WAVE) + r * WAVE + lane;
            const uint32_t rel = key[r] >> PG_PART_REL_SHIFT;
            if (!valid[r]) return tile_first < B.n_reads ? tile_first : 0u;
            return rel != PG_PART_REL_UNKNOWN ? tile_first + rel : owner_of(B, O, g);
        };
        // the ops in front of row r's op inside its group of four, handed along the lanes (all lanes take part: DPP)
        auto partial_of = [&](int r) {
            const uint32_t v1 = dpp_zero<0x111, 0xF>(opn[r]), v2 = dpp_zero<0x112, 0xF>(opn[r]), v3 = dpp_zero<0x113, 0xF>(opn[r]); // lanes - 1, - 2, - 3
            const uint32_t m = (uint32_t)lane & 3u;
            return (m > 0 ? v1 : 0u) + (m > 1 ? v2 : 0u) + (m > 2 ? v3 : 0u);
        };
        if (have_row) {
            constexpr int HB = PG_PART_ROWS / 2;
            // the op sums in front of every op of the wave's 512 from the op_n the lanes hold (as k_rank_emit2, round 5): a wave scan per row
            // on top of the block-sum prefix at the wave's first op (one uniform load) instead of two table loads per row
            uint32_t rowpre[PG_PART_ROWS];
            {
                const uint64_t seg0 = T0 + w * (PG_PART_ROWS * WAVE);
                uint32_t runp = Bp[(seg0 < n ? seg0 : (uint64_t)n - 1u) >> 8];
#pragma unroll
                for (int r = 0; r < PG_PART_ROWS; ++r) {
                    const uint32_t inc = wave_incl_scan_u32(opn[r]);
                    rowpre[r] = runp + inc - opn[r];
                    runp += (uint32_t)__builtin_amdgcn_readlane((int)inc, WAVE - 1);
                }
            }
#pragma unroll
            for (int r0 = 0; r0 < PG_PART_ROWS; r0 += HB) {
                uint2 lq[HB]; uint4 fs[HB]; uint32_t pre[HB], rd[HB];
#pragma unroll
                for (int i = 0; i < HB; ++i) {
                    const int r = r0 + i;
                    rd[i] = read_of(r);
                    const PgReadMeta *mt = O.meta + rd[i];
                    lq[i] = *reinterpret_cast<const uint2 *>(&mt->L);     // L, qs
                    fs[i] = *reinterpret_cast<const uint4 *>(&mt->flags); // flags, opsum0, sig0
                    pre[i] = rowpre[r];
                }
#pragma unroll
                for (int i = 0; i < HB; ++i) {
                    const int r = r0 + i;
                    const uint32_t part_r = partial_of(r); // (DPP: in front of the branch)
                    if (!valid[r]) continue;
                    const uint64_t g = T0 + w * (PG_PART_ROWS * WAVE) + r * WAVE + lane;
                    const uint32_t slot = key[r] & ((1u << PG_PART_REL_SHIFT) - 1u);
                    if (fs[i].x & PG_RM_GENERIC) L.stage[j[r]] = event_element(B, W, O, Bp, slot, g, rd[i], true, opn[r], part_r); // window from the generic walk's arrays
                    else L.stage[j[r]] = element_finish(W, slot, rd[i], lq[i], fs[i], opn[r], pre[i]);
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < PG_PART_ROWS; ++r) {
                const uint32_t part_r = partial_of(r);
                if (!valid[r]) continue;
                const uint64_t g = T0 + w * (PG_PART_ROWS * WAVE) + r * WAVE + lane;
                L.stage[j[r]] = event_element(B, W, O, Bp, key[r] & ((1u << PG_PART_REL_SHIFT) - 1u), g, read_of(r), false, opn[r], part_r);
            }
        }
        __syncthreads();
        // consecutive threads store consecutive elements of a digit's run
        const uint32_t total = L.ls[ndig];
        for (uint32_t jj = tid; jj < total; jj += PG_PART_THREADS) {
            const uint4 e = L.stage[jj];
            const uint32_t d = (e.x >> shift) & (ndig - 1u);
            const uint64_t dst = (uint64_t)L.aux[d] + (jj - L.ls[d]);
            elemA[dst] = e; // (streaming "nt" stores here: 164 -> 229 us, and 168 -> 248 us for pass B's records: profiles/r04_probes.txt)
            loA[dst] = (uint16_t)(e.x & ((1u << shift) - 1u)); // the low digits once more, 2 bytes each: all that pass B's count kernel reads
        }
    }
}

// =====================================================================================================
// pass B
// =====================================================================================================
#define PG_G2_SUB 1024   // events a workgroup of the chunked gather holds in LDS at once: records + offsets = 20 KB (4 per thread in the scan)
#define PG_PLACE_CSPAN 64 // chunks of the gather a region tile's kept events usually span (LDS sums; beyond: global atomics)
__device__ __forceinline__ bool region_tile(uint32_t t, const uint32_t *__restrict__ n_tilesB, const uint32_t *__restrict__ tile_region,
                                            const uint32_t *__restrict__ rbase, const uint32_t *__restrict__ totals, uint32_t &r, uint64_t &first, uint32_t &nv) {
    if (t >= *n_tilesB) return false;
    r = tile_region[t];
    first = (uint64_t)t * PG_SORT_TILE;
    const uint64_t end = (uint64_t)rbase[r] + totals[r];
    nv = end - first < PG_SORT_TILE ? (uint32_t)(end - first) : PG_SORT_TILE;
    return true;
}

// events per (region tile, low digit): histB[tile][digit] (one contiguous row per tile)
__global__ __launch_bounds__(256) void k_region_count(const uint16_t *__restrict__ loA, const uint32_t *__restrict__ n_tilesB, const uint32_t *__restrict__ tile_region,
                                                      const uint32_t *__restrict__ rbase, const uint32_t *__restrict__ totals, int lo_bits, uint32_t *__restrict__ histB) {
    __shared__ uint32_t cnt[PG_RANK_MAX_DIGITS];
    uint32_t r, nv; uint64_t first;
    if (!region_tile(blockIdx.x, n_tilesB, tile_region, rbase, totals, r, first, nv)) return;
    const uint32_t ndig = 1u << lo_bits, tid = threadIdx.x;
    for (uint32_t d = tid; d < ndig; d += 256) cnt[d] = 0;
    __syncthreads();
    // 8 digits (16 bytes) per thread and load
    const uint4 *__restrict__ keys = reinterpret_cast<const uint4 *>(loA + first); // (first is a multiple of the tile: 16-byte aligned)
    uint4 kv[PG_SORT_TILE / 256 / 8];
#pragma unroll
    for (int i = 0; i < PG_SORT_TILE / 256 / 8; ++i) kv[i] = keys[i * 256 + tid];
#pragma unroll
    for (int i = 0; i < PG_SORT_TILE / 256 / 8; ++i) {
        const uint32_t x0 = (i * 256 + tid) * 8u, wd[4] = {kv[i].x, kv[i].y, kv[i].z, kv[i].w};
#pragma unroll
        for (int q = 0; q < 8; ++q) if (x0 + q < nv) atomicAdd(&cnt[(wd[q >> 1] >> (16 * (q & 1))) & (ndig - 1u)], 1u);
    }
    __syncthreads();
    for (uint32_t d = tid; d < ndig; d += 256) histB[(uint64_t)blockIdx.x * ndig + d] = cnt[d];
}

// one workgroup per region, one thread per low digit: exclusive prefixes along the region's tiles (in place); the totals are the
// accepted events of the slots (region << lo_bits | digit) -- what pg_count hands out
__global__ __launch_bounds__(1024) void k_region_scan(uint32_t *__restrict__ histB, const uint32_t *__restrict__ rbase, int lo_bits, uint32_t n_slots,
                                                      uint64_t *__restrict__ acc_cnt, uint64_t *__restrict__ acc_copy) {
    const uint32_t r = blockIdx.x, d = threadIdx.x, ndig = 1u << lo_bits;
    if (d >= ndig) return;
    const uint32_t t0 = rbase[r] / PG_SORT_TILE, t1 = rbase[r + 1] / PG_SORT_TILE;
    uint32_t run = 0;
    uint32_t t = t0;
    for (; t + 4 <= t1; t += 4) { // four independent loads in flight
        uint32_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = histB[(uint64_t)(t + u) * ndig + d];
#pragma unroll
        for (int u = 0; u < 4; ++u) { histB[(uint64_t)(t + u) * ndig + d] = run; run += v[u]; }
    }
    for (; t < t1; ++t) { const uint32_t v = histB[(uint64_t)t * ndig + d]; histB[(uint64_t)t * ndig + d] = run; run += v; }
    const uint64_t slot = ((uint64_t)r << lo_bits) | d;
    if (slot < n_slots) { acc_cnt[slot] = run; if (acc_copy) acc_copy[slot] = run; }
}

// The same + the sample_limit cut of pg_submit (base = the context's running counts; gmove.cpp:925-927, 945-950) in ONE launch (round 5;
// k_slot_cut behind k_region_scan was 27 us of chained scan + a kernel boundary): a region's workgroup holds its slots' counts in registers,
// cuts them, scans the kept counts inside the region, PUBLISHES the region's kept total and full count, and adds up the words of the
// regions in front of it -- every workgroup of the launch publishes after its own column pass and waits for nobody before that, so there
// is no chain: one round of loads. A word = kept total (40 bits) | complete k-mers (11 bits) << 40 | epoch (13 bits) << 51, one relaxed
// agent-scope store / load; the epoch is the launch's serial number (1 .. 8191; the host clears the table when it starts again), so
// nothing is reset between launches but the two counters in front of it (by the workgroup that finishes last). Regions are handed
// out by ticket: a waiting workgroup only waits for workgroups that have started.
struct PgRegionCut {
    const uint64_t *running_in; uint64_t *running_out, *keep, *ev_off, *totals;
    uint32_t *keep32;
    uint64_t *state; // [0] lo: regions handed out, hi: regions done; [1 + r] region words
    uint32_t limit, epoch;
};
#define PG_RCUT_EPOCHS 8191u
__global__ __launch_bounds__(1024) void k_region_scan_cut(uint32_t *__restrict__ histB, const uint32_t *__restrict__ rbase, int lo_bits, uint32_t n_slots, uint32_t R,
                                                          uint64_t *__restrict__ acc_cnt, PgRegionCut Q) {
    __shared__ uint32_t sh_r, sh_last; __shared__ uint64_t wsum[16], wfull[16], sh_base, sh_fullb;
    const uint32_t d = threadIdx.x, ndig = 1u << lo_bits, w = d >> 6;
    const int lane = lane_id();
    uint32_t *__restrict__ tickets = reinterpret_cast<uint32_t *>(Q.state);
    if (d == 0) sh_r = __hip_atomic_fetch_add(tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const uint32_t r = sh_r;
    uint32_t run = 0;
    const uint64_t slot = ((uint64_t)r << lo_bits) | d;
    const bool mine = d < ndig && slot < n_slots;
    if (d < ndig) { // exclusive prefixes along the region's tiles (in place), as k_region_scan
        const uint32_t t0 = rbase[r] / PG_SORT_TILE, t1 = rbase[r + 1] / PG_SORT_TILE;
        uint32_t t = t0;
        for (; t + 4 <= t1; t += 4) {
            uint32_t v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = histB[(uint64_t)(t + u) * ndig + d];
#pragma unroll
            for (int u = 0; u < 4; ++u) { histB[(uint64_t)(t + u) * ndig + d] = run; run += v[u]; }
        }
        for (; t < t1; ++t) { const uint32_t v = histB[(uint64_t)t * ndig + d]; histB[(uint64_t)t * ndig + d] = run; run += v; }
    }
    uint32_t kp = 0; uint32_t isfull = 0;
    if (mine) {
        const uint64_t b = Q.running_in[slot];
        const uint64_t rm = b >= Q.limit ? 0 : (uint64_t)Q.limit - b;
        kp = (uint32_t)((uint64_t)run < rm ? run : rm);
        isfull = (Q.limit > 0 && b + run >= Q.limit) ? 1u : 0u; // at limit 0 no k-mer ever completes (see k_slot_plan)
        acc_cnt[slot] = run; Q.running_out[slot] = b + run; Q.keep[slot] = kp; Q.keep32[slot] = kp;
    }
    const uint32_t inc = wave_incl_scan_u32(kp);
    const uint32_t nf = (uint32_t)__popcll(__ballot(isfull != 0));
    if (lane == WAVE - 1) { wsum[w] = inc; wfull[w] = nf; }
    __syncthreads();
    uint64_t off = 0, tot = 0, full = 0;
    for (uint32_t ww = 0; ww < (blockDim.x >> 6); ++ww) { if (ww < w) off += wsum[ww]; tot += wsum[ww]; full += wfull[ww]; }
    const uint64_t ep = (uint64_t)Q.epoch << 51;
    uint64_t *__restrict__ words = Q.state + 1;
    if (d == 0) __hip_atomic_store(reinterpret_cast<unsigned long long *>(words + r), (unsigned long long)(tot | (full << 40) | ep), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // the regions in front: every thread takes some of their words, all requested together; re-asked until they carry this launch's epoch
    uint64_t bsum = 0, fsum = 0;
    for (uint32_t i0 = 0; i0 < r; i0 += blockDim.x) {
        const uint32_t i = i0 + d;
        uint64_t wv = ep;
        if (i < r) {
            do { wv = __hip_atomic_load(reinterpret_cast<unsigned long long *>(words + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if ((wv >> 51) != Q.epoch) __builtin_amdgcn_s_sleep(1); } while ((wv >> 51) != Q.epoch);
        }
        bsum += wv & ((1ull << 40) - 1); fsum += (wv >> 40) & 0x7FFull;
    }
    for (int o = 32; o >= 1; o >>= 1) { bsum += __shfl_xor(bsum, o, WAVE); fsum += __shfl_xor(fsum, o, WAVE); }
    __syncthreads(); // (wsum / wfull have been read)
    if (lane == 0) { wsum[w] = bsum; wfull[w] = fsum; }
    __syncthreads();
    if (d == 0) { uint64_t b2 = 0, f2 = 0; for (uint32_t ww = 0; ww < (blockDim.x >> 6); ++ww) { b2 += wsum[ww]; f2 += wfull[ww]; } sh_base = b2; sh_fullb = f2; }
    __syncthreads();
    const uint64_t base = sh_base;
    if (mine) Q.ev_off[slot] = base + off + inc - kp;
    if (r == R - 1 && d == 0) { // the last region knows the job's totals
        Q.ev_off[n_slots] = base + tot; Q.totals[0] = base + tot; Q.totals[1] = sh_fullb + full;
        Q.totals[3] = ~0ull >> 1; // (read as a signed tile index by the direct ranking only)
    }
    // slots of a region that lies beyond n_slots entirely hold nothing; the counters for the next launch by whoever finishes last
    if (d == 0) sh_last = __hip_atomic_fetch_add(tickets + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == R - 1;
    __syncthreads();
    if (sh_last && d == 0) { __hip_atomic_store(tickets, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(tickets + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}

// pass B proper: the kept events of a region tile leave as records, in k-mer-major order
__global__ __launch_bounds__(PG_PART_THREADS) __attribute__((amdgpu_waves_per_eu(2, 4))) void k_region_place(const uint4 *__restrict__ elemA, const uint32_t *__restrict__ n_tilesB, const uint32_t *__restrict__ tile_region,
                                                                   const uint32_t *__restrict__ rbase, const uint32_t *__restrict__ totals, int lo_bits, uint32_t n_slots,
                                                                   const uint32_t *__restrict__ histB, const uint32_t *__restrict__ keep32, const uint64_t *__restrict__ ev_off,
                                                                   PgWalkOut O, PgKeptOut K, uint64_t *__restrict__ part, uint32_t chunk_shift) {
    const uint32_t ndig = 1u << lo_bits, tid = threadIdx.x, w = tid >> 6;
    const int lane = lane_id();
    const uint32_t nB = *n_tilesB; // (uniform; <= gridDim.x)
    if (blockIdx.x >= nB) return;
    const uint32_t tB = xcd_contiguous(blockIdx.x, nB); // an XCD takes a contiguous range of region tiles (pg_dev.h)
    // the tile's elements are requested in front of the look-ups that say how many of them count (the buffer holds every tile of the grid)
    uint32_t ex[PG_PART_ROWS], ey[PG_PART_ROWS], ez[PG_PART_ROWS], ew[PG_PART_ROWS];
#pragma unroll
    for (int rr = 0; rr < PG_PART_ROWS; ++rr) {
        const uint4 v = elemA[(uint64_t)tB * PG_SORT_TILE + w * (PG_PART_ROWS * WAVE) + rr * WAVE + lane];
        ex[rr] = v.x; ey[rr] = v.y; ez[rr] = v.z; ew[rr] = v.w;
    }
    uint32_t r, nv; uint64_t first;
    if (!region_tile(tB, n_tilesB, tile_region, rbase, totals, r, first, nv)) return;
    const PartLds L = part_lds(ndig);
    // window lengths of this tile's kept events per chunk of the gather. 32 bits are enough: `part` is only passed when the caller's
    // "chunked" condition holds -- (longest window + 1) * 4096 < 2^32 (pg_api.hip collect_impl) -- and a tile holds 4096 events
    __shared__ uint32_t csum[PG_PLACE_CSPAN];
    if (tid < PG_PLACE_CSPAN) csum[tid] = 0;
    const uint64_t c_lo = ev_off[(uint64_t)r << lo_bits] >> chunk_shift; // first chunk the region's kept events can fall into
    for (uint32_t i = tid; i < PG_PART_WAVES * (ndig / 2 ? ndig / 2 : 1); i += PG_PART_THREADS) L.cnt[i] = 0;
    // per digit (= slot of this region): events of the slot in the region's earlier tiles, how many it still keeps, where they go
    uint32_t tp[2] = {0, 0}, kp[2] = {0, 0}, eo[2] = {0, 0};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const uint32_t d = tid + q * PG_PART_THREADS;
        const uint64_t slot = ((uint64_t)r << lo_bits) | d;
        if (d < ndig && slot < n_slots) { tp[q] = histB[(uint64_t)tB * ndig + d]; kp[q] = keep32[slot]; eo[q] = (uint32_t)ev_off[slot]; }
    }
    bool valid[PG_PART_ROWS]; uint32_t dig[PG_PART_ROWS];
#pragma unroll
    for (int rr = 0; rr < PG_PART_ROWS; ++rr) {
        const uint32_t x = w * (PG_PART_ROWS * WAVE) + rr * WAVE + lane;
        valid[rr] = x < nv;
        dig[rr] = ex[rr] & (ndig - 1u);
    }
    __syncthreads();
    uint32_t j[PG_PART_ROWS];
    tile_digit_order(dig, valid, lo_bits, ndig, L, j);
#pragma unroll
    for (int rr = 0; rr < PG_PART_ROWS; ++rr) if (valid[rr]) L.stage[j[rr]] = make_uint4(ex[rr], ey[rr], ez[rr], ew[rr]);
    // element at stage position jj (digit d) has rank tp[d] + jj - ls[d] inside its slot: kept iff jj < ls[d] + (keep - tp), and then it
    // goes to ev_off + rank = jj + (ev_off + tp - ls[d])
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const uint32_t d = tid + q * PG_PART_THREADS;
        if (d < ndig) { const uint32_t lsd = L.ls[d]; L.aux[d] = lsd + (kp[q] > tp[q] ? kp[q] - tp[q] : 0u); L.aux[ndig + d] = eo[q] + tp[q] - lsd; }
    }
    __syncthreads();
    const uint32_t total = L.ls[ndig];
    for (uint32_t jj = tid; jj < total; jj += PG_PART_THREADS) {
        const uint4 e = L.stage[jj];
        const uint32_t d = e.x & (ndig - 1u);
        if (jj >= L.aux[d]) continue; // its k-mer is full (gmove.cpp:925-927)
        const uint32_t dst = jj + L.aux[ndig + d], len = e.z & 0xffffffu;
        if (len == 0) { report_error(O, e.w, (int)e.y); K.rec[dst] = PgKeptRec{0, 0, e.w}; } // the verdict event_element left
        else K.rec[dst] = PgKeptRec{(uint64_t)e.y | ((uint64_t)(e.z >> 24) << 32), len, e.w};
        if (K.read_needed) K.read_needed[e.w] = 1;
        if (!part) continue;
        const uint64_t ch = ((uint64_t)dst >> chunk_shift) - c_lo;
        if (ch < PG_PLACE_CSPAN) atomicAdd(&csum[ch], len);
        else { // a region that keeps more than 64 chunks of events
            atomicAdd(reinterpret_cast<unsigned long long *>(part + ((uint64_t)dst >> chunk_shift)), (unsigned long long)len);
            atomicAdd(reinterpret_cast<unsigned long long *>(part + PG_CHUNK_FINE + ((uint64_t)dst >> chunk_shift >> 6)), (unsigned long long)len);
        }
    }
    __syncthreads();
    if (part && tid < PG_PLACE_CSPAN) { // (the first wave) integer sums: any order. The fine sums, one add per chunk the tile touched ...
        const uint32_t v = csum[tid];
        if (v) atomicAdd(reinterpret_cast<unsigned long long *>(part + c_lo + tid), (unsigned long long)v);
        // ... and the coarse sums of their groups of 64 chunks (pg_internal.h: PG_CHUNK_FINE): the 64 chunks span at most two groups, each
        // reduced in the wave to ONE add (64 adds to one address per workgroup took this kernel from 159 to 310 us)
        static_assert(PG_PLACE_CSPAN == WAVE, "one wave holds the tile's chunk sums");
        const uint64_t g0 = c_lo >> 6;
        const bool second = ((c_lo + tid) >> 6) != g0;
        uint64_t a = second ? 0ull : v, b = second ? v : 0ull;
        for (int o = 32; o >= 1; o >>= 1) { a += __shfl_xor(a, o, WAVE); b += __shfl_xor(b, o, WAVE); }
        if (tid == 0 && a) atomicAdd(reinterpret_cast<unsigned long long *>(part + PG_CHUNK_FINE + g0), (unsigned long long)a);
        if (tid == 1 && b) atomicAdd(reinterpret_cast<unsigned long long *>(part + PG_CHUNK_FINE + g0 + 1), (unsigned long long)b);
    }
}

// =====================================================================================================
// direct ranking (<= 1024 slots), the placing kernel for MANY useful tiles (round 3)
// =====================================================================================================
// k_rank_emit (pg_kernels.hip) is built for the default limit, where a few dozen tiles place events: 16 waves per tile, 64 KB of per-wave
// counters, the tile's column of the [slot][tile] table gathered line by line. With every tile useful (sample_limit 5000: 1588 tiles,
// 2.1 M kept events) that cost 128 us = 6 % of the HBM roofline. Here: 8 waves per tile and 16 KB of packed 16-bit counters (three
// workgroups per CU), count and rank in one ordered walk, FOUR consecutive tiles per workgroup so that a slot's four tile prefixes
// are one 16-byte load. A kept event's window: what it needs of its read (signal offset, query start, length, the op sums in front of the
// read's first op) comes from a table of the tile's first PG_EV_TBL reads that is filled once per tile in LDS; its own op_n is the lane's
// row value, the ops in front of it inside its group of four come along the lanes (DPP); the block-sum prefix adds the rest. 128 us ->
// 77 (ranking as above) -> 61 (the table and the row). The ranking alone takes 35 us (PG_PROBE_EMIT2_NOWIN).
#define PG_EMIT2_TILES 4
#ifndef PG_EMIT2_WAVES
#define PG_EMIT2_WAVES 4 // waves per SIMD the kernel is compiled for (6 = three workgroups per CU, if it fits 80 registers)
#endif
__global__ __launch_bounds__(PG_PART_THREADS) __attribute__((amdgpu_waves_per_eu(PG_EMIT2_WAVES, 6))) void k_rank_emit2(const uint32_t *__restrict__ keys, uint32_t n, int nbits, uint32_t n_slots, uint32_t n_tiles,
        const uint32_t *__restrict__ hist, const uint64_t *__restrict__ keep, const uint64_t *__restrict__ ev_off, const uint64_t *__restrict__ totals,
        PgDevBatch B, PgWalkParams W, PgWalkOut O, PgKeptOut K, const uint32_t *__restrict__ Bp) {
    __shared__ uint32_t cnt[PG_PART_WAVES * (PG_RANK_MAX_DIGITS / 2)];
    __shared__ uint32_t s_hcol[PG_RANK_MAX_DIGITS], s_keep[PG_RANK_MAX_DIGITS], s_off[PG_RANK_MAX_DIGITS];
    // the tile's reads (those k_events could name in an event's upper bits): what a kept event needs of its read, fetched ONCE per tile
    // and read instead of five scattered record loads + the op sums in front of the read's first op per kept event
    __shared__ uint64_t s_rsig0[PG_EV_TBL];
    __shared__ uint32_t s_rqs[PG_EV_TBL], s_rL[PG_EV_TBL], s_rpre[PG_EV_TBL], s_rgen[PG_EV_TBL];
    const uint32_t tid = threadIdx.x, w = tid >> 6, ndig = 1u << nbits, half = ndig / 2 ? ndig / 2 : 1;
    const int lane = lane_id();
    const int64_t last_tile = (int64_t)totals[3];
    const bool have_row = W.sig_move_offset == 0;
    const uint64_t n_loads = B.n_reads && B.op_off[B.n_reads] < n ? B.op_off[B.n_reads] : n; // (a caller's n_ops that is too large is not followed behind the op arrays)
    const uint32_t n_groups = (n_tiles + PG_EMIT2_TILES - 1) / PG_EMIT2_TILES;
    bool first = true;
    for (uint32_t q = blockIdx.x; q < n_groups && (int64_t)q * PG_EMIT2_TILES <= last_tile; q += gridDim.x) {
        const uint32_t tile0 = q * PG_EMIT2_TILES;
        // slots tid and tid + 512: the four tile prefixes (one 16-byte load each: n_tiles is a multiple of 4), keep, offset
        uint4 hc[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)}; uint32_t kp[2] = {0, 0}, eo[2] = {0, 0};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const uint32_t d = tid + u * PG_PART_THREADS;
            if (d < n_slots) { hc[u] = *reinterpret_cast<const uint4 *>(hist + (uint64_t)d * n_tiles + tile0); kp[u] = (uint32_t)keep[d]; eo[u] = (uint32_t)ev_off[d]; }
        }
        bool stop = false;
        for (uint32_t k = 0; k < PG_EMIT2_TILES && !stop; ++k) {
            const uint32_t tile = tile0 + k;
            const uint64_t T0 = (uint64_t)tile * PG_SORT_TILE;
            if (T0 >= n || (int64_t)tile > last_tile) break; // (block-uniform)
            if (!first) __syncthreads(); // the previous tile's tables are read until its last thread is through
            first = false;
            const uint32_t tile_first = O.tile_read[tile];
            uint32_t key[PG_PART_ROWS];
#pragma unroll
            for (int r = 0; r < PG_PART_ROWS; ++r) { const uint64_t g = T0 + w * (PG_PART_ROWS * WAVE) + r * WAVE + lane; key[r] = g < n ? keys[g] : PG_INVALID_SLOT; }
            // every lane's own op_n (sig_move_offset 0): the window length of its event and, handed along the lanes, the ops in front of it
            // inside its group of four -- what the 4-op granularity of k_events' in-block sums leaves to add (k_part_scatter does the same)
            uint32_t opn[PG_PART_ROWS];
#pragma unroll
            for (int r = 0; r < PG_PART_ROWS; ++r) { const uint64_t g = T0 + w * (PG_PART_ROWS * WAVE) + r * WAVE + lane; opn[r] = (have_row && g < n_loads) ? B.op_n[g] : 0u; }
            for (uint32_t i = tid; i < PG_PART_WAVES * half; i += PG_PART_THREADS) cnt[i] = 0;
            int any = 0; // does any slot still have room at this tile's position in the (read, event) order?
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const uint32_t d = tid + u * PG_PART_THREADS;
                if (d < ndig) {
                    const uint32_t h = k == 0 ? hc[u].x : (k == 1 ? hc[u].y : (k == 2 ? hc[u].z : hc[u].w));
                    s_hcol[d] = h; s_keep[d] = kp[u]; s_off[d] = eo[u];
                    if (d < n_slots && h < kp[u]) any = 1;
                }
            }
            if (tid >= PG_PART_THREADS - PG_EV_TBL) { // (the last two waves: the first ones carry the slot columns above)
                const uint32_t t = tid - (PG_PART_THREADS - PG_EV_TBL), r = tile_first + t;
                if (r < B.n_reads) {
                    const KeptRead kr = kept_read(O, r);
                    s_rsig0[t] = kr.sig0; s_rqs[t] = kr.qs; s_rL[t] = kr.L; s_rgen[t] = kr.generic ? 1u : 0u;
                    s_rpre[t] = kr.generic ? 0u : op_prefix(B, O, Bp, kr.o0);
                }
            }
            if (!__syncthreads_or(any)) { stop = true; break; } // every k-mer this tile could feed is full (gmove.cpp:925-927), and so are the tiles behind it
            bool valid[PG_PART_ROWS]; uint32_t dig[PG_PART_ROWS], lrank[PG_PART_ROWS];
#pragma unroll
            for (int r = 0; r < PG_PART_ROWS; ++r) { valid[r] = key[r] != PG_INVALID_SLOT; dig[r] = key[r] & (ndig - 1u); }
            wave_digit_ranks(dig, valid, nbits, cnt + w * half, lrank);
            __syncthreads();
            if (tid < half) { // exclusive prefix over the waves, in place (both 16-bit halves of a word at once)
                uint32_t run = 0;
#pragma unroll
                for (int ww = 0; ww < PG_PART_WAVES; ++ww) { const uint32_t word = cnt[ww * half + tid]; cnt[ww * half + tid] = run; run += word; }
            }
            __syncthreads();
            // the kept events: rank < keep. Their windows in two rounds of independent loads: the read's record, then the op sums
            uint32_t dst[PG_PART_ROWS];
#pragma unroll
            for (int r = 0; r < PG_PART_ROWS; ++r) {
                const uint32_t d = dig[r], sh = (d & 1u) * 16u;
                const uint32_t rank = s_hcol[d] + ((cnt[w * half + (d >> 1)] >> sh) & 0xffffu) + lrank[r];
                dst[r] = (valid[r] && rank < s_keep[d]) ? s_off[d] + rank : 0xFFFFFFFFu;
            }
            // the op sums in front of every op of the wave's 512 (round 5): a wave scan per row over the op_n the lanes hold, from the block-sum
            // prefix at the wave's first op (a multiple of 256 ops: ONE uniform load) -- the window loop below then needs no load of the op-sum
            // tables at all (it asked for two per kept row, inside its branches: eight dependent round trips per tile)
            uint32_t rowpre[PG_PART_ROWS];
            if (have_row) {
                const uint64_t seg0 = T0 + w * (PG_PART_ROWS * WAVE);
                uint32_t runp = Bp[(seg0 < n ? seg0 : (uint64_t)n - 1u) >> 8];
#pragma unroll
                for (int r = 0; r < PG_PART_ROWS; ++r) {
                    const uint32_t inc = wave_incl_scan_u32(opn[r]);
                    rowpre[r] = runp + inc - opn[r];
                    runp += (uint32_t)__builtin_amdgcn_readlane((int)inc, WAVE - 1);
                }
            }
#pragma unroll
            for (int r = 0; r < PG_PART_ROWS; ++r) {
                if (dst[r] == 0xFFFFFFFFu) continue;
                const uint64_t g = T0 + w * (PG_PART_ROWS * WAVE) + r * WAVE + lane, ge = g + W.sig_move_offset;
                const uint32_t rel = key[r] >> PG_SLOT_BITS;
                uint32_t rd, ws, wl, qs, L, pre0; uint64_t sig0; bool gen;
                if (rel < PG_EV_TBL) { rd = tile_first + rel; sig0 = s_rsig0[rel]; qs = s_rqs[rel]; L = s_rL[rel]; pre0 = s_rpre[rel]; gen = s_rgen[rel] != 0; }
                else { // a tile that more than PG_EV_TBL reads touch: the read is named beyond the table, or not at all
                    rd = rel != PG_REL_UNKNOWN ? tile_first + rel : owner_of(B, O, g);
                    const KeptRead kr = kept_read(O, rd);
                    sig0 = kr.sig0; qs = kr.qs; L = kr.L; gen = kr.generic; pre0 = gen ? 0u : op_prefix(B, O, Bp, kr.o0);
                }
                bool ok = true;
                if (gen) { ws = O.m_start[ge]; wl = O.m_len[ge]; }
                else {
                    wl = have_row ? opn[r] : B.op_n[ge];
                    const uint32_t pre = have_row ? rowpre[r] : op_prefix(B, O, Bp, ge);
                    const uint64_t st = (uint64_t)qs + (uint32_t)(pre - pre0);
                    ws = (uint32_t)st;
                    ok = st + wl <= 0x7fffffffull;
                }
                if (!ok) { report_error(O, rd, PGR_ERR_RANGE); ws = 0; wl = 0; }
                uint32_t start = ws - W.print_margin;
                const uint64_t we64 = (uint64_t)ws + wl + W.print_margin;
                uint32_t we = (uint32_t)(we64 > L ? L : we64);
                // a kept event's window must be printable (gmove.cpp:928-944 is undefined for margin > start or an empty window); ONE verdict
                // per event, as event_element leaves it for the partitioned path: a window that failed the range check is not judged again
                if (!ok) start = we = 0;
                else if (W.print_margin > ws || we <= start) { report_error(O, rd, PGR_ERR_WINDOW); start = we = 0; }
                K.rec[dst[r]] = PgKeptRec{sig0 + start, we - start, rd};
                if (K.read_needed) K.read_needed[rd] = 1;
                // (the chunked gather's chunk sums are NOT accumulated here as k_region_place does: a tile's kept events go to ~1000 different
                // k-mers, i.e. chunks -- nothing to combine in LDS first, and 2.1 M global 64-bit atomics took the kernel from 77 to 190 us)
            }
        }
        if (stop) break;
    }
}

// =====================================================================================================
// sample offsets + gather of many kept events in one kernel (gmove.cpp:773-775, 938-944)
// =====================================================================================================

// a gather workgroup's base: the kept samples in front of chunk c = the coarse sums of the groups of 64 chunks below its own + the fine
// sums of its group below it (pg_internal.h: PG_CHUNK_FINE); every wave works it out for itself (three loads per lane and a reduction)
__device__ __forceinline__ uint64_t chunk_base(const uint64_t *__restrict__ part, uint32_t c) {
    const uint32_t lane = (uint32_t)lane_id(), nc = c >> 6, f0 = c & ~63u;
    uint64_t s = 0;
    for (uint32_t i = lane; i < nc; i += WAVE) s += part[PG_CHUNK_FINE + i];
    if (f0 + lane < c) s += part[f0 + lane];
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, WAVE);
    return s;
}

// one workgroup per chunk of sub_per_chunk * PG_G2_SUB kept events: the exclusive scan of their window lengths (chunk base from
// chunk_base), the offsets written for the batch's other consumers, and the gather of the windows -- G lanes per event as k_gather.
// The gather is a chain record -> window -> stores per event, and what bounds it is how many of those chains the chip has in flight
// (random ~25..60-byte reads: scatter_probe): E events per lane group and trip have their loads requested together.
#ifndef PG_GC_EVENTS
#define PG_GC_EVENTS 2
#endif
#ifndef PG_GC_WAVES
#define PG_GC_WAVES 4 // waves per SIMD the chunked gather is compiled for
#endif
// The sub-chunk's records sit in LDS (they came in with one coalesced pass, which also feeds the scan of their lengths): an event's
// chain is then LDS -> window + calibration (one memory round) -> stores, not record -> window -> stores (two)
template <int G, int P>
__device__ __forceinline__ void gather_chunk(const PgDevBatch &B, uint64_t total, const uint4 *s_rec, uint32_t cnt, uint64_t base,
                                             const uint32_t *s_off, int scaling, double pa_min, double pa_max, const double *__restrict__ gcal, double *__restrict__ samples) {
    const int lane = lane_id();
    const int g0 = lane & ~(G - 1);
    const uint32_t sub = (uint32_t)lane & (uint32_t)(G - 1);
    constexpr int E = PG_GC_EVENTS;
    constexpr uint32_t STEP = 256 / G;
    for (uint32_t i0 = threadIdx.x / G; i0 < cnt; i0 += E * STEP) {
        uint32_t len[E], so[E]; uint64_t src[E]; GatherRegs<P> R[E];
#pragma unroll
        for (int u = 0; u < E; ++u) {
            const uint32_t i = i0 + u * STEP;
            const uint4 r = s_rec[i < cnt ? i : 0u]; // every lane of the group reads the same 16 bytes: a broadcast
            so[u] = s_off[i < cnt ? i : 0u];
            len[u] = i < cnt ? r.z : 0u; src[u] = (uint64_t)r.x | ((uint64_t)r.y << 32);
            if (i < cnt) gather_load<G, P>(B, sub, r.w, len[u], src[u], total, scaling, nullptr, nullptr, gcal, R[u]);
        }
#pragma unroll
        for (int u = 0; u < E; ++u) {
            if (i0 + u * STEP < cnt) gather_finish<G, P>(B, sub, g0, len[u], src[u], base + so[u], total, scaling, pa_min, pa_max, samples, R[u], gcal != nullptr);
            __builtin_amdgcn_sched_barrier(0); // one event's conversions (FP64 divisions: ~20 registers each) at a time
        }
    }
}
template <int G, int P> __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PG_GC_WAVES, PG_GC_WAVES))) void k_gather_chunks(PgDevBatch B, const uint64_t *__restrict__ n_kept_ptr, const PgKeptRec *__restrict__ rec, const uint64_t *__restrict__ part,
                                                       uint32_t sub_per_chunk, uint64_t *__restrict__ samp_off, uint64_t *__restrict__ total_out, int scaling, double pa_min, double pa_max,
                                                       double *__restrict__ samples, const double *__restrict__ gcal) {
    __shared__ uint4 s_rec[PG_G2_SUB];
    __shared__ uint32_t s_off[PG_G2_SUB];
    __shared__ uint32_t wsum[4];
    const uint64_t n_kept = n_kept_ptr[0];
    const uint64_t c0 = (uint64_t)blockIdx.x * sub_per_chunk * PG_G2_SUB;
    if (c0 >= n_kept) { if (n_kept == 0 && blockIdx.x == 0 && threadIdx.x == 0) { samp_off[0] = 0; total_out[0] = 0; } return; }
    const uint64_t total = B.sig_off[B.n_reads];
    const uint32_t tid = threadIdx.x;
    uint64_t run = chunk_base(part, blockIdx.x);
    for (uint32_t sc = 0; sc < sub_per_chunk; ++sc) {
        const uint64_t e0 = c0 + (uint64_t)sc * PG_G2_SUB;
        if (e0 >= n_kept) break;
        const uint32_t cnt = n_kept - e0 < PG_G2_SUB ? (uint32_t)(n_kept - e0) : PG_G2_SUB;
        if (sc) __syncthreads(); // the previous sub-chunk's gather reads the LDS arrays until its last thread is through
        // the records: coalesced (event i * 256 + tid: a wave takes 1 KB), into LDS
        const uint4 *__restrict__ rv = reinterpret_cast<const uint4 *>(rec + e0);
        uint4 q[PG_G2_SUB / 256];
#pragma unroll
        for (int i = 0; i < PG_G2_SUB / 256; ++i) { const uint32_t x = i * 256 + tid; q[i] = x < cnt ? rv[x] : make_uint4(0, 0, 0, 0); }
#pragma unroll
        for (int i = 0; i < PG_G2_SUB / 256; ++i) s_rec[i * 256 + tid] = q[i];
        __syncthreads();
        // exclusive scan of the lengths, consecutive events per thread
        constexpr int PT = PG_G2_SUB / 256;
        uint32_t v[PT], s = 0;
#pragma unroll
        for (int i = 0; i < PT; ++i) { v[i] = s_rec[tid * PT + i].z; s += v[i]; }
        const uint32_t inc = wave_incl_scan_u32(s); // (a sub-chunk's samples fit 32 bits: the caller's bound on the window length)
        if (lane_id() == WAVE - 1) wsum[tid >> 6] = inc;
        __syncthreads();
        uint32_t off = inc - s, tot = 0;
        for (uint32_t w = 0; w < 4; ++w) { if (w < (tid >> 6)) off += wsum[w]; tot += wsum[w]; }
#pragma unroll
        for (int i = 0; i < PT; ++i) { s_off[tid * PT + i] = off; off += v[i]; }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PG_G2_SUB / 256; ++i) { const uint32_t x = i * 256 + tid; if (x < cnt) samp_off[e0 + x] = run + s_off[x]; }
        gather_chunk<G, P>(B, total, s_rec, cnt, run, s_off, scaling, pa_min, pa_max, gcal, samples);
        run += tot;
    }
    if (c0 + (uint64_t)sub_per_chunk * PG_G2_SUB >= n_kept && tid == 0) { samp_off[n_kept] = run; total_out[0] = run; } // the last chunk: all kept samples
}

// ---- the wave form (round 4): one lane per PAIR OF OUTPUT SAMPLES instead of a lane group per event -------------------------------------
// k_gather_chunks gives an event G lanes and walks its window in passes of 2 * G samples: at k = 9 (mean window 12.4, a third of the
// windows longer than 16) 35 % of the lane-slots it issues carry a sample, and its stores are ragged 16-byte pieces. Here the OUTPUT range
// is the index space: lane j of a trip owns the 16-byte aligned pair of output samples 2j, 2j + 1, finds the event(s) they belong to,
// loads their int16 samples, converts and writes one 16-byte store -- every lane busy, every store instruction 1 KB of consecutive
// doubles. The event of an output sample comes from a bit map of the window starts over the group's sample range, kept in LDS with the
// running count of set bits per word: event = prefix + popcount(word & mask) - 1, one 8-byte LDS read. Per event the stage holds
// (source - offset) -- a sample's source index is that plus its output index --, the read's calibration and the reciprocal of its MAD.
// A WAVE owns 64 consecutive events at a time and a 4 KB slice of the LDS: one 16-byte record per lane, offsets by DPP scan, bit map +
// per-word prefix by the wave alone (LDS operations of one wave execute in order), then its trips over the group's output pairs; records
// and calibrations are requested one group ahead. The only workgroup barrier is the one behind the group sums of a chunk segment (2048
// events). (A first form with a workgroup per 512 events and seven barriers per sub-chunk was 10 % SLOWER than k_gather_chunks.)
// What bounds it, measured: profiles/r04_gather_bound.txt. k = 9: 984 -> 875 us, sample_limit 5000: 247 -> 218 us on one box.
#define PG_GW_SPAN 4096        // output samples per bit map (128 words)
#define PG_GW_SEG 2048         // events per segment of a chunk (32 groups of 64)
#ifndef PG_GW_TRIPS
#define PG_GW_TRIPS 4
#endif
#ifndef PG_GW_WAVES
#define PG_GW_WAVES 3
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PG_GW_WAVES, 8))) void k_gather_wave(PgDevBatch B, const uint64_t *__restrict__ n_kept_ptr, const PgKeptRec *__restrict__ rec,
        const uint64_t *__restrict__ part, uint32_t sub_per_chunk /* in units of PG_G2_SUB */, uint64_t *__restrict__ samp_off, uint64_t *__restrict__ total_out, int scaling, double pa_min, double pa_max,
        double *__restrict__ samples, const double *__restrict__ gcal, const int32_t *__restrict__ stat_flags) {
    const bool exact_div = stat_flags && stat_flags[3] != 0; // (uniform) pg_select.h: pg_div_domain_ok failed for a read of the batch
    __shared__ uint32_t gsum[PG_GW_SEG / 64];
    __shared__ uint4 s_ev_all[4][64];                 // per non-empty event of the wave's group: source index - offset inside the group (64 bits), read, -
    __shared__ double s_cal_all[4][64 * 4];           // its read's offset, scale, median, MAD
    __shared__ uint2 s_bits_all[4][PG_GW_SPAN / 32];  // x: window starts over this tile of the group's output samples, y: non-empty events in front of the word
    const uint64_t n_kept = n_kept_ptr[0];
    const uint64_t c0 = (uint64_t)blockIdx.x * sub_per_chunk * PG_G2_SUB;
    if (c0 >= n_kept) { if (n_kept == 0 && blockIdx.x == 0 && threadIdx.x == 0) { samp_off[0] = 0; total_out[0] = 0; } return; }
    const uint32_t tid = threadIdx.x, w = tid >> 6;
    const int lane = lane_id();
    uint4 *s_ev = s_ev_all[w]; double *s_cal = s_cal_all[w]; uint2 *s_bits = s_bits_all[w];
    const uint64_t c1 = c0 + (uint64_t)sub_per_chunk * PG_G2_SUB < n_kept ? c0 + (uint64_t)sub_per_chunk * PG_G2_SUB : n_kept;
    const int16_t *__restrict__ sig = B.sig;
    uint64_t run = chunk_base(part, blockIdx.x);
    for (uint64_t seg = c0; seg < c1; seg += PG_GW_SEG) {
        const uint32_t nseg = c1 - seg < PG_GW_SEG ? (uint32_t)(c1 - seg) : PG_GW_SEG;
        // ---- sums of the window lengths per group of 64 events (8 threads x 8 events each)
        if (seg != c0) __syncthreads(); // the previous segment's sums are read until its last wave has scanned them
        {
            const uint32_t *__restrict__ lens = reinterpret_cast<const uint32_t *>(rec + seg) + 2;
            uint32_t sm = 0;
#pragma unroll
            for (int i = 0; i < PG_GW_SEG / 256; ++i) { const uint32_t x = tid * (PG_GW_SEG / 256) + i; sm += x < nseg ? lens[4u * x] : 0u; }
            sm += (uint32_t)__shfl_xor((int)sm, 1, WAVE); sm += (uint32_t)__shfl_xor((int)sm, 2, WAVE); sm += (uint32_t)__shfl_xor((int)sm, 4, WAVE);
            if ((tid & 7u) == 0u) gsum[tid >> 3] = sm;
        }
        __syncthreads();
        const uint32_t gv = lane < PG_GW_SEG / 64 ? gsum[lane] : 0u, ginc = wave_incl_scan_u32(gv);
        const uint32_t segtot = (uint32_t)__builtin_amdgcn_readlane((int)ginc, WAVE - 1);
        // the records and calibrations of a group are requested one group ahead (records: two), so that the two dependent memory rounds
        // in front of a group's trips (record -> calibration of its read) run beside the previous group's trips
        auto load_rec = [&](uint32_t g) {
            const uint32_t gn = g * 64u < nseg ? (nseg - g * 64u < 64u ? nseg - g * 64u : 64u) : 0u;
            return (uint32_t)lane < gn ? reinterpret_cast<const uint4 *>(rec + seg + g * 64u)[lane] : make_uint4(0, 0, 0, 0);
        };
        // lanes 2i and 2i + 1 fetch the two halves of event i's (then event 32 + i's) 32-byte calibration record: one request per record
        auto load_cal = [&](const uint4 &qq, double2 (&c)[2]) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int evl = hh * 32 + (lane >> 1);
                const uint32_t rd = (uint32_t)__shfl((int)qq.w, evl, WAVE), ok = (uint32_t)__shfl((int)qq.z, evl, WAVE);
                c[hh] = make_double2(0.0, 0.0);
                if (ok) {
                    if (gcal) c[hh] = *reinterpret_cast<const double2 *>(gcal + 4ull * rd + 2u * (lane & 1)); // {offset, range / digitisation as the statistics used it}, {median, MAD}
                    else c[hh] = (lane & 1) ? make_double2(0.0, 1.0) : make_double2(B.off[rd], B.range[rd] / B.dig[rd]);
                }
            }
        };
        uint4 q_cur = load_rec(w), q_nxt = load_rec(w + 4);
        double2 c_cur[2];
        load_cal(q_cur, c_cur);
        for (uint32_t g = w; g * 64u < nseg; g += 4) {
            const uint64_t e0 = seg + g * 64u, gbase = run + (uint32_t)__shfl((int)(ginc - gv), (int)g, WAVE);
            const uint32_t n = nseg - g * 64u < 64u ? nseg - g * 64u : 64u;
            // ---- this group: a record per lane
            const uint4 q = q_cur;
            const double2 c0 = c_cur[0], c1 = c_cur[1];
            q_cur = q_nxt; load_cal(q_cur, c_cur); q_nxt = load_rec(g + 8);
            const uint32_t len = q.z, nzf = len != 0u;
            const uint32_t inc = wave_incl_scan_u32(len), off = inc - len, tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, WAVE - 1);
            const uint32_t idx = wave_incl_scan_u32(nzf) - nzf;
            if ((uint32_t)lane < n) samp_off[e0 + lane] = gbase + off;
            __builtin_amdgcn_wave_barrier(); // (the previous group's trips have read the stage)
            if (nzf) {
                const uint64_t srcbase = ((uint64_t)q.x | ((uint64_t)q.y << 32)) - off;
                *reinterpret_cast<uint2 *>(&s_ev[idx]) = make_uint2((uint32_t)srcbase, (uint32_t)(srcbase >> 32)); // (.z, .w: the reciprocal of the MAD, below)
            }
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int evl = hh * 32 + (lane >> 1);
                const uint32_t ok = (uint32_t)__shfl((int)nzf, evl, WAVE), ix = (uint32_t)__shfl((int)idx, evl, WAVE);
                if (ok) {
                    const double2 cc = hh ? c1 : c0;
                    *reinterpret_cast<double2 *>(s_cal + 4u * ix + 2u * (lane & 1)) = cc;
                    // the reciprocal of the read's MAD, once per event (conv below): the odd lane holds {median, MAD}
                    if (lane & 1) *reinterpret_cast<double *>(&s_ev[ix].z) = 1.0 / cc.y;
                }
            }
            uint32_t before = 0; // non-empty events that start in front of the tile
            for (uint32_t tb = 0; tb < tot; tb += PG_GW_SPAN) {
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < PG_GW_SPAN / 32 / 64; ++i) s_bits[lane + 64 * i].x = 0u;
                __builtin_amdgcn_wave_barrier();
                const uint32_t rel = off - tb; // (wraps for events in front of the tile)
                if (nzf && rel < PG_GW_SPAN) atomicOr(&s_bits[rel >> 5].x, 1u << (rel & 31u));
                __builtin_amdgcn_wave_barrier();
                constexpr int WT = PG_GW_SPAN / 32 / 64;
                uint32_t wd[WT], pc = 0;
#pragma unroll
                for (int i = 0; i < WT; ++i) { wd[i] = s_bits[lane * WT + i].x; pc += (uint32_t)__popc(wd[i]); }
                const uint32_t ipc = wave_incl_scan_u32(pc);
                uint32_t pre = before + ipc - pc;
#pragma unroll
                for (int i = 0; i < WT; ++i) { s_bits[lane * WT + i].y = pre; pre += (uint32_t)__popc(wd[i]); }
                before += (uint32_t)__builtin_amdgcn_readlane((int)ipc, WAVE - 1);
                __builtin_amdgcn_wave_barrier();
                // ---- the pairs: global output samples 2 * gp, 2 * gp + 1; the tile's first and last pair may hold one sample of a neighbour
                const uint32_t tn = tot - tb < PG_GW_SPAN ? tot - tb : PG_GW_SPAN;
                const uint32_t a = (uint32_t)((gbase + tb) & 1ull);
                const uint32_t npairs = (tn + a + 1u) >> 1;
                double *__restrict__ out = samples + gbase + tb - a; // 16-byte aligned
                // (x - median) / MAD of gmove.cpp:774 through the reciprocal y = 1 / MAD taken once per event (pg_select.h: pg_div_by_recip; the
                // correctly rounded quotient, DESIGN.md section 6: q0 faithful -> Markstein's step; q0 a second ulp off only where a / b is far
                // from every rounding midpoint). exact_div (uniform; a read of the batch has a calibration outside pg_div_domain_ok: no
                // sequencer's) or -DPG_GATHER_DIV_INSN: the division itself. FP64 is half rate here and the division was 2/3 of a sample's
                // arithmetic: 948 -> 8xx us at k = 9.
                auto conv = [&](auto EX, int raw, const double4 &c, double y) {
                    const double pA = ((double)raw + c.x) * c.y;                 // TO_PICOAMPS, poregen.h:30
                    double x = (pA < pa_min || pA > pa_max) ? 0.0 : pA;          // gmove.cpp:756-759
                    if (scaling) {
                        const double num = x - c.z;
                        if constexpr (decltype(EX)::value) x = num / c.w; else x = pg_div_by_recip(num, c.w, y);
                    }
                    return x;
                };
                // A trip's body is branch-free up to its store, so that the compiler requests the LDS reads and the window samples of all
                // PG_GW_TRIPS trips together (the first form of this loop had five dependent waits per trip and was bound by exactly that
                // chain at five waves per SIMD): out-of-range lanes work on a clamped position of the tile and drop the result.
                auto trips = [&](auto EX) { // (EX: the batch divides with the instruction -- decided once per kernel, compiled twice, no branch per sample)
                for (uint32_t j0 = lane; j0 < npairs; j0 += 64 * PG_GW_TRIPS) {
                    uint32_t qp[PG_GW_TRIPS][2]; bool v[PG_GW_TRIPS][2]; uint2 bw[PG_GW_TRIPS][2];
#pragma unroll
                    for (int u = 0; u < PG_GW_TRIPS; ++u) {
                        const uint32_t j = j0 + u * 64;
                        const uint32_t p1 = 2u * j + 1u - a, p0 = p1 - 1u; // positions inside the tile (p0 wraps for the first pair when a = 1)
                        v[u][0] = j < npairs && p0 < tn; v[u][1] = j < npairs && p1 < tn;
                        qp[u][0] = v[u][0] ? p0 : (v[u][1] ? p1 : 0u);
                        qp[u][1] = v[u][1] ? p1 : qp[u][0];
                        bw[u][0] = s_bits[qp[u][0] >> 5]; bw[u][1] = s_bits[qp[u][1] >> 5];
                    }
                    uint32_t ev[PG_GW_TRIPS][2]; uint4 ee[PG_GW_TRIPS][2];
#pragma unroll
                    for (int u = 0; u < PG_GW_TRIPS; ++u)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            ev[u][h] = bw[u][h].y + (uint32_t)__popc(bw[u][h].x & ((2u << (qp[u][h] & 31u)) - 1u)) - 1u;
                            ee[u][h] = s_ev[ev[u][h]];
                        }
                    int raw[PG_GW_TRIPS][2];
#pragma unroll
                    for (int u = 0; u < PG_GW_TRIPS; ++u)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            raw[u][h] = sig[((uint64_t)ee[u][h].x | ((uint64_t)ee[u][h].y << 32)) + tb + qp[u][h]];
                        }
#pragma unroll
                    for (int u = 0; u < PG_GW_TRIPS; ++u) {
                        const uint32_t j = j0 + u * 64;
                        const double x0 = conv(EX, raw[u][0], *reinterpret_cast<const double4 *>(s_cal + 4u * ev[u][0]), __hiloint2double((int)ee[u][0].w, (int)ee[u][0].z));
                        const double x1 = conv(EX, raw[u][1], *reinterpret_cast<const double4 *>(s_cal + 4u * ev[u][1]), __hiloint2double((int)ee[u][1].w, (int)ee[u][1].z));
                        typedef double pg_d2 __attribute__((ext_vector_type(2)));
                        if (v[u][0] && v[u][1]) { pg_d2 xx; xx.x = x0; xx.y = x1; __builtin_nontemporal_store(xx, reinterpret_cast<pg_d2 *>(out + 2u * j)); }
                        else if (v[u][0]) out[2u * j] = x0;
                        else if (v[u][1]) out[2u * j + 1u] = x1;
                    }
                } };
                if (exact_div) trips(std::true_type{}); else trips(std::false_type{});
            }
        }
        run += segtot;
    }
    if (c1 == n_kept && tid == 0) { samp_off[n_kept] = run; total_out[0] = run; } // the last chunk: all kept samples
}

#ifndef PG_GE_TRIPS
#define PG_GE_TRIPS 4
#endif
#ifndef PG_GE_WAVES
#define PG_GE_WAVES 4
#endif
// ---- the event-pair form: k_gather_wave with the lanes' pairs taken inside ONE window (pair slots), so that a lane needs one event
// look-up, one calibration and ONE 8-byte load (profiles/r04_gather_bound.txt 5, 6: scattered loads cost a cycle per lane, and a second,
// masked load inside a trip costs more than it saves). Stores are 16 bytes at 8-byte alignment, 8 bytes for the odd tail of a window.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PG_GE_WAVES, 8))) void k_gather_evpair(PgDevBatch B, const uint64_t *__restrict__ n_kept_ptr, const PgKeptRec *__restrict__ rec,
        const uint64_t *__restrict__ part, uint32_t sub_per_chunk /* in units of PG_G2_SUB */, uint64_t *__restrict__ samp_off, uint64_t *__restrict__ total_out, int scaling, double pa_min, double pa_max,
        double *__restrict__ samples, const double *__restrict__ gcal, const int32_t *__restrict__ stat_flags) {
    const bool exact_div = stat_flags && stat_flags[3] != 0; // (uniform) pg_select.h: pg_div_domain_ok failed for a read of the batch
    __shared__ uint32_t gsum[PG_GW_SEG / 64];
    __shared__ uint4 s_ev_all[4][64];                 // per non-empty event of the wave's group: source index - 2 * first pair slot (64 bits), output offset - 2 * first pair slot, 2 * first pair slot + length
    __shared__ double s_cal_all[4][64 * 6];           // its read's offset, scale, median, MAD; 1 / MAD; -
    __shared__ uint2 s_bits_all[4][PG_GW_SPAN / 64];  // x: first pair slots of the events over this tile of the group's pair slots, y: non-empty events in front of the word
    const uint64_t n_kept = n_kept_ptr[0];
    const uint64_t c0 = (uint64_t)blockIdx.x * sub_per_chunk * PG_G2_SUB;
    if (c0 >= n_kept) { if (n_kept == 0 && blockIdx.x == 0 && threadIdx.x == 0) { samp_off[0] = 0; total_out[0] = 0; } return; }
    const uint32_t tid = threadIdx.x, w = tid >> 6;
    const int lane = lane_id();
    uint4 *s_ev = s_ev_all[w]; double *s_cal = s_cal_all[w]; uint2 *s_bits = s_bits_all[w];
    const uint64_t c1 = c0 + (uint64_t)sub_per_chunk * PG_G2_SUB < n_kept ? c0 + (uint64_t)sub_per_chunk * PG_G2_SUB : n_kept;
    const int16_t *__restrict__ sig = B.sig;
    const uint32_t *__restrict__ sig32 = reinterpret_cast<const uint32_t *>(B.sig);
    const uint64_t total = B.sig_off[B.n_reads];
    uint64_t run = chunk_base(part, blockIdx.x);
    for (uint64_t seg = c0; seg < c1; seg += PG_GW_SEG) {
        const uint32_t nseg = c1 - seg < PG_GW_SEG ? (uint32_t)(c1 - seg) : PG_GW_SEG;
        // ---- sums of the window lengths per group of 64 events (8 threads x 8 events each)
        if (seg != c0) __syncthreads(); // the previous segment's sums are read until its last wave has scanned them
        {
            const uint32_t *__restrict__ lens = reinterpret_cast<const uint32_t *>(rec + seg) + 2;
            uint32_t sm = 0;
#pragma unroll
            for (int i = 0; i < PG_GW_SEG / 256; ++i) { const uint32_t x = tid * (PG_GW_SEG / 256) + i; sm += x < nseg ? lens[4u * x] : 0u; }
            sm += (uint32_t)__shfl_xor((int)sm, 1, WAVE); sm += (uint32_t)__shfl_xor((int)sm, 2, WAVE); sm += (uint32_t)__shfl_xor((int)sm, 4, WAVE);
            if ((tid & 7u) == 0u) gsum[tid >> 3] = sm;
        }
        __syncthreads();
        const uint32_t gv = lane < PG_GW_SEG / 64 ? gsum[lane] : 0u, ginc = wave_incl_scan_u32(gv);
        const uint32_t segtot = (uint32_t)__builtin_amdgcn_readlane((int)ginc, WAVE - 1);
        // the records and calibrations of a group are requested one group ahead (records: two), so that the two dependent memory rounds
        // in front of a group's trips (record -> calibration of its read) run beside the previous group's trips
        auto load_rec = [&](uint32_t g) {
            const uint32_t gn = g * 64u < nseg ? (nseg - g * 64u < 64u ? nseg - g * 64u : 64u) : 0u;
            return (uint32_t)lane < gn ? reinterpret_cast<const uint4 *>(rec + seg + g * 64u)[lane] : make_uint4(0, 0, 0, 0);
        };
        // lanes 2i and 2i + 1 fetch the two halves of event i's (then event 32 + i's) 32-byte calibration record: one request per record
        auto load_cal = [&](const uint4 &qq, double2 (&c)[2]) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int evl = hh * 32 + (lane >> 1);
                const uint32_t rd = (uint32_t)__shfl((int)qq.w, evl, WAVE), ok = (uint32_t)__shfl((int)qq.z, evl, WAVE);
                c[hh] = make_double2(0.0, 0.0);
                if (ok) {
                    if (gcal) c[hh] = *reinterpret_cast<const double2 *>(gcal + 4ull * rd + 2u * (lane & 1)); // {offset, range / digitisation as the statistics used it}, {median, MAD}
                    else c[hh] = (lane & 1) ? make_double2(0.0, 1.0) : make_double2(B.off[rd], B.range[rd] / B.dig[rd]);
                }
            }
        };
        uint4 q_cur = load_rec(w), q_nxt = load_rec(w + 4);
        double2 c_cur[2];
        load_cal(q_cur, c_cur);
        for (uint32_t g = w; g * 64u < nseg; g += 4) {
            const uint64_t e0 = seg + g * 64u, gbase = run + (uint32_t)__shfl((int)(ginc - gv), (int)g, WAVE);
            const uint32_t n = nseg - g * 64u < 64u ? nseg - g * 64u : 64u;
            // ---- this group: a record per lane
            const uint4 q = q_cur;
            const double2 c0 = c_cur[0], c1 = c_cur[1];
            q_cur = q_nxt; load_cal(q_cur, c_cur); q_nxt = load_rec(g + 8);
            const uint32_t len = q.z, nzf = len != 0u;
            const uint32_t inc = wave_incl_scan_u32(len), off = inc - len;
            const uint32_t idx = wave_incl_scan_u32(nzf) - nzf;
            const uint32_t len2 = (len + 1u) >> 1, inc2 = wave_incl_scan_u32(len2), off2 = inc2 - len2, tot2 = (uint32_t)__builtin_amdgcn_readlane((int)inc2, WAVE - 1); // pair slots: lanes own samples (2k, 2k + 1) of ONE window
            if ((uint32_t)lane < n) samp_off[e0 + lane] = gbase + off;
            __builtin_amdgcn_wave_barrier(); // (the previous group's trips have read the stage)
            if (nzf) {
                const uint64_t srcbase = ((uint64_t)q.x | ((uint64_t)q.y << 32)) - 2ull * off2;
                s_ev[idx] = make_uint4((uint32_t)srcbase, (uint32_t)(srcbase >> 32), off - 2u * off2, 2u * off2 + len);
            }
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int evl = hh * 32 + (lane >> 1);
                const uint32_t ok = (uint32_t)__shfl((int)nzf, evl, WAVE), ix = (uint32_t)__shfl((int)idx, evl, WAVE);
                if (ok) {
                    const double2 cc = hh ? c1 : c0;
                    *reinterpret_cast<double2 *>(s_cal + 6u * ix + 2u * (lane & 1)) = cc;
                    // the reciprocal of the read's MAD, once per event (conv below): the odd lane holds {median, MAD}
                    if (lane & 1) s_cal[6u * ix + 4u] = 1.0 / cc.y;
                }
            }
            uint32_t before = 0; // non-empty events that start in front of the tile
            constexpr uint32_t SPAN2 = PG_GW_SPAN / 2; // pair slots per bit map: one word per lane
            for (uint32_t tb = 0; tb < tot2; tb += SPAN2) {
                __builtin_amdgcn_wave_barrier();
                s_bits[lane].x = 0u;
                __builtin_amdgcn_wave_barrier();
                const uint32_t rel = off2 - tb; // (wraps for events in front of the tile)
                if (nzf && rel < SPAN2) atomicOr(&s_bits[rel >> 5].x, 1u << (rel & 31u));
                __builtin_amdgcn_wave_barrier();
                const uint32_t wd = s_bits[lane].x, pc = (uint32_t)__popc(wd);
                const uint32_t ipc = wave_incl_scan_u32(pc);
                s_bits[lane].y = before + ipc - pc;
                before += (uint32_t)__builtin_amdgcn_readlane((int)ipc, WAVE - 1);
                __builtin_amdgcn_wave_barrier();
                const uint32_t tn = tot2 - tb < SPAN2 ? tot2 - tb : SPAN2; // pair slots of this tile
                double *__restrict__ out = samples + gbase;
                // (x - median) / MAD of gmove.cpp:774 through the reciprocal y = 1 / MAD taken once per event (pg_select.h: pg_div_by_recip; the
                // correctly rounded quotient, DESIGN.md section 6: q0 faithful -> Markstein's step; q0 a second ulp off only where a / b is far
                // from every rounding midpoint). exact_div (uniform; a read of the batch has a calibration outside pg_div_domain_ok: no
                // sequencer's) or -DPG_GATHER_DIV_INSN: the division itself. FP64 is half rate here and the division was 2/3 of a sample's
                // arithmetic: 948 -> 8xx us at k = 9.
                auto conv = [&](auto EX, int raw, const double4 &c, double y) {
                    const double pA = ((double)raw + c.x) * c.y;                 // TO_PICOAMPS, poregen.h:30
                    double x = (pA < pa_min || pA > pa_max) ? 0.0 : pA;          // gmove.cpp:756-759
                    if (scaling) {
                        const double num = x - c.z;
                        if constexpr (decltype(EX)::value) x = num / c.w; else x = pg_div_by_recip(num, c.w, y);
                    }
                    return x;
                };
                // A trip: lane j owns pair slot tb + j = samples (2k, 2k + 1) of ONE window: one event look-up, one 8-byte load (the two dwords
                // that hold both samples whatever the parity of the source index), two conversions, one 16-byte store at 8-byte alignment
                // (8 bytes for the odd tail of a window). No pair straddles two events, so nothing in a trip branches but the store width.
                auto trips = [&](auto EX) { // (EX: the batch divides with the instruction -- decided once per kernel, compiled twice, no branch per sample)
                for (uint32_t j0 = lane; j0 < tn; j0 += 64 * PG_GE_TRIPS) {
                    uint32_t qp[PG_GE_TRIPS]; bool v[PG_GE_TRIPS]; uint2 bw[PG_GE_TRIPS];
#pragma unroll
                    for (int u = 0; u < PG_GE_TRIPS; ++u) {
                        const uint32_t j = j0 + u * 64;
                        v[u] = j < tn; qp[u] = v[u] ? j : tn - 1u;
                        bw[u] = s_bits[qp[u] >> 5];
                    }
                    uint32_t ev[PG_GE_TRIPS]; uint4 ee[PG_GE_TRIPS];
#pragma unroll
                    for (int u = 0; u < PG_GE_TRIPS; ++u) {
                        ev[u] = bw[u].y + (uint32_t)__popc(bw[u].x & ((2u << (qp[u] & 31u)) - 1u)) - 1u;
                        ee[u] = s_ev[ev[u]];
                    }
                    uint64_t i0[PG_GE_TRIPS]; bool tail = false;
#pragma unroll
                    for (int u = 0; u < PG_GE_TRIPS; ++u) {
                        i0[u] = ((uint64_t)ee[u].x | ((uint64_t)ee[u].y << 32)) + 2ull * (tb + qp[u]);
                        tail |= (i0[u] | 1ull) + 2ull >= total; // the dwords of samples i, i + 1 end behind the batch's signal
                    }
                    int raw[PG_GE_TRIPS][2];
                    if (!__ballot(tail)) { // (wave-uniform; only the wave that holds the batch's last samples takes the other side)
#pragma unroll
                        for (int u = 0; u < PG_GE_TRIPS; ++u) {
                            typedef uint32_t pg_a2 __attribute__((ext_vector_type(2), aligned(4)));
                            const pg_a2 qq = *reinterpret_cast<const pg_a2 *>(sig32 + (i0[u] >> 1));
                            const uint64_t v64 = ((uint64_t)qq.y << 32) | qq.x;
                            const uint32_t sh = (uint32_t)(i0[u] & 1u) * 16u;
                            raw[u][0] = (int)(short)(uint16_t)(v64 >> sh); raw[u][1] = (int)(short)(uint16_t)(v64 >> (sh + 16u));
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < PG_GE_TRIPS; ++u) { raw[u][0] = sig[i0[u]]; raw[u][1] = i0[u] + 1 < total ? sig[i0[u] + 1] : 0; }
                    }
#pragma unroll
                    for (int u = 0; u < PG_GE_TRIPS; ++u) {
                        const double4 cc = *reinterpret_cast<const double4 *>(s_cal + 6u * ev[u]);
                        const double y = s_cal[6u * ev[u] + 4u];
                        const double x0 = conv(EX, raw[u][0], cc, y), x1 = conv(EX, raw[u][1], cc, y);
                        const uint32_t s2 = 2u * (tb + qp[u]);                 // the pair's first sample, counted in pair slots x 2
                        double *dst = out + (uint32_t)(ee[u].z + s2);          // + (offset of the window in the group - 2 x its first pair slot)
                        typedef double pg_d2 __attribute__((ext_vector_type(2), aligned(8)));
                        if (v[u] && s2 + 1u < ee[u].w) { pg_d2 xx; xx.x = x0; xx.y = x1; __builtin_nontemporal_store(xx, reinterpret_cast<pg_d2 *>(dst)); }
                        else if (v[u]) *dst = x0; // the odd tail of a window
                    }
                } };
                if (exact_div) trips(std::true_type{}); else trips(std::false_type{});
            }
        }
        run += segtot;
    }
    if (c1 == n_kept && tid == 0) { samp_off[n_kept] = run; total_out[0] = run; } // the last chunk: all kept samples
}

// the kept events' lengths and reads as arrays of their own (pg_result / pg_device_view; off the step's path)
__global__ __launch_bounds__(256) void k_unpack_recs(const PgKeptRec *__restrict__ rec, uint64_t n, uint32_t *__restrict__ ev_len, uint32_t *__restrict__ ev_read) {
    const uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (e < n) { const PgKeptRec x = rec[e]; ev_len[e] = x.len; ev_read[e] = x.read; }
}

// =====================================================================================================
// launchers
// =====================================================================================================
hipError_t pg_launch_part_bases(hipStream_t st, const PgPartBufs &P, const PgDevBatch &B, const PgWalkOut &O) {
    PG_LAUNCH(k_part_bases, dim3(1 + (B.n_reads + 1023) / 1024), dim3(1024), 0, st, (const uint32_t *)P.totals, 1u << P.hi_bits, P.rbase, P.tile_region, P.n_tilesB,
              P.tilesB_cap, B, O, (const uint32_t *)P.Bp);
    return hipSuccess;
}

hipError_t pg_launch_part_scatter(hipStream_t st, const PgPartBufs &P, const uint32_t *ev_slot, uint64_t n, const PgDevBatch &B, const PgWalkParams &W, const PgWalkOut &O) {
    if (!n) return hipSuccess;
    const uint32_t n_tiles = (uint32_t)((n + PG_SORT_TILE - 1) / PG_SORT_TILE), n_tiles_pad = pg_tiles(n, true);
    const size_t lds = part_lds_bytes(1u << P.hi_bits);
    // more than 64 KB of dynamic LDS needs the attribute (per device: set at every launch, a host-side table entry)
    PG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_part_scatter), hipFuncAttributeMaxDynamicSharedMemorySize, (int)part_lds_bytes(PG_RANK_MAX_DIGITS)));
    PG_LAUNCH(k_part_scatter, dim3((n_tiles + PG_PART_TILES_PER_WG - 1) / PG_PART_TILES_PER_WG), dim3(PG_PART_THREADS), lds, st, ev_slot, (uint32_t)n, P.lo_bits, (int)P.hi_bits, n_tiles_pad,
              (const uint32_t *)P.hist, (const uint32_t *)P.rbase, B, W, O, (const uint32_t *)P.Bp, P.elemA, P.loA);
    return hipSuccess;
}

uint32_t pg_region_cut_epochs(void) { return PG_RCUT_EPOCHS; }
hipError_t pg_launch_region_counts(hipStream_t st, const PgPartBufs &P, uint32_t n_slots, uint64_t *acc_cnt, uint64_t *acc_copy, const PgRegionCutArgs *cut) {
    PG_LAUNCH(k_region_count, dim3(P.tilesB_cap), dim3(256), 0, st, (const uint16_t *)P.loA, (const uint32_t *)P.n_tilesB, (const uint32_t *)P.tile_region,
              (const uint32_t *)P.rbase, (const uint32_t *)P.totals, (int)P.lo_bits, P.histB);
    const uint32_t threads = (1u << P.lo_bits) < 64u ? 64u : (1u << P.lo_bits);
    if (cut && !acc_copy) { // pg_submit: the sample_limit cut rides in the scan's launch
        PgRegionCut Q{cut->running, cut->running, cut->keep, cut->ev_off, cut->totals, cut->keep32, cut->state, cut->limit, cut->epoch};
        PG_LAUNCH(k_region_scan_cut, dim3(1u << P.hi_bits), dim3(threads), 0, st, P.histB, (const uint32_t *)P.rbase, (int)P.lo_bits, n_slots, 1u << P.hi_bits, acc_cnt, Q);
        return hipSuccess;
    }
    PG_LAUNCH(k_region_scan, dim3(1u << P.hi_bits), dim3(threads), 0, st, P.histB, (const uint32_t *)P.rbase, (int)P.lo_bits, n_slots, acc_cnt, acc_copy);
    return hipSuccess;
}

static uint32_t pg_chunk_shift(uint64_t n_kept_cap) {
    uint32_t m = 1, sh = 10;
    if (n_kept_cap) (void)pg_gather_chunks(n_kept_cap, &m);
    while ((1u << sh) < m * PG_G2_SUB) ++sh; // PG_G2_SUB * m, a power of two
    return sh;
}
hipError_t pg_launch_rank_emit2(hipStream_t st, const uint32_t *ev_slot, uint64_t n, uint32_t n_slots, const uint32_t *hist, const uint64_t *keep, const uint64_t *ev_off,
                                const uint64_t *totals, const PgDevBatch &B, const PgWalkParams &W, const PgWalkOut &O, const PgKeptOut &K, const uint32_t *Bp) {
    int nbits = 1; while ((1u << nbits) < n_slots) ++nbits;
    const uint32_t n_tiles = pg_tiles(n, true);
    if (!n_tiles) return hipSuccess;
    const uint32_t groups = n_tiles / PG_EMIT2_TILES;
    PG_LAUNCH(k_rank_emit2, dim3(groups < 1024u ? groups : 1024u), dim3(PG_PART_THREADS), 0, st, ev_slot, (uint32_t)n, nbits, n_slots, n_tiles, hist, keep, ev_off, totals, B, W, O, K, Bp);
    return hipSuccess;
}

// part (n_kept_cap != 0): the chunked gather's per-chunk sums of kept window lengths, accumulated here (zeroed by pg_launch_part_tile_scan) --
// k_len_partials' pass over the records is not needed behind this launch
hipError_t pg_launch_region_place(hipStream_t st, const PgPartBufs &P, uint32_t n_slots, const uint32_t *keep32, const uint64_t *ev_off, const PgWalkOut &O, const PgKeptOut &K,
                                  uint64_t *part, uint64_t n_kept_cap) {
    const uint32_t chunk_shift = pg_chunk_shift(n_kept_cap);
    const size_t lds = part_lds_bytes(1u << P.lo_bits);
    PG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_region_place), hipFuncAttributeMaxDynamicSharedMemorySize, (int)part_lds_bytes(PG_RANK_MAX_DIGITS)));
    PG_LAUNCH(k_region_place, dim3(P.tilesB_cap), dim3(PG_PART_THREADS), lds, st, (const uint4 *)P.elemA, (const uint32_t *)P.n_tilesB, (const uint32_t *)P.tile_region,
              (const uint32_t *)P.rbase, (const uint32_t *)P.totals, (int)P.lo_bits, n_slots, (const uint32_t *)P.histB, keep32, ev_off, O, K, part, chunk_shift);
    return hipSuccess;
}

uint32_t pg_gather_chunks(uint64_t n_kept_cap, uint32_t *sub_per_chunk) {
    uint32_t m = 1;
    while ((n_kept_cap + (uint64_t)m * PG_G2_SUB - 1) / ((uint64_t)m * PG_G2_SUB) > PG_CHUNK_FINE) m *= 2; // the fine chunk sums
    *sub_per_chunk = m;
    return (uint32_t)((n_kept_cap + (uint64_t)m * PG_G2_SUB - 1) / ((uint64_t)m * PG_G2_SUB));
}

// lanes: 0 = k_gather_wave (the default); else lanes per kept event (4, 8 or 16) of k_gather_chunks: any value is correct for any window length
hipError_t pg_launch_gather_chunks(hipStream_t st, const PgDevBatch &B, uint64_t n_kept_cap, const uint64_t *n_kept_ptr, const PgKeptRec *rec, const uint64_t *part,
                                   uint64_t *samp_off, uint64_t *total_out, int scaling, double pa_min, double pa_max, double *samples, const double *gcal, int lanes, const int32_t *stat_flags) {
    if (n_kept_cap == 0) return hipSuccess;
    uint32_t m; const uint32_t n_chunks = pg_gather_chunks(n_kept_cap, &m);
    if (lanes == 1) { // the event-pair form (a lane per pair of samples of ONE window)
        PG_LAUNCH(k_gather_evpair, dim3(n_chunks), dim3(256), 0, st, B, n_kept_ptr, rec, part, m, samp_off, total_out, scaling, pa_min, pa_max, samples, gcal, stat_flags);
        return hipSuccess;
    }
    if (lanes == 0) { // the wave form: a lane per pair of output samples, a wave per 64 events
        PG_LAUNCH(k_gather_wave, dim3(n_chunks), dim3(256), 0, st, B, n_kept_ptr, rec, part, m, samp_off, total_out, scaling, pa_min, pa_max, samples, gcal, stat_flags);
        return hipSuccess;
    }
    if (lanes <= 4) PG_LAUNCH((k_gather_chunks<4, 4>), dim3(n_chunks), dim3(256), 0, st, B, n_kept_ptr, rec, part, m, samp_off, total_out, scaling, pa_min, pa_max, samples, gcal);
    else if (lanes <= 8) PG_LAUNCH((k_gather_chunks<8, 3>), dim3(n_chunks), dim3(256), 0, st, B, n_kept_ptr, rec, part, m, samp_off, total_out, scaling, pa_min, pa_max, samples, gcal);
    else PG_LAUNCH((k_gather_chunks<16, 2>), dim3(n_chunks), dim3(256), 0, st, B, n_kept_ptr, rec, part, m, samp_off, total_out, scaling, pa_min, pa_max, samples, gcal);
    return hipSuccess;
}

hipError_t pg_launch_unpack_recs(hipStream_t st, const PgKeptRec *rec, uint64_t n, uint32_t *ev_len, uint32_t *ev_read) {
    if (!n) return hipSuccess;
    PG_LAUNCH(k_unpack_recs, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, rec, n, ev_len, ev_read);
    return hipSuccess;
}
