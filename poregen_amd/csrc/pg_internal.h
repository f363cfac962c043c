// pg_internal.h -- device-side data layout and kernel launch interface of libpgmove (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---- batch as the kernels see it (all device pointers) ------------------------------------------
struct PgDevBatch {
    uint32_t n_reads;
    uint64_t n_ops;      // op_off[n_reads]
    const int16_t *sig;
    const uint64_t *sig_off;
    const double *dig, *off, *range;
    const int32_t *qstart, *tstart, *tend;
    const uint8_t *seq;
    const uint64_t *seq_off;
    const uint32_t *op_n;
    const uint8_t *op_t;
    const uint64_t *op_off;
};

struct PgWalkParams {
    uint32_t k, sig_move_offset, print_margin, max_dur, min_dur;
    int32_t pick_margin, allow_rna, short_ok;
    int32_t no_generic; // PG_BATCH_ALL_MATCHES: k_walk is not launched, a read that needs it has no events and fails the batch
    uint32_t n_codes; // 4^k
    const int32_t *table_t, *table_u; // table_u == table_t + n_codes (one allocation)
    // Table x (0: T-spelled, 1: U-spelled) is "affine" when aff_ok[x]: slot = code + aff_delta[x] for aff_lo[x] <= code <= aff_hi[x], -1
    // elsewhere (pg_create looks; an empty table has lo > hi). True for the table a generated k-mer list is spelled in and for every
    // contiguous slice of it -- the reference's default --, false for a --kmer_file in another order (and for the OTHER spelling's table,
    // which holds the T / U-free k-mers only). k_events' partitioned variant then computes the slot instead of looking it up: a scattered
    // 4-byte load costs the texture addresser a cycle per LANE (profiles/r04_gather_bound.txt 5), 16 K of them per workgroup.
    int32_t aff_ok[2]; uint32_t aff_lo[2], aff_hi[2]; int32_t aff_delta[2];
};

// per-read status codes written by k_walk_events (negative = error, reported through the C ABI)
enum {
    PGR_OK = 0,
    PGR_SKIPPED = 1,          // fetched sequence shorter than k (gmove.cpp:806-808)
    PGR_ERR_RNA = -1,         // RNA-oriented record without --rna (gmove.cpp:795-797)
    PGR_ERR_NEG = -2,         // negative query_start/target_start/target_end or query_start >= len (gmove.cpp:752)
    PGR_ERR_OP = -3,          // op type not in {0,1,2}
    PGR_ERR_SEQ_OVERRUN = -4, // ss consumes more bases than were fetched (reads past the string in the reference)
    PGR_ERR_SHORT = -5,       // fewer than k matched bases: unsigned wrap at gmove.cpp:891
    PGR_ERR_WINDOW = -6,      // an accepted event's window is empty / starts beyond the signal / margin > start
    PGR_ERR_RANGE = -7,       // sample index does not fit the reference's int arrays
    PGR_ERR_SCALE = -8,       // range/digitisation not positive finite (pg_select.h)
    PGR_ERR_WIDE = -9,        // pa window spans more than PG_STATS_BINS codes (not implemented yet)
    PGR_ERR_LAYOUT = -10      // the read's ops end behind pg_batch.n_ops, or op_off is not monotone (device batches)
};

// per-read record (64 bytes) written by k_batch_init for EVERY read; the generic walk refines n / m of the reads it walks
struct __attribute__((aligned(16))) PgReadMeta {
    uint64_t o0;     // op_off[r]
    uint64_t s0;     // seq_off[r]
    uint32_t nops;   // ss ops of the read
    uint32_t slen;   // fetched bases
    uint32_t n;      // matched bases (fastq_len after refinement, gmove.cpp:872): nops for a read of matches only, else set by the
                     // generic walk; 0 = the read has no events (skipped, failed, fewer than k matches)
    uint32_t m;      // number of I/D ops
    int32_t st_k, end_k; // min / max of the PAF target columns (gmove.cpp:792-794)
    uint32_t L;      // len_raw_signal (clamped to 2^32 - 1; > INT32_MAX is an error)
    int32_t qs;      // query_start
    uint32_t flags;  // PG_RM_*
    uint32_t opsum0; // partitioned ranking only (k_part_bases): sum of op_n over all ops in front of o0, modulo 2^32 (pg_place.hip: op_prefix)
    uint64_t sig0;   // sig_off[r]: first sample of the read in the batch's signal
};
enum {
    PG_RM_RNA = 1u,       // target_start > target_end
    PG_RM_LIVE = 2u,      // passed the per-read checks: not skipped, no error so far
    PG_RM_DIRECT_OK = 4u, // live, k <= nops <= fetched bases: takes the op-parallel event path unless one of its ops is an I or a D
    PG_RM_GENERIC = 8u    // partitioned ranking only (k_part_bases): the read is on the generic list (gen_flag[r] == batch_id), folded in here
};

struct PgWalkOut {
    uint32_t *m_start; // [n_ops] GENERIC reads: window start of match j of read r at op_off[r]+j  (end_raw_idx in the reference)
    uint32_t *m_len;   // [n_ops] GENERIC reads: window length
    uint8_t *m_base;   // [n_ops] GENERIC reads longer than the LDS window: 2-bit base code of the matched base, 4 = not ACGT/U
    uint32_t *m_tix;   // [-front .. n_ops + pad) GENERIC reads: I/D ops in front of match j of read r at op_off[r]+j = #{indel_pos
                       // entries <= j} (the interior of indel_pos, gmove.cpp:843,845, is never materialised); readable from index
                       // -kmer_pick_margin: the pointer sits PG_TIX_FRONT entries into its buffer
    uint32_t *ev_slot; // [n_ops] slot of event i of read r at op_off[r]+i, 0xFFFFFFFF = not accepted
    PgReadMeta *meta;  // [n_reads]
    const uint8_t *oor; // [n_reads] or nullptr: 1 = the read holds an out-of-range sample and the caller wants such reads skipped
    int32_t *status;   // [n_reads]
    // lowest failing read of the batch, never reset: (batch_id << 32) | (0xFFFFFFFF - read), raised with a 64-bit atomicMax, so a later
    // batch always wins and inside a batch the lowest read does; a value whose high half is not batch_id means "no error"
    unsigned long long *err;
    int32_t *layout_err; // [1] pg_batch.n_ops is not op_off[n_reads] (k_batch_init)
    // who owns an op index: blk_read[g >> 6] = read of op (g & ~63), then a short probe along op_off (pg_kernels.hip: owner_of)
    uint32_t *blk_read; // [n_ops / 64 + 1]
    // reads that need the generic (wave per read) walk: an I / D / unknown op, fewer than k ops, more ops than bases, or all live
    // reads with PG_FLAG_DEBUG_SPLIT_WALK. gen_flag[r] == batch_id marks them (no clearing between batches), gen_list holds them in
    // arrival order, gen_count[batch_id & 1] is the list length (the other entry is zeroed for the next batch)
    uint32_t *gen_flag;  // [n_reads]
    uint32_t *gen_list;  // [n_reads]
    uint32_t *gen_count; // [2]
    uint32_t batch_id;
    // op_n summed for the window starts of DIRECT reads (only ever evaluated for kept events, in the emit kernels):
    // cum[g >> 2] = sum of op_n over [g & ~255, g & ~3), btot[g >> 8] = sum over the whole 256-op block (ops < 2^24: no overflow)
    uint32_t *cum;   // [n_ops / 4 + 64]
    uint32_t *btot;  // [n_ops / 256 + 1]
    // direct ranking (<= 1024 slots): an accepted event's ev_slot entry also names its read, relative to the first read of its tile of
    // 4096 op indices: slot | (read - tile_read[tile]) << PG_SLOT_BITS, PG_REL_UNKNOWN in the upper bits = look the read up (owner_of)
    uint32_t *tile_read; // [n_tiles + 1]
};
#define PG_SLOT_BITS 10
#define PG_EV_TBL 128        // reads per tile (and its 16-op halo) k_events keeps in LDS (an event names its read relative to the first read of its tile: an entry of this table, a larger number from the scalar path, or "unknown")
#define PG_SLOT_MASK ((1u << PG_SLOT_BITS) - 1u)
#define PG_REL_UNKNOWN ((1u << (32 - PG_SLOT_BITS)) - 2u) // not all ones: slot 1023 with an unknown read must not read as PG_INVALID_SLOT

// partitioned ranking: slots of up to PG_PART_MAX_KEY_BITS (20) bits, the read's table entry above them
#define PG_PART_REL_SHIFT 20
#define PG_PART_REL_UNKNOWN ((1u << (32 - PG_PART_REL_SHIFT)) - 2u)

#define PG_INVALID_SLOT 0xFFFFFFFFu
#define PG_TIX_FRONT(margin) (((uint64_t)((margin) > 0 ? (margin) : 0) + 3) & ~3ull)

// ---- radix sort geometry ---------------------------------------------------------------------------
#define PG_SORT_ROWS 16                       // rows of 64 keys per wave
#define PG_SORT_TILE (4 * PG_SORT_ROWS * 64)  // keys per 256-thread workgroup
#define PG_RANK_MAX_BITS 10                   // digit width: 1024 LDS counters per wave
#define PG_RANK_MAX_DIGITS (1 << PG_RANK_MAX_BITS)
#define PG_DIRECT_MAX_SLOTS PG_RANK_MAX_DIGITS // up to this many slots the slot itself is the (only) digit

struct PgSortBufs {
    uint32_t *keys[2];
    uint32_t *vals[2];
    uint32_t *hist;   // [ndig][n_tiles]  (digit-major), ndig <= PG_RANK_MAX_DIGITS
    uint32_t *wcnt;   // [n_tiles][4][ndig]
    uint32_t *totals; // [ndig]
    uint32_t *dbase;  // [ndig]
    uint32_t *count;  // [2] count[0] = number of keys entering the current pass, count[1] = scratch
    uint32_t n_tiles; // for the capacity N the buffers were sized for
};

// ---- partitioned ranking (PG_DIRECT_MAX_SLOTS < n_slots <= 2^PG_PART_MAX_KEY_BITS; pg_place.hip) ------------------------------
// the chunked gather's chunk sums of kept window lengths: [0, PG_CHUNK_FINE) one per chunk, [PG_CHUNK_FINE, + PG_CHUNK_COARSE) their sums over
// groups of 64 chunks. Both are ADDED to by the kernel that places the kept events (or written by k_len_partials); a gather workgroup
// takes its base from them itself -- the coarse sums below its group + the fine sums of its group below it: ~190 values for one wave --
// so no scan launch stands between the placing kernel and the gather (round 5; k_partials_scan was 6-11 us + a kernel boundary)
#define PG_CHUNK_FINE 8192u
#define PG_CHUNK_COARSE 128u
#define PG_CHUNK_PART_N (PG_CHUNK_FINE + PG_CHUNK_COARSE + 8u)
#define PG_PART_MAX_KEY_BITS 20
struct PgPartBufs {
    uint4 *elemA;          // [n_ops + (R + 1) * PG_SORT_TILE] accepted events {slot, window start, length | start's high bits, read}, partitioned by the
                           // high digit of the slot (region), stable; region r starts at rbase[r], a multiple of PG_SORT_TILE
    uint16_t *loA;         // [as elemA] the low digit of every element once more (pass B's count kernel reads 2 bytes per element, not 16)
    uint32_t *hist;        // [R][tiles of op indices] counts (k_events<2>) -> exclusive tile prefixes (k_rank_scan)
    uint32_t *totals;      // [R] accepted events per region
    uint32_t *rbase;       // [R + 1]
    uint32_t *tile_region; // [tilesB_cap] region of every tile of elemA
    uint32_t *n_tilesB;    // [1] tiles of elemA in use
    uint32_t *histB;       // [tilesB_cap][1 << lo_bits] counts per low digit -> exclusive prefixes along the region
    uint32_t *Bp;          // [n_ops / 256 + 2] exclusive prefix of PgWalkOut::btot, modulo 2^32
    uint32_t hi_bits, lo_bits, tilesB_cap;
};
static inline uint32_t pg_part_tiles_cap(uint64_t n_ops, uint32_t hi_bits) { return (uint32_t)((n_ops + PG_SORT_TILE - 1) / PG_SORT_TILE) + (1u << hi_bits); }
hipError_t pg_launch_part_tile_scan(hipStream_t st, const PgPartBufs &P, uint64_t n_ops, const uint32_t *btot, uint64_t *zero64 /* PG_CHUNK_PART_N chunk sums, zeroed by the extra workgroup */);
hipError_t pg_launch_part_bases(hipStream_t st, const PgPartBufs &P, const PgDevBatch &B, const PgWalkOut &O);
hipError_t pg_launch_part_scatter(hipStream_t st, const PgPartBufs &P, const uint32_t *ev_slot, uint64_t n, const PgDevBatch &B, const PgWalkParams &W, const PgWalkOut &O);
// acc_cnt / acc_copy (may be null): accepted events per slot
// cut (may be null; pg_submit: base = the context's running counts): the sample_limit cut rides in the region scan's launch (k_region_scan_cut) --
// keep / keep32 / ev_off / totals as pg_launch_slot_plan leaves them, running updated in place; state: >= (2^hi_bits + 1) uint64, zeroed when
// allocated and whenever the epoch starts again at 1; epoch: the launch's serial number, 1 .. pg_region_cut_epochs()
struct PgRegionCutArgs { uint64_t *running, *keep, *ev_off, *totals; uint32_t *keep32; uint64_t *state; uint32_t limit, epoch; };
uint32_t pg_region_cut_epochs(void);
hipError_t pg_launch_region_counts(hipStream_t st, const PgPartBufs &P, uint32_t n_slots, uint64_t *acc_cnt, uint64_t *acc_copy, const PgRegionCutArgs *cut);
struct PgKeptOut; struct PgKeptRec; struct PgRareArgs;
// part / n_kept_cap: the chunked gather's per-chunk sums of kept window lengths are accumulated by this launch (n_kept_cap = 0: not wanted)
hipError_t pg_launch_region_place(hipStream_t st, const PgPartBufs &P, uint32_t n_slots, const uint32_t *keep32, const uint64_t *ev_off, const PgWalkOut &O, const PgKeptOut &K,
                                  uint64_t *part, uint64_t n_kept_cap);
// direct ranking with many useful tiles (large sample_limit): Bp = prefix of the block sums (pg_launch_rank_direct_count's extra workgroup)
hipError_t pg_launch_rank_emit2(hipStream_t st, const uint32_t *ev_slot, uint64_t n, uint32_t n_slots, const uint32_t *hist, const uint64_t *keep, const uint64_t *ev_off,
                                const uint64_t *totals, const PgDevBatch &B, const PgWalkParams &W, const PgWalkOut &O, const PgKeptOut &K, const uint32_t *Bp);
// many kept events: sample offsets inside the gather's workgroups (part: >= 8192 entries of work space)
// (part zeroed beforehand: k_rank_scan's extra workgroup). rare (may be null): the rare statistics ride in the same launch
hipError_t pg_launch_len_partials(hipStream_t st, uint64_t n_kept_cap, const uint64_t *n_kept_ptr, const PgKeptRec *rec, uint64_t *part, const PgRareArgs *rare);
uint32_t pg_gather_chunks(uint64_t n_kept_cap, uint32_t *sub_per_chunk);
// total_out: [0] = all kept samples (= samp_off[n_kept], written too), left by the workgroup of the last chunk
hipError_t pg_launch_gather_chunks(hipStream_t st, const PgDevBatch &B, uint64_t n_kept_cap, const uint64_t *n_kept_ptr, const PgKeptRec *rec, const uint64_t *part,
                                   uint64_t *samp_off, uint64_t *total_out, int scaling, double pa_min, double pa_max, double *samples, const double *gcal, int lanes,
                                   const int32_t *stat_flags /* the batch's statistics flags (may be null): [3] != 0 = the wave / event-pair forms divide instead of using the reciprocal */);
// pg_text.hip: the dump files' text on the device (flag[0] != 0 afterwards: a sample the fixed-point formatter does not take)
hipError_t pg_launch_text_lens(hipStream_t st, const double *samples, const uint64_t *samp_off, uint64_t n_events, uint32_t *tlen, uint32_t *flag);
hipError_t pg_launch_text_write(hipStream_t st, const double *samples, const uint64_t *samp_off, uint64_t n_events, const uint64_t *toff, char *text,
                                const uint64_t *ev_off, uint32_t n_slots, uint64_t *slot_toff);
hipError_t pg_launch_unpack_recs(hipStream_t st, const PgKeptRec *rec, uint64_t n, uint32_t *ev_len, uint32_t *ev_read);

// ---- stats ---------------------------------------------------------------------------------------
#define PG_STATS_BINS 2048 // in-range codes the LDS histogram can hold; wider reads take the global-memory path
#define PG_HUGE_BLOCKS 64   // workgroups (and 65600-word scratch histograms) of the global-memory path
// per-read record written by k_read_plan (pg_select.h's PgReadPlan + the read's sample range and calibration)
struct PgStatRec {
    uint64_t beg, end;   // sig_off[r], sig_off[r+1]
    int32_t c_lo, span, z0;
    int32_t mode;        // PG_STAT_*
    double offset, scale;
    double inv;          // 1.0 / scale: only ever places the candidate windows of the selection (pg_select.h), computed once per read here
    uint32_t sym;        // pg_sym_guard: the symmetric selection path may be tried (pg_select.h)
    uint32_t split;      // long read cut into slices (PgLongState): index of its first helper + 1; 0 = one wave streams the whole read
};
enum { PG_STAT_RUN = 0, PG_STAT_SKIP = 1, PG_STAT_BAD = 2 };
// ---- long reads (round 5): one wave streaming a whole read is 244 dependent passes for a 10^6-sample DNA read -- the tail of the batch.
// A read above PG_LONG_MIN samples whose in-range interval fits the 1024-bin histogram is cut into slices of PG_LONG_SLICE samples
// (doubled until at most PG_LONG_MAX_SLICES are left): slice 0 is binned by the read's own wave, the others by HELPER waves -- extra
// workgroups at the end of k_read_stats' grid, assigned through a table that the record pass fills (a reservation per long read by one
// atomic add). Every slice adds its LDS histogram to the read's histogram in global memory; the slice that finishes LAST (a counter)
// takes the sum back, leaves the memory zeroed for the next batch and runs the selection: nobody waits for anybody. The selection is
// the same code on the same counts, so the results are the one-wave path's bit for bit. A batch that wants more helpers than were
// launched is still correct: the reads that found no room stay on the one-wave path, and the count the batch wanted sizes the next launch.
#ifndef PG_LONG_MIN
#define PG_LONG_MIN 32768u
#endif
#ifndef PG_LONG_SLICE
#define PG_LONG_SLICE 16384u
#endif
#define PG_LONG_MAX_SLICES 1024u
#define PG_LONG_WORDS 1128u   // a read's histogram in global memory: the LDS layout of the 1024-bin histogram (1124 words), [1127] = slices done
#define PG_LONG_INVALID 0xFFFFFFFFu
struct PgLongState {
    uint2 *tab;        // [cap] helper h -> {read, slice}; {PG_LONG_INVALID, -} = reserved by a read that found no room
    uint32_t *hist;    // [cap][PG_LONG_WORDS], zero between batches; a long read uses the entry of its FIRST helper
    int32_t *cnt;      // this batch's counters: [0] helpers reserved (may exceed cap), [1] reads that were split
    int32_t *cnt_next; // the counters of the batch after the next one, zeroed by k_batch_init (a ring of four entries: pg_api.hip, pg_ctx::long_ring)
    uint32_t cap;      // helpers this batch may use: table entries, histograms, extra workgroups of k_read_stats
};
static inline __host__ __device__ void pg_long_geometry(uint64_t L, uint32_t *slices, uint64_t *slen) {
    uint64_t sl = PG_LONG_SLICE;
    while ((L + sl - 1) / sl > PG_LONG_MAX_SLICES) sl *= 2;
    *slen = sl; *slices = (uint32_t)((L + sl - 1) / sl);
}
#define PG_STAT_REC_BYTES 64
#define PG_HUGE_SCRATCH_WORDS ((size_t)PG_HUGE_BLOCKS * (65536 + 64))

// pg_collect_gathered: the all_gather's receive buffer uint64[world][n_slots]; all_counts == nullptr: a plain base array is used
struct PgGathered {
    const uint64_t *all_counts;
    uint32_t world, rank;
    uint64_t *total, *freq; // [n_slots] the job's accepted events per slot and min(total, sample_limit) (pg_job_totals_device)
};
// ---- launchers (all asynchronous on `st`; the result is hipGetLastError() behind the launch / the queued copy's status) -----------------------------------------------------------
// a kept event in the batch's final (k-mer-major) order: ONE 16-byte record, so that the kernels that place events -- each at a
// position of its own -- pay one memory transaction per event, not three (round 2 kept three arrays: 3 x 32-byte write requests per
// event, 1.47 GB of write traffic for 0.25 GB of data at k = 9)
struct __attribute__((aligned(16))) PgKeptRec {
    uint64_t src;  // window start as a sample index of the BATCH (sig_off[read] + start inside the read, margin applied)
    uint32_t len;  // window length incl. margin, clamped to the signal
    uint32_t read; // read index inside the batch
};
struct PgKeptOut {
    PgKeptRec *rec;       // [n_kept]
    uint8_t *read_needed; // [n_reads] set to 1 for reads that own a kept event (may be nullptr)
};
// profiling (PG_FLAG_PROFILE): the pair of events the NEXT kernel launch of this host thread carries (pg_kernels.hip: PG_LAUNCH); null = none
extern thread_local hipEvent_t pg_prof_start, pg_prof_stop;
// resets the per-batch flags of the main chain in one launch: err words, read_needed[n], and (if zero_running) the
// context's running per-slot counts
// stat_flags (may be null): the statistics flags of this batch (see pg_launch_read_plan), reset here to save a launch
hipError_t pg_launch_batch_init(hipStream_t st, uint32_t n_reads, uint8_t *read_needed, uint64_t *running, uint32_t n_slots,
                          int zero_running, int32_t *stat_flags,
                          // plan_buf (may be null): also write the statistics record of every read (what k_read_plan does, needed == null)
                          const PgDevBatch &B, double pa_min, double pa_max, void *plan_buf, int32_t *stat_status,
                          // the per-read records, the owner index and the classification of the reads (PgWalkOut)
                          const PgWalkParams &W, const PgWalkOut &O, int force_generic, const PgLongState &LS);
// tiles of PG_SORT_TILE events; in direct mode (n_slots <= PG_DIRECT_MAX_SLOTS) the count is padded to a multiple of 4:
// k_rank_count_direct handles 4 tiles per workgroup and writes their counts of a slot as one 16-byte store
static inline uint32_t pg_tiles(uint64_t n_events, bool direct) {
    const uint64_t t = (n_events + PG_SORT_TILE - 1) / PG_SORT_TILE;
    return (uint32_t)(direct ? (t + 3) & ~3ull : t);
}
// base[s] = sum over g < rank of all_counts[g * n_slots + s]
struct PgSlotModel; struct PgSlotDwell; // pg_model.h
hipError_t pg_launch_slot_model(hipStream_t st, uint32_t n_slots, const int any_kind[4], const uint64_t *ev_off, const uint64_t *samp_off,
                                const uint32_t *ev_len, const double *samples, uint32_t drop_first, PgSlotModel *out, PgSlotDwell *dwell,
                                void *scratch /* >= pg_slot_model_scratch_bytes(n_slots): the lists of the rarer kinds */);
size_t pg_slot_model_scratch_bytes(uint32_t n_slots);
// the generic walk (one wave per listed read: walk + event loop) over O.gen_list
hipError_t pg_launch_walk(hipStream_t st, const PgDevBatch &B, const PgWalkParams &W, const PgWalkOut &O);
// the op-parallel event kernel over ALL op indices: computes the events of direct reads, passes the generic reads' through, and (hist
// != null, direct ranking) counts the accepted events per tile and slot as pg_launch_rank_direct_count's first kernel would
// part_hi_bits != 0 (partitioned ranking, pg_place.hip): hist = P.hist, counts per (tile, slot >> part_lo_bits)
hipError_t pg_launch_events(hipStream_t st, const PgDevBatch &B, const PgWalkParams &W, const PgWalkOut &O, uint32_t n_slots, uint32_t *hist,
                            uint32_t part_hi_bits, uint32_t part_lo_bits);
// SAM/BAM front-end: reads with an out-of-range sample become skipped reads (behind the statistics, in front of the walk)
hipError_t pg_launch_apply_oor(hipStream_t st, const PgDevBatch &B, const PgWalkOut &O);
// direct ranking (n_slots <= PG_DIRECT_MAX_SLOTS): tile prefix of the per-tile per-slot counts pg_launch_events left in S.hist;
// acc_cnt = events per slot
hipError_t pg_launch_rank_direct_count(hipStream_t st, const uint32_t *ev_slot, uint64_t n, uint32_t n_slots, const PgSortBufs &S,
                                 uint64_t *acc_cnt, uint64_t *running, uint32_t limit, int32_t *tile_last,
                                 uint64_t *acc_copy /* device, may be null: second copy of acc_cnt (pg_count's output) */,
                                 // plan_keep != null (pg_submit: base = the context's running counts): the sample_limit cut rides in the same
                                 // launch (its last workgroup does pg_launch_slot_plan's work); *plan_done says whether it did
                                 uint64_t *plan_keep, uint64_t *plan_ev_off, uint64_t *plan_totals, uint32_t *plan_ticket, bool *plan_done,
                                 const uint32_t *btot /* PgWalkOut::btot */, uint32_t *Bp /* may be null: its prefix, for pg_launch_rank_emit2 */,
                                 uint64_t *zero64 /* with Bp: PG_CHUNK_PART_N chunk sums, zeroed by the same extra workgroup (may be null) */);
hipError_t pg_launch_rank_direct_emit(hipStream_t st, const uint32_t *ev_slot, uint64_t n, uint32_t n_slots, const PgSortBufs &S,
                                const uint64_t *keep, const uint64_t *ev_off, const uint64_t *totals, const PgDevBatch &B,
                                const PgWalkParams &W, const PgWalkOut &O, const PgKeptOut &K);
// generic ranking: stable LSD radix sort of (ev_slot, index) pairs by slot, dropping PG_INVALID_SLOT; result in
// S.keys[*sorted_idx]/S.vals[*sorted_idx], number of sorted pairs in S.count[0].
hipError_t pg_launch_sort_events(hipStream_t st, const uint32_t *ev_slot, uint64_t n, uint32_t key_bits, const PgSortBufs &S, int *sorted_idx);
hipError_t pg_launch_slot_bounds(hipStream_t st, const uint32_t *skey, const uint32_t *m_ptr, uint64_t n_upper,
                           uint32_t *slot_start, uint32_t *slot_end, uint32_t n_slots, uint64_t *acc_cnt, uint64_t *acc_copy);
// dst_scratch: uint32[n_ops] work space (the radix sort's spare value buffer)
hipError_t pg_launch_kept_meta(hipStream_t st, const uint32_t *skey, const uint32_t *sval, const uint32_t *m_ptr, uint64_t n_upper,
                         const uint32_t *slot_start, const uint64_t *keep, const uint64_t *ev_off,
                         const PgDevBatch &B, const PgWalkParams &W, const PgWalkOut &O, const PgKeptOut &K, uint32_t *dst_scratch);
// keep[s] = min(cnt[s], max(0, limit - base[s])); ev_off = exclusive scan of keep ([n_slots+1]);
// totals[0] = kept events, totals[1] = slots that are full after this batch; running[s] = base[s] + cnt[s]
// hist/n_tiles (direct mode, else nullptr): also computes totals[3] = last tile that can still place an event
// keep32 (uint32[n_slots]) + scan_scratch (as pg_launch_scan_u32_u64, >= ceil(n_slots/4096)+80 entries): work space of
// the many-slot path (n_slots > 4096, sort mode); may be null for small slot counts
hipError_t pg_launch_slot_plan(hipStream_t st, const uint64_t *acc_cnt, const uint64_t *base, uint64_t *running,
                         uint32_t limit, uint32_t n_slots, uint64_t *keep, uint64_t *ev_off, uint64_t *totals,
                         const uint32_t *hist, uint32_t n_tiles, uint32_t *keep32, uint64_t *scan_scratch,
                         const int32_t *tile_last /* direct mode, base == running: from pg_launch_rank_direct_count; else null */,
                         const PgGathered &G /* all_counts != null: base = the rows below G.rank, summed in the same pass */);
// the rare statistics (reads whose in-range interval needs more than 1024 bins): workers stride over the two lists the main launch
// has filled -- wide_list from the front (<= PG_STATS_BINS codes: LDS histogram), from the back (more: 65536 bins in global memory)
struct PgRareArgs {
    PgDevBatch B;
    const PgStatRec *plan;
    double *med, *mad, *gcal;
    int32_t *status, *err;
    int win;                   // half-width (<= 15) of the exact candidate window placed by the integer model; 0 forces the fallback search
    const uint32_t *wide_list;
    const int32_t *wide_count; // [0] / [1]: lengths of the wide / huge list
    uint32_t *scratch;         // PG_HUGE_SCRATCH_WORDS words
    uint8_t *oor;
    int range_only;
    uint32_t wide_blocks;      // hint: wide-list length of the last settled batch (64 .. 2048 workers)
};
// out[i] = sum_{j<i} in[j] for i in [0, n], n = *n_ptr <= n_cap; scratch >= ceil(n_cap/4096)+80 uint64, its first 72
// entries ZERO before the first use (every launch leaves them zero again). rare (may be null): the rare statistics ride in the
// same launch (short inputs) or get their own launch in front of the scan kernels
// total_out (may be null): the total (= out[n]) once more, where the consumer of the offsets finds it next to n
// in[i * stride]: stride in dwords (1 = a plain array; 4 = the len field of PgKeptRec records)
hipError_t pg_launch_scan_u32_u64(hipStream_t st, const uint32_t *in, uint32_t stride, uint64_t n_cap, const uint64_t *n_ptr, uint64_t *out, uint64_t *scratch,
                                  const PgRareArgs *rare, uint64_t *total_out);
hipError_t pg_launch_read_stats_rare(hipStream_t st, const PgRareArgs &A);
// the host's view of a finished batch, packed by one launch into host-mapped memory (pg_api.hip: settle_batch)
struct PgSettlePack { uint32_t errflag[6]; int32_t stat_err[6]; uint64_t n_kept, full_slots, n_samples; uint32_t cancel[2]; };
hipError_t pg_launch_settle_pack(hipStream_t st, const uint32_t *errflag, const int32_t *stat_err, const int32_t *long_cnt, const uint64_t *totals, const uint64_t *samp_off,
                                 uint64_t samp_off_entries, const uint32_t *cancel_flag, PgSettlePack *out);
// flag[0] = some slot still open below this rank (sum of rows_below rows of all_counts < limit), flag[1] = statistics cancelled; if none
// is open, the reads' statistics records (plan_buf) are set to "skip"
hipError_t pg_launch_stats_cancel_if_full(hipStream_t st, const uint64_t *all_counts, uint32_t rows_below, uint32_t n_slots, uint64_t limit, uint32_t *flag, void *plan_buf, uint32_t n_reads);
// plan_buf: one PgStatRec (64 bytes) per read: everything k_read_stats needs of a read, in one scalar load
// flags[0] = lowest failing read (reset to INT_MAX here), flags[1] = length of wide_list (reset to 0 here): reads whose
// in-range interval needs the PG_STATS_BINS histogram; stat_status[r] is reset to 0
hipError_t pg_launch_read_plan(hipStream_t st, const PgDevBatch &B, const uint8_t *read_needed, double pa_min, double pa_max, void *plan_buf,
                         int32_t *flags, int32_t *stat_status, bool flags_are_reset, const PgLongState &LS);
// the main statistics launch (one wave per read, 1024 LDS bins); reads that need more put themselves on wide_list
// gcal (may be null): double[4 * n_reads], per read {offset, scale, median, MAD} for k_gather
hipError_t pg_launch_read_stats(hipStream_t st, const PgDevBatch &B, const void *plan_buf, double *med, double *mad, int32_t *status, int32_t *err, int win,
                                uint32_t *wide_list, int32_t *wide_count, uint8_t *oor, int range_only, double *gcal, const PgLongState &LS);
// pg_finish over several batches: a batch's kept samples stay on the device until then; (k-mer, batch) segments are copied into the
// job's k-mer-major order by one launch and leave the device once
struct PgSeg { const double *src; uint64_t dst_off, n; };
hipError_t pg_launch_merge_segments(hipStream_t st, const PgSeg *d_seg, uint32_t n_seg, double *dst, uint64_t n_total /* samples of all segments: picks the kernel's shape */);
// n_kept_ptr: [0] kept events, [2] their samples (the offset scan's total_out)
hipError_t pg_launch_gather(hipStream_t st, const PgDevBatch &B, uint64_t n_kept_cap, const uint64_t *n_kept_ptr, const PgKeptRec *rec,
                      const uint64_t *samp_off, int scaling, double pa_min,
                      double pa_max, const double *med, const double *mad, double *samples, const double *gcal);
