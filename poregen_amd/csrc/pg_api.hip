// pg_api.hip -- C ABI of libpgmove (include/pgmove.h): context, buffers, launch sequence, result merge.
// Host code only; every device computation lives in pg_kernels.hip. There is no CPU fallback here:
// without a usable HIP device pg_create fails.
#include "../../include/pgmove.h"
#include "pg_internal.h"
#include "pg_hostmem.h"
#include "pg_select.h"
#include "pg_model.h"

#include <algorithm>
#include <thread>
#include <climits>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#define PG_VERSION_STR "pgmove 0.1.0 (gfx950)"

static thread_local std::string g_create_error;

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return e; }
        cap = want;
        return hipSuccess;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};


struct HostBatchResult { // one collected batch, downloaded
    uint64_t n_reads = 0, n_events = 0, n_samples = 0;
    std::vector<uint64_t> keep, ev_off;
    BigVec64 samp_off;
    BigVec32 ev_len, ev_read;
    SampleVec samples;
    std::vector<uint8_t> skipped;
    // more batches follow (or came before): the samples stay on the device, in a buffer of their own, until pg_finish merges the
    // batches there and downloads the result ONCE -- no per-batch download in front of the next submit, no second copy on the host
    DevBuf dsamples; bool on_device = false;
    bool in_ctx = false; // on_device, and still in the context's own sample buffer (pg_finish_deferred of a one-batch job): no copy at all
};

struct ProfEntry { const char *name; hipEvent_t a, b; bool bracket; };

} // namespace

struct pg_ctx {
    pg_params prm{};
    int device = 0;
    hipStream_t st = nullptr, st2 = nullptr, st3 = nullptr, own_st = nullptr; // st3: the gather of a two-stream context (created at its first use)
    hipEvent_t ev_fork = nullptr;
    hipEvent_t ev_join[2] = {nullptr, nullptr}, ev_gathered[2] = {nullptr, nullptr}; // per statistics slot
    bool slot_used[2] = {false, false};
    // the kept records, their offsets, totals and chunk sums exist twice only for the side gather (PGMOVE_GATHER_SIDE: batch i's gather beside batch
    // i + 1's chain); without it one set serves every batch (the chain of batch i + 1 follows the gather of batch i on its stream): rslot = 0
    int rslot = 0; bool side_enabled = false;
    PgSettlePack *settle_host = nullptr; // host-mapped: what settle_batch learns of a finished batch, packed by one launch (k_settle_pack)
    int slot = 0; // statistics buffers are double-buffered so that batch i+1's statistics overlap batch i's tail
    bool user_stream = false, batch_is_host = false;
    std::string err;
    uint32_t n_codes = 0, key_bits = 1;

    DevBuf table_t, table_u;
    int32_t tab_ok[2] = {0, 0}; uint32_t tab_lo[2] = {1, 1}, tab_hi[2] = {0, 0}; int32_t tab_delta[2] = {0, 0}; // PgWalkParams::aff_*
    // staged copy of a host batch
    DevBuf s_sig, s_sig_off, s_dig, s_off, s_range, s_qs, s_ts, s_te, s_seq, s_seq_off, s_op_n, s_op_t, s_op_off;
    // per-batch work buffers
    DevBuf m_start, m_len, m_base, m_tix, ev_slot, status, errflag;
    DevBuf sk[2], sv[2], hist, wcnt, totals, dbase, scount;
    DevBuf slot_start, slot_end, acc_cnt, running, keep, keep32, ev_off, plan_totals[2], base_stage, tile_last; // plan_totals, chunk_part: per statistics slot (the gather of batch i may run beside the chain of batch i + 1)
    DevBuf ev_rec[2], ev_len, ev_read, read_needed, samp_off[2], scan_scratch, samples; // ev_rec, samp_off: per statistics slot, like plan_totals // ev_len / ev_read: unpacked from the records on demand (ensure_unpacked)
    bool unpacked = false;
    uint32_t win_hint = 0; // mean kept window of the last settled batch (samples): picks the gather's lanes per event
    // partitioned ranking (1024 < slots <= 2^20; pg_place.hip)
    bool part_mode = false; uint32_t part_hi = 0, part_lo = 0;
    DevBuf part_elem, part_lodig, part_rbase, part_tile_region, part_ntiles, part_histB, part_Bp, chunk_part[2], region_state; uint32_t region_epoch = 0; // region_state / region_epoch: k_region_scan_cut
    bool gather_side = false; int gather_side_slot = 0; bool side_used[2] = {false, false}; // the last chunked gather was queued on the second stream (two-stream mode) and nothing on `st` has waited for it yet
    DevBuf med[2], mad[2], gcal[2], read_plan[2], stat_status[2], stat_err[2], wide_list[2];
    bool stat_flags_reset = false; // stat_err[slot] was reset by k_batch_init of the current batch
    // long-read counters (PgLongState::cnt), a ring of four entries of four words: batch number b uses entry b & 3 and its k_batch_init zeroes
    // entry (b + 2) & 3. Round 5 kept them beside the statistics flags of the two slots, zeroed by the batch in front -- but the reservations
    // of batch b + 1 (k_read_plan on the statistics stream, released by the gather of batch b - 1) are not ordered behind k_batch_init of
    // batch b on the chain's stream: both start when that gather ends, and a zero could wipe reservations. Two batches of distance put the
    // zeroing in front of the gather the reserving launch waits for.
    DevBuf long_ring; uint32_t long_seq = 0;
    DevBuf meta, huge_scratch, oor;
    DevBuf blk_read, gen_flag, gen_list, cum, btot, tile_read; // PgWalkOut: owner index, generic-read list, block sums of op_n
    uint32_t batch_id = 0;  // serial number of the batch being counted (tags gen_flag entries and the error word)
    PgRareArgs rare{};      // the rare statistics launch of the current batch ...
    bool rare_pending = false; // ... still to be issued: with the sample-offset scan of pg_collect
    bool in_submit = false, plan_done = false; // pg_submit: the sample_limit cut was applied inside the tile scan's launch (no k_slot_plan)
    uint32_t gen_reads = 0; // generic reads of the last settled batch
    bool batch_all_matches = false; // PG_BATCH_ALL_MATCHES of the current batch (and not PG_FLAG_DEBUG_SPLIT_WALK)
    DevBuf job_total, job_freq; bool have_job_totals = false; // pg_collect_gathered / pg_job_totals_device
    DevBuf md_ev_off, md_samp_off, md_ev_len, md_samples, md_out, md_dwell, md_class; // pg_model
    bool zero_running = false;
    bool stats_in_flight = false, totals_known = false;
    uint32_t wide_blocks = 0;    // wide-list length of the last settled batch (sizes the next rare launch)
    bool plan_in_init = false;   // this batch's statistics records were written by its k_batch_init
    bool stats_deferred = false; // PG_FLAG_DEFER_STATS: pg_count left the statistics to pg_stats / pg_collect
    DevBuf cancel_flag; bool cancel_pending = false; uint64_t stats_cancelled = 0; // pgi_stats_gathered: the device's own rank-level early-out
    // long reads (PgLongState): helper table + per-read histograms in global memory, sized for long_cap helpers; long_want = helpers the
    // next batch is expected to want (host batches: counted from sig_off; device batches: what the last settled batch wanted, at least 256)
    DevBuf long_tab, long_hist; uint32_t long_cap = 0, long_use = 0, long_want = 256; uint64_t long_reads_split = 0, long_helpers_short = 0;

    PgDevBatch B{};       // current batch (device view)
    bool have_count = false, have_batch_result = false, downloaded = true;
    int sorted_idx = 0;
    uint64_t cur_n_kept = 0, cur_n_samples = 0;
    uint64_t reads_before = 0; // reads submitted in earlier batches
    uint64_t full_slots = 0;
    bool full_before_batch = false; // every slot was full before the current batch was counted

    std::vector<HostBatchResult> batches;
    bool single_moved = false; // batches[0]'s arrays currently live in r_* (pg_finish of a one-batch job)
    bool want_samples = true;       // false inside pg_finish_deferred: the kept samples stay on the device (pg_fetch_samples)
    const double *fin_dev = nullptr; // the job's k-mer-major sample stream on the device, when the merged view does not hold it on the host
    bool merged_valid = false; // r_* hold the merged view of all downloaded batches (a repeated pg_finish / pg_model returns it as it is)
    uint64_t m_events = 0, m_samples = 0, m_reads = 0;
    // merged view
    std::vector<uint64_t> r_counts, r_ev_off;
    BigVec64 r_samp_off;
    BigVec32 r_ev_len, r_ev_read;
    SampleVec r_samples;
    DevBuf dmerged, dseg; // pg_finish over device-held batches: merged samples, segment descriptors
    DevBuf tx_samp_off, tx_ev_off, tx_len, tx_off, tx_text, tx_slot_off, tx_flag; // pg_text
    std::vector<uint64_t> tx_slot_off_host; uint64_t tx_bytes = 0;
    std::vector<uint8_t> r_skipped;
    // pg_model
    std::vector<PgSlotModel> mo_raw;
    std::vector<PgSlotDwell> mo_dw;
    std::vector<uint64_t> mo_n, mo_s2lo, mo_s2hi, mo_dn;
    std::vector<int64_t> mo_lo, mo_hi, mo_origin, mo_s1;
    std::vector<double> mo_med, mo_sd, mo_dmed;

    std::vector<ProfEntry> prof;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_pool;
    std::map<std::string, std::pair<uint64_t, double>> prof_acc;
    std::vector<std::string> prof_names;
};

static pg_status fail(pg_ctx *c, pg_status code, const char *fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define HIP_TRY(c, expr)                                                                              \
    do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail((c), PG_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } while (0)

static const char *read_status_text(int code) {
    switch (code) {
        case PGR_ERR_RNA: return "record is RNA-oriented (target_start > target_end) but allow_rna is 0 (gmove.cpp:795-797)";
        case PGR_ERR_NEG: return "negative PAF column or query_start >= len_raw_signal (assert at gmove.cpp:752)";
        case PGR_ERR_OP: return "ss op type outside {',','I','D'}";
        case PGR_ERR_SEQ_OVERRUN: return "ss consumes more bases than the fetched sequence holds (undefined in the reference, gmove.cpp:850-852)";
        case PGR_ERR_SHORT: return "fewer than k matched bases (unsigned wrap at gmove.cpp:891 in the reference)";
        case PGR_ERR_WINDOW: return "an accepted event has an empty/out-of-signal window or margin > start (undefined in the reference, gmove.cpp:928-941)";
        case PGR_ERR_RANGE: return "sample index exceeds INT32_MAX";
        case PGR_ERR_SCALE: return "range/digitisation is not a positive finite number";
        case PGR_ERR_WIDE: return "internal: statistics histogram too narrow for this read";
        case PGR_ERR_LAYOUT: return "op_off of the device batch is not monotone or ends behind pg_batch.n_ops";
        default: return "unknown";
    }
}

// ------------------------------------------------------------------------------------------------------
// profiling helpers
// bracket: two recorded events around whatever is queued in between (several launches, launches that do not go through PG_LAUNCH);
// otherwise the next kernel launch carries the pair itself
static void prof_begin(pg_ctx *c, const char *name, hipStream_t st, bool bracket = false) {
    if (!(c->prm.flags & PG_FLAG_PROFILE)) return;
    std::pair<hipEvent_t, hipEvent_t> ev;
    if (!c->prof_pool.empty()) { ev = c->prof_pool.back(); c->prof_pool.pop_back(); }
    else { (void)hipEventCreate(&ev.first); (void)hipEventCreate(&ev.second); }
    if (bracket) (void)hipEventRecord(ev.first, st);
    else { pg_prof_start = ev.first; pg_prof_stop = ev.second; } // carried by the next kernel launch (PG_LAUNCH)
    c->prof.push_back({name, ev.first, ev.second, bracket});
}
static void prof_end(pg_ctx *c, hipStream_t st) {
    if (!(c->prm.flags & PG_FLAG_PROFILE)) return;
    if (c->prof.back().bracket) { (void)hipEventRecord(c->prof.back().b, st); return; }
    if (pg_prof_start) { // nothing was launched in between: an empty interval
        (void)hipEventRecord(c->prof.back().a, st); (void)hipEventRecord(c->prof.back().b, st);
        pg_prof_start = nullptr; pg_prof_stop = nullptr;
    }
}
static void prof_drain(pg_ctx *c) { // requires the streams to be idle
    for (auto &p : c->prof) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            auto it = c->prof_acc.find(p.name);
            if (it == c->prof_acc.end()) { c->prof_acc[p.name] = {1, (double)ms}; c->prof_names.push_back(p.name); }
            else { it->second.first++; it->second.second += ms; }
        }
        c->prof_pool.push_back({p.a, p.b});
    }
    c->prof.clear();
}

// ------------------------------------------------------------------------------------------------------
extern "C" {

const char *pg_version(void) { return PG_VERSION_STR; }

const char *pg_last_error(const pg_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

void pg_default_params(pg_params *p) {
    // init_opt, src/poregen.cpp:209-237 + src/poregen.h:30-43; scaling 0 is the effective default (gmove.cpp:229)
    memset(p, 0, sizeof(*p));
    p->struct_size = sizeof(pg_params);
    p->kmer_size = 9; p->sig_move_offset = 0; p->signal_print_margin = 0; p->sample_limit = 100;
    p->max_dur = 70; p->min_dur = 5; p->kmer_pick_margin = 2; p->scaling = 0; p->allow_rna = 0;
    p->pa_min = 40.0; p->pa_max = 180.0; p->n_slots = 0; p->flags = 0; p->device = 0;
}

pg_status pg_build_slot_tables(uint32_t k, const char *const *kmers, uint32_t n_slots, int32_t *table_t, int32_t *table_u) {
    if (k < 1 || k > 13 || !kmers || !table_t || !table_u) return fail(nullptr, PG_ERR_INVALID_ARG, "pg_build_slot_tables: bad arguments (1 <= k <= 13)");
    const size_t n_codes = (size_t)1 << (2 * k);
    for (size_t i = 0; i < n_codes; i++) table_t[i] = table_u[i] = -1;
    std::map<std::string, uint32_t> seen;
    for (uint32_t s = 0; s < n_slots; s++) {
        const char *km = kmers[s];
        if (!km || strlen(km) != k) return fail(nullptr, PG_ERR_INVALID_ARG, "k-mer %u does not have length %u", s, k);
        if (!seen.emplace(km, s).second) return fail(nullptr, PG_ERR_INVALID_ARG, "duplicate k-mer %s in the slice", km);
        uint32_t code = 0; bool has_t = false, has_u = false, other = false;
        for (uint32_t i = 0; i < k; i++) {
            uint32_t b;
            switch (km[i]) {
                case 'A': b = 0; break; case 'C': b = 1; break; case 'G': b = 2; break;
                case 'T': b = 3; has_t = true; break; case 'U': b = 3; has_u = true; break;
                default: b = 0; other = true;
            }
            code = (code << 2) | b;
        }
        if (other || (has_t && has_u)) continue; // can never equal a window of the fetched sequence
        if (!has_u) table_t[code] = (int32_t)s;  // matches DNA-oriented windows (spelled with T)
        if (!has_t) table_u[code] = (int32_t)s;  // matches RNA-oriented windows (T->U applied)
    }
    return PG_OK;
}

void pg_destroy(pg_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->st) (void)hipStreamSynchronize(c->st);
    if (c->st2) (void)hipStreamSynchronize(c->st2);
    if (c->st3) (void)hipStreamSynchronize(c->st3);
    if (c->settle_host) (void)hipHostFree(c->settle_host);
    DevBuf *bufs[] = {&c->table_t, &c->table_u, &c->s_sig, &c->s_sig_off, &c->s_dig, &c->s_off, &c->s_range, &c->s_qs, &c->s_ts,
                      &c->s_te, &c->s_seq, &c->s_seq_off, &c->s_op_n, &c->s_op_t, &c->s_op_off, &c->m_start, &c->m_len, &c->m_base,
                      &c->m_tix, &c->ev_slot, &c->status, &c->errflag, &c->sk[0], &c->sk[1], &c->sv[0], &c->sv[1],
                      &c->hist, &c->wcnt, &c->totals, &c->dbase, &c->scount, &c->slot_start, &c->slot_end, &c->acc_cnt, &c->running,
                      &c->keep, &c->keep32, &c->tile_last, &c->ev_off, &c->plan_totals[0], &c->plan_totals[1], &c->base_stage, &c->dmerged, &c->dseg, &c->ev_rec[0], &c->ev_rec[1], &c->ev_len, &c->ev_read, &c->read_needed,
                      &c->tx_samp_off, &c->tx_ev_off, &c->tx_len, &c->tx_off, &c->tx_text, &c->tx_slot_off, &c->tx_flag,
                      &c->part_elem, &c->part_lodig, &c->part_rbase, &c->part_tile_region, &c->part_ntiles, &c->part_histB, &c->part_Bp, &c->chunk_part[0], &c->chunk_part[1], &c->region_state,
                      &c->samp_off[0], &c->samp_off[1], &c->cancel_flag, &c->long_tab, &c->long_hist, &c->long_ring, &c->scan_scratch, &c->samples, &c->med[0], &c->mad[0], &c->gcal[0], &c->gcal[1], &c->read_plan[0], &c->stat_status[0], &c->stat_err[0], &c->wide_list[0],
                      &c->med[1], &c->mad[1], &c->read_plan[1], &c->stat_status[1], &c->stat_err[1], &c->wide_list[1], &c->meta, &c->huge_scratch, &c->oor,
                      &c->blk_read, &c->gen_flag, &c->gen_list, &c->cum, &c->btot, &c->tile_read,
                      &c->job_total, &c->job_freq, &c->md_ev_off, &c->md_samp_off, &c->md_ev_len, &c->md_samples, &c->md_out, &c->md_dwell, &c->md_class};
    for (DevBuf *b : bufs) b->release();
    for (auto &hb : c->batches) hb.dsamples.release();
    for (auto &p : c->prof) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto &p : c->prof_pool) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (int i = 0; i < 2; i++) { if (c->ev_join[i]) (void)hipEventDestroy(c->ev_join[i]); if (c->ev_gathered[i]) (void)hipEventDestroy(c->ev_gathered[i]); }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->own_st) (void)hipStreamDestroy(c->own_st);
    if (c->st2) (void)hipStreamDestroy(c->st2);
    if (c->st3) (void)hipStreamDestroy(c->st3);
    delete c;
}

// PGMOVE_TIMING=1: wall-clock marks of the host side of a batch on stderr (where does the first batch's time go?)
static double wall_now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static bool timing_on() { static const bool on = getenv("PGMOVE_TIMING") != nullptr; return on; }
#define PG_TMARK(label) do { if (timing_on()) { const double n_ = wall_now(); fprintf(stderr, "[pgmove timing] %-34s %8.3f ms\n", label, (n_ - tmark_) * 1e3); tmark_ = n_; } } while (0)

pg_status pg_runtime_init(int32_t device) {
    double tmark_ = timing_on() ? wall_now() : 0.0;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    PG_TMARK("runtime: hipGetDeviceCount");
    if (e != hipSuccess || ndev <= 0) return fail(nullptr, PG_ERR_NO_DEVICE, "no HIP device available (%s); libpgmove has no CPU fallback", e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device < 0 || device >= ndev) return fail(nullptr, PG_ERR_NO_DEVICE, "device %d out of range (have %d)", device, ndev);
    e = hipSetDevice(device);
    if (e == hipSuccess) e = hipFree(nullptr); // creates the device's primary context
    PG_TMARK("runtime: hipSetDevice + hipFree(0)");
    if (e != hipSuccess) return fail(nullptr, PG_ERR_NO_DEVICE, "HIP runtime on device %d: %s", device, hipGetErrorString(e));
    return PG_OK;
}

pg_status pg_create(const pg_params *p, pg_ctx **out) {
    if (!p || !out) return fail(nullptr, PG_ERR_INVALID_ARG, "pg_create: null argument");
    *out = nullptr;
    if (p->struct_size != sizeof(pg_params)) return fail(nullptr, PG_ERR_INVALID_ARG, "pg_params.struct_size mismatch");
    if (p->kmer_size < 1 || p->kmer_size > 13) return fail(nullptr, PG_ERR_INVALID_ARG, "kmer_size must be in [1,13]");
    if (p->sig_move_offset > p->kmer_size) return fail(nullptr, PG_ERR_INVALID_ARG, "sig_move_offset > kmer_size indexes past the reference's arrays (gmove.cpp:892)");
    if (p->kmer_pick_margin < 0) return fail(nullptr, PG_ERR_INVALID_ARG, "negative kmer_pick_margin is undefined in the reference (gmove.cpp:204-211)");
    if (p->scaling != 0 && p->scaling != 1) return fail(nullptr, PG_ERR_INVALID_ARG, "scaling must be 0 or 1");
    if (p->n_slots < 1 || !p->table_t || !p->table_u) return fail(nullptr, PG_ERR_INVALID_ARG, "n_slots/table_t/table_u missing");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return fail(nullptr, PG_ERR_NO_DEVICE, "no HIP device available (%s); libpgmove has no CPU fallback", e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (p->device < 0 || p->device >= ndev) return fail(nullptr, PG_ERR_NO_DEVICE, "device %d out of range (have %d)", p->device, ndev);
    e = hipSetDevice(p->device);
    if (e != hipSuccess) return fail(nullptr, PG_ERR_NO_DEVICE, "hipSetDevice(%d): %s", p->device, hipGetErrorString(e));

    double tmark_ = timing_on() ? wall_now() : 0.0;
    pg_ctx *c = new pg_ctx();
    c->prm = *p;
    // two-stream mode is the default wherever it applies (include/pgmove.h: PG_FLAG_ONE_STREAM)
    if (!(p->flags & (PG_FLAG_ONE_STREAM | PG_FLAG_PROFILE | PG_FLAG_LAZY_STATS | PG_FLAG_SKIP_OUT_OF_RANGE | PG_FLAG_DEFER_STATS | PG_FLAG_OVERLAP_TAIL)) && p->scaling == 1 &&
        !getenv("PGMOVE_ONE_STREAM"))
        c->prm.flags |= PG_FLAG_OVERLAP;
    c->device = p->device;
    c->n_codes = 1u << (2 * p->kmer_size);
    c->key_bits = 1; while ((1ull << c->key_bits) < (uint64_t)p->n_slots) c->key_bits++;
    // more than 1024 slots: partitioned ranking (two digits of <= 10 bits); beyond 2^20 slots (k >= 11 with the whole list) the LSD radix
    // sort of round 1 stays (PGMOVE_LSD_SORT=1 forces it: tests, A/B)
    c->part_mode = p->n_slots > PG_DIRECT_MAX_SLOTS && c->key_bits <= PG_PART_MAX_KEY_BITS && !getenv("PGMOVE_LSD_SORT");
    if (c->part_mode) { c->part_hi = (c->key_bits + 1) / 2; c->part_lo = c->key_bits - c->part_hi; }
    auto bail = [&](pg_status s) { g_create_error = c->err; pg_destroy(c); return s; };
#define CTRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { fail(c, PG_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); return bail(PG_ERR_HIP); } } while (0)
    // The main chain is a sequence of short, latency-bound kernels; the statistics kernel is one long throughput-bound
    // launch. Give the chain the higher dispatch priority so that its workgroups get wave slots as soon as they are
    // ready instead of queueing behind the statistics kernel's 50 000 workgroups.
    int prio_low = 0, prio_high = 0;
    CTRY(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high)); // "least" and "greatest" priority (numerically high / low)
    CTRY(hipStreamCreateWithPriority(&c->own_st, hipStreamNonBlocking, prio_high));
    c->st = c->own_st;
    if (c->prm.flags & (PG_FLAG_OVERLAP | PG_FLAG_OVERLAP_TAIL)) // the second stream only exists in the modes that use it: a hardware queue costs 15-20 ms to create
    {
        // PG_FLAG_OVERLAP: the statistics stream may use three quarters of the compute units (the mask's bits go round the XCDs, so
        // every XCD keeps a quarter of its CUs free of it). Without the reservation the streaming kernel's one-wave workgroups refill
        // every wave slot they free and the chain's 16-wave workgroups wait for it to drain (measured, profiles/r02_two_stream_ab.txt:
        // 0.181 ms per step with the whole chip, 0.172 with 32 CUs withheld, 0.157 with 64, 0.159 with 96, 0.174 with 128).
        // PGMOVE_STATS_CU_WITHHELD=N overrides the number (measurements).
        const char *wh = getenv("PGMOVE_STATS_CU_WITHHELD");
        hipDeviceProp_t prop;
        CTRY(hipGetDeviceProperties(&prop, p->device));
        const int cus = prop.multiProcessorCount;
        // A job that can keep many events (n_slots x sample_limit from 2^20 up: configs[2] / [3]) spends most of a batch in the chain's
        // placing kernels and the gather; the statistics then get half of the chip: one shard of configs[2] 0.403 / 0.404 / 0.393 ->
        // 0.396 / 0.397 / 0.386 ms, configs[3] 1.373 / 1.318 / 1.346 -> 1.358 / 1.312 / 1.339 ms (three alternating runs on one box,
        // profiles/r05_probes.txt 7), while the headline (102 400 events at most) loses a fifth with that mask.
        const bool many_kept = (uint64_t)p->n_slots * p->sample_limit >= (1ull << 20);
        const int withheld = wh ? atoi(wh) : ((c->prm.flags & PG_FLAG_OVERLAP) ? (many_kept ? cus / 2 : cus / 4) : 0);
        bool masked = false;
        if (withheld > 0 && withheld < cus) {
            std::vector<uint32_t> mask((size_t)(cus + 31) / 32, 0u);
            for (int i = 0; i < cus - withheld; i++) mask[(size_t)i / 32] |= 1u << (i % 32);
            masked = hipExtStreamCreateWithCUMask(&c->st2, (uint32_t)mask.size(), mask.data()) == hipSuccess;
            if (!masked) (void)hipGetLastError(); // a runtime that refuses masks: the plain stream below
        }
        if (!masked) CTRY(hipStreamCreateWithPriority(&c->st2, hipStreamNonBlocking, prio_low));
    }
    PG_TMARK("create: streams");
    // The events that order this context's streams against each other -- never waited for by the host (it synchronises the streams). By default
    // recording an event ends the kernel in front of it with a SYSTEM-scope release (caches written back and invalidated); hipEventDisableSystemFence
    // leaves the device-scope ordering every kernel boundary has: 5.7 -> 4.6 us behind a record in the kernel trace, the step 2 % shorter.
    // (PGMOVE_EVENT_SYSTEM_FENCE=1: the default events, for A/B.)
    auto make_event = [&](hipEvent_t *ev) -> hipError_t {
        if (!getenv("PGMOVE_EVENT_SYSTEM_FENCE")) {
            if (hipEventCreateWithFlags(ev, hipEventDisableTiming | hipEventDisableSystemFence) == hipSuccess) return hipSuccess;
            (void)hipGetLastError(); // a runtime without the flag: the plain event below
        }
        return hipEventCreateWithFlags(ev, hipEventDisableTiming);
    };
    for (int i = 0; i < 2; i++) { CTRY(make_event(&c->ev_join[i])); CTRY(make_event(&c->ev_gathered[i])); }
    CTRY(make_event(&c->ev_fork));
    c->side_enabled = getenv("PGMOVE_GATHER_SIDE") != nullptr;
    CTRY(hipHostMalloc((void **)&c->settle_host, sizeof(PgSettlePack), hipHostMallocDefault));
    const size_t tb = (size_t)c->n_codes * sizeof(int32_t);
    // one allocation, the U-spelled table right behind the T-spelled one: a kernel reaches both from ONE uniform base with a 32-bit
    // lane offset (PgWalkParams::table_u == table_t + n_codes)
    CTRY(c->table_t.ensure(2 * tb));
    CTRY(hipMemcpy(c->table_t.p, p->table_t, tb, hipMemcpyHostToDevice));
    CTRY(hipMemcpy(c->table_t.as<char>() + tb, p->table_u, tb, hipMemcpyHostToDevice));
    { // is a table affine (PgWalkParams::aff_ok)?
        const int32_t *tabs[2] = {p->table_t, p->table_u};
        for (int x = 0; x < 2; x++) {
            const int32_t *t = tabs[x];
            size_t lo = 0, hi = c->n_codes;
            while (lo < c->n_codes && t[lo] < 0) ++lo;
            while (hi > lo && t[hi - 1] < 0) --hi; // [lo, hi)
            c->tab_ok[x] = 1; c->tab_lo[x] = 1; c->tab_hi[x] = 0; c->tab_delta[x] = 0;
            if (lo < hi) {
                c->tab_lo[x] = (uint32_t)lo; c->tab_hi[x] = (uint32_t)(hi - 1); c->tab_delta[x] = t[lo] - (int32_t)lo;
                for (size_t i = lo; i < hi; i++) if (t[i] != (int32_t)i + c->tab_delta[x]) { c->tab_ok[x] = 0; break; }
            }
            if (getenv("PGMOVE_NO_AFFINE")) c->tab_ok[x] = 0; // (tests, A/B: always look the slot up)
        }
    }
    c->prm.table_t = nullptr; c->prm.table_u = nullptr;
    const uint32_t ns = p->n_slots;
    CTRY(c->slot_start.ensure(ns * 4ull)); CTRY(c->slot_end.ensure(ns * 4ull));
    CTRY(c->acc_cnt.ensure(ns * 8ull)); CTRY(c->running.ensure(ns * 8ull)); CTRY(c->keep.ensure(ns * 8ull)); CTRY(c->tile_last.ensure(ns * 4ull));
    CTRY(c->ev_off.ensure((ns + 1) * 8ull)); CTRY(c->plan_totals[0].ensure(64)); CTRY(c->plan_totals[1].ensure(64)); CTRY(c->base_stage.ensure(ns * 8ull));
    CTRY(c->job_total.ensure(ns * 8ull)); CTRY(c->job_freq.ensure(ns * 8ull)); // allocated once: callers may cache the pointers
    CTRY(c->totals.ensure(256 * 4)); CTRY(c->dbase.ensure(256 * 4)); CTRY(c->scount.ensure(16)); CTRY(c->errflag.ensure(32));
    CTRY(hipMemset(c->errflag.p, 0, 32)); // [0] u64 error word, [8] i32 layout flag, [16] u32 gen_count[2] (PgWalkOut), [24] u32 ticket (k_rank_scan)
    CTRY(c->stat_err[0].ensure(32)); CTRY(c->stat_err[1].ensure(32)); // [0..2] statistics flags, [3] pg_div_domain_ok failed, [4..5] long-read counters (PgLongState::cnt)
    CTRY(hipMemset(c->stat_err[0].p, 0, 32)); CTRY(hipMemset(c->stat_err[1].p, 0, 32));
    CTRY(c->long_ring.ensure(64)); CTRY(hipMemset(c->long_ring.p, 0, 64));
    CTRY(hipMemset(c->running.p, 0, ns * 8ull));
    PG_TMARK("create: events, tables, first buffers");
#undef CTRY
    *out = c;
    return PG_OK;
}

pg_status pg_reset(pg_ctx *c) {
    if (!c) return PG_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->prm.flags & PG_FLAG_PROFILE) {
        HIP_TRY(c, hipStreamSynchronize(c->st));
        if (c->st2) HIP_TRY(c, hipStreamSynchronize(c->st2));
    if (c->st3) HIP_TRY(c, hipStreamSynchronize(c->st3));
        prof_drain(c);
    }
    // the running per-slot counts are zeroed by the next batch's init kernel (stream order is enough)
    c->zero_running = true;
    for (auto &hb : c->batches) hb.dsamples.release();
    c->batches.clear(); c->single_moved = false; c->have_job_totals = false; c->merged_valid = false; c->fin_dev = nullptr;
    c->have_count = c->have_batch_result = false; c->downloaded = true; c->totals_known = false;
    c->reads_before = 0; c->full_slots = 0; c->full_before_batch = false; c->cur_n_kept = c->cur_n_samples = 0;
    return PG_OK;
}

static pg_status settle_batch(pg_ctx *c);
static pg_status check_read_errors(pg_ctx *c, const PgSettlePack &pk);
static pg_status ensure_unpacked(pg_ctx *c);

// copy the finished batch's device results to the host (needed before its buffers are reused)
// more_coming: called in front of the next batch (pg_count); false: from pg_finish
static pg_status download_last(pg_ctx *c, bool more_coming) {
    if (!c->have_batch_result || c->downloaded) return PG_OK;
    { pg_status s0 = settle_batch(c); if (s0 != PG_OK) return s0; }
    c->batches.emplace_back();
    c->merged_valid = false;
    HostBatchResult &h = c->batches.back();
    const uint32_t ns = c->prm.n_slots;
    h.n_reads = c->B.n_reads; h.n_events = c->cur_n_kept; h.n_samples = c->cur_n_samples;
    h.keep.resize(ns); h.ev_off.resize(ns + 1); h.samp_off.resize(h.n_events + 1);
    h.ev_len.resize(h.n_events); h.ev_read.resize(h.n_events); h.samples.resize(h.n_samples);
    HIP_TRY(c, hipMemcpy(h.keep.data(), c->keep.p, ns * 8ull, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(h.ev_off.data(), c->ev_off.p, (ns + 1) * 8ull, hipMemcpyDeviceToHost));
    if (h.n_events) {
        { pg_status su = ensure_unpacked(c); if (su != PG_OK) return su; }
        HIP_TRY(c, hipMemcpy(h.samp_off.data(), c->samp_off[c->rslot].p, (h.n_events + 1) * 8ull, hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemcpy(h.ev_len.data(), c->ev_len.p, h.n_events * 4ull, hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemcpy(h.ev_read.data(), c->ev_read.p, h.n_events * 4ull, hipMemcpyDeviceToHost));
    } else h.samp_off[0] = 0;
    // one of several batches with a sizeable result: its samples stay on the device (see HostBatchResult)
    // (PGMOVE_HOST_MERGE=1: never -- A/B; PGMOVE_HOLD_MIN_BYTES=n: from n bytes on instead of 8 MB -- tests run small jobs through the device merge)
    const char *hm = getenv("PGMOVE_HOLD_MIN_BYTES");
    const uint64_t hold_min = hm ? strtoull(hm, nullptr, 10) : (8ull << 20);
    if (h.n_samples && !c->want_samples && !more_coming && c->batches.size() == 1) { // pg_finish_deferred of a one-batch job: they stay where they are
        h.on_device = true; h.in_ctx = true;
        h.samples.clear(); h.samples.shrink_to_fit();
    } else if (h.n_samples && h.n_samples * 8ull >= hold_min && (more_coming || c->batches.size() > 1) && !getenv("PGMOVE_HOST_MERGE")) {
        if (h.dsamples.ensure(h.n_samples * 8ull) == hipSuccess &&
            hipMemcpyAsync(h.dsamples.p, c->samples.p, h.n_samples * 8ull, hipMemcpyDeviceToDevice, c->st) == hipSuccess) {
            h.on_device = true;
            h.samples.clear(); h.samples.shrink_to_fit();
        } else { (void)hipGetLastError(); h.dsamples.release(); h.samples.resize(h.n_samples); } // no room: through the host
    }
    if (h.n_samples && !h.on_device) {
        // pageable destination: the runtime stages the copy through pinned bounce buffers with one host thread per call;
        // several calls on slices run side by side (first touch of the fresh pages included)
        const uint64_t bytes = h.n_samples * 8ull;
        const unsigned parts = bytes >= (64ull << 20) ? 8u : 1u;
        if (parts == 1) HIP_TRY(c, hipMemcpy(h.samples.data(), c->samples.p, bytes, hipMemcpyDeviceToHost));
        else {
            std::vector<hipError_t> rc(parts, hipSuccess);
            std::vector<std::thread> pool;
            const uint64_t step = ((h.n_samples + parts - 1) / parts + 511) & ~511ull; // samples per slice, 4 KB multiples
            for (unsigned t = 0; t < parts; t++)
                pool.emplace_back([&, t]() {
                    const uint64_t a = std::min<uint64_t>(h.n_samples, t * step), b = std::min<uint64_t>(h.n_samples, a + step);
                    if (b <= a) return;
                    rc[t] = hipSetDevice(c->device);
                    if (rc[t] == hipSuccess) rc[t] = hipMemcpy(h.samples.data() + a, c->samples.as<double>() + a, (b - a) * 8ull, hipMemcpyDeviceToHost);
                });
            for (auto &th : pool) th.join();
            for (hipError_t e : rc) HIP_TRY(c, e);
        }
    }
    std::vector<int32_t> st(h.n_reads);
    if (h.n_reads) HIP_TRY(c, hipMemcpy(st.data(), c->status.p, h.n_reads * 4ull, hipMemcpyDeviceToHost));
    h.skipped.resize(h.n_reads);
    for (uint64_t i = 0; i < h.n_reads; i++) h.skipped[i] = st[i] == PGR_SKIPPED;
    c->downloaded = true;
    return PG_OK;
}

static pg_status stage_host_batch(pg_ctx *c, const pg_batch *b) {
    const uint32_t n = b->n_reads;
    if (!b->sig_off || !b->seq_off || !b->op_off) return fail(c, PG_ERR_INVALID_ARG, "batch offsets missing");
    const uint64_t ns = b->sig_off[n], nq = b->seq_off[n], no = b->op_off[n];
    if (b->sig_off[0] != 0 || b->seq_off[0] != 0 || b->op_off[0] != 0) return fail(c, PG_ERR_INVALID_ARG, "host batch offsets must start at 0");
    uint64_t helpers = 0; // long reads of this batch (PgLongState): their helper slices, counted here (an upper bound: wide reads are not split)
    for (uint32_t r = 0; r < n; r++) {
        if (b->sig_off[r + 1] < b->sig_off[r] || b->seq_off[r + 1] < b->seq_off[r] || b->op_off[r + 1] < b->op_off[r])
            return fail(c, PG_ERR_INVALID_ARG, "batch offsets of read %u are not monotone", r);
        const uint64_t L = b->sig_off[r + 1] - b->sig_off[r];
        if (L > PG_LONG_MIN) { uint32_t S; uint64_t sl; pg_long_geometry(L, &S, &sl); helpers += S - 1; }
    }
    c->long_want = (uint32_t)std::min<uint64_t>(helpers, 1u << 22);
    struct Item { DevBuf *d; const void *src; size_t bytes; };
    Item items[] = {{&c->s_sig, b->sig, ns * 2}, {&c->s_sig_off, b->sig_off, (n + 1) * 8ull}, {&c->s_dig, b->digitisation, n * 8ull},
                    {&c->s_off, b->offset, n * 8ull}, {&c->s_range, b->range, n * 8ull}, {&c->s_qs, b->query_start, n * 4ull},
                    {&c->s_ts, b->target_start, n * 4ull}, {&c->s_te, b->target_end, n * 4ull}, {&c->s_seq, b->seq, nq},
                    {&c->s_seq_off, b->seq_off, (n + 1) * 8ull}, {&c->s_op_n, b->op_n, no * 4}, {&c->s_op_t, b->op_t, no},
                    {&c->s_op_off, b->op_off, (n + 1) * 8ull}};
    for (auto &it : items) {
        if (it.bytes && !it.src) return fail(c, PG_ERR_INVALID_ARG, "batch array missing");
        HIP_TRY(c, it.d->ensure(it.bytes + 16));
        if (it.bytes) HIP_TRY(c, hipMemcpyAsync(it.d->p, it.src, it.bytes, hipMemcpyHostToDevice, c->st));
    }
    PgDevBatch &B = c->B;
    B.n_reads = n; B.n_ops = no;
    B.sig = c->s_sig.as<int16_t>(); B.sig_off = c->s_sig_off.as<uint64_t>();
    B.dig = c->s_dig.as<double>(); B.off = c->s_off.as<double>(); B.range = c->s_range.as<double>();
    B.qstart = c->s_qs.as<int32_t>(); B.tstart = c->s_ts.as<int32_t>(); B.tend = c->s_te.as<int32_t>();
    B.seq = c->s_seq.as<uint8_t>(); B.seq_off = c->s_seq_off.as<uint64_t>();
    B.op_n = c->s_op_n.as<uint32_t>(); B.op_t = c->s_op_t.as<uint8_t>(); B.op_off = c->s_op_off.as<uint64_t>();
    return PG_OK;
}

static pg_status check_read_errors(pg_ctx *c, const PgSettlePack &pk) {
    struct { unsigned long long word; int32_t layout, pad; uint32_t gen_count[2]; } ef;
    int32_t errv[1] = {INT_MAX}, errs[6] = {INT_MAX, 0, 0, 0, 0, 0};
    static_assert(sizeof ef == sizeof pk.errflag, "PgWalkOut's error words");
    memcpy(&ef, pk.errflag, sizeof ef);
    if (ef.layout) return fail(c, PG_ERR_INVALID_ARG, "pg_batch.n_ops (%llu) is not op_off[n_reads] of the device batch", (unsigned long long)c->B.n_ops);
    if ((uint32_t)(ef.word >> 32) == c->batch_id) errv[0] = (int32_t)(0xFFFFFFFFu - (uint32_t)ef.word); // PgWalkOut::err
    c->gen_reads = ef.gen_count[c->batch_id & 1u];
    if (c->batch_all_matches && c->gen_reads) return fail(c, PG_ERR_INVALID_ARG, "pg_batch.flags says PG_BATCH_ALL_MATCHES but %u reads hold I / D / unknown ops (or fewer ops than k, or more than bases)", c->gen_reads);
    memcpy(errs, pk.stat_err, sizeof errs);
    // long reads (PgLongState): what this batch wanted sizes a device batch's next launch; a host batch is counted before it is staged
    if (errs[4] > 0 || errs[5] > 0) {
        c->long_reads_split += (uint64_t)errs[5];
        if ((uint32_t)errs[4] > c->long_use) c->long_helpers_short++;
    }
    if (!c->batch_is_host) c->long_want = std::max<uint32_t>(256u, errs[4] > 0 ? (uint32_t)errs[4] : 0u);
    // the rare statistics launch of the NEXT batch is sized by what this one needed (its blocks stride over the list: the
    // size only decides how fast a wide list is worked off; jobs without wide reads pay for 128 workgroups, not 2112)
    c->wide_blocks = errs[1] > 0 ? (uint32_t)errs[1] : 0u;
    if (c->prm.scaling != 1 && !(c->prm.flags & PG_FLAG_SKIP_OUT_OF_RANGE)) errs[0] = INT_MAX;
    if (errv[0] == INT_MAX && errs[0] == INT_MAX) return PG_OK;
    const bool walk = errv[0] <= errs[0];
    const int32_t idx = walk ? errv[0] : errs[0];
    if (c->prm.flags & PG_FLAG_STOP_WHEN_FULL) {
        // would the reference have read this line at all? It stops once every k-mer is complete (gmove.cpp:733-735)
        if (c->full_before_batch) return PG_OK; // complete before this batch: none of its reads is looked at
        const uint64_t tot[2] = {pk.n_kept, pk.full_slots};
        // every slot full and nothing kept here: the bases alone (earlier batches / lower ranks) had completed the job
        if (tot[1] == c->prm.n_slots && tot[0] == 0) return PG_OK;
        if (tot[1] == c->prm.n_slots && tot[0] > 0) { // complete inside this batch: at the read of its last kept event
            std::vector<uint32_t> er(tot[0]);
            const uint64_t kept_before = c->cur_n_kept; c->cur_n_kept = tot[0]; // (settle_batch sets it behind this check)
            const pg_status su = ensure_unpacked(c);
            c->cur_n_kept = kept_before; c->unpacked = false;
            if (su != PG_OK) return su;
            HIP_TRY(c, hipMemcpy(er.data(), c->ev_read.p, tot[0] * 4ull, hipMemcpyDeviceToHost));
            uint32_t last = 0;
            for (uint32_t v : er) last = v > last ? v : last;
            if ((uint32_t)idx > last) return PG_OK; // the lowest failing read lies behind it: never reached
        }
    }
    int32_t code = 0;
    HIP_TRY(c, hipMemcpy(&code, (walk ? c->status.as<int32_t>() : c->stat_status[c->slot].as<int32_t>()) + idx, 4, hipMemcpyDeviceToHost));
    pg_status s = code == PGR_ERR_RNA ? PG_ERR_RNA_FLAG : (code == PGR_ERR_WIDE ? PG_ERR_UNSUPPORTED : (code == PGR_ERR_LAYOUT ? PG_ERR_INVALID_ARG : PG_ERR_INPUT));
    return fail(c, s, "read %llu of the batch (global read %llu): %s", (unsigned long long)idx,
                (unsigned long long)(c->reads_before + idx), read_status_text(code));
}

// buffers of the statistics of the current batch (slot c->slot); idempotent
static pg_status ensure_stats_buffers(pg_ctx *c) {
    const uint32_t n = c->B.n_reads;
    const int sl = c->slot;
    if (c->prm.flags & PG_FLAG_SKIP_OUT_OF_RANGE) HIP_TRY(c, c->oor.ensure(n + 1ull));
    HIP_TRY(c, c->med[sl].ensure((n + 1) * 8ull)); HIP_TRY(c, c->mad[sl].ensure((n + 1) * 8ull));
    HIP_TRY(c, c->gcal[sl].ensure((n + 1) * 32ull));
    HIP_TRY(c, c->read_plan[sl].ensure((n + 1) * (size_t)PG_STAT_REC_BYTES)); HIP_TRY(c, c->wide_list[sl].ensure((n + 1) * 4ull));
    HIP_TRY(c, c->stat_status[sl].ensure((n + 2) * 4ull)); HIP_TRY(c, c->huge_scratch.ensure(PG_HUGE_SCRATCH_WORDS * 4));
    return PG_OK;
}

static pg_status fill_long(pg_ctx *c, PgLongState &LS, bool ensure);
// statistics of every read of the current batch: both LDS-histogram variants are queued back to back, each
// handles the reads whose in-range code interval fits it (no host decision, no sync)
// plan_done: the records (and the flag reset) were written by this batch's k_batch_init
static pg_status launch_stats(pg_ctx *c, hipStream_t st, const uint8_t *needed, bool forked_behind_init = false, bool plan_done = false, bool rare_with_scan = false) {
    // by this batch's k_batch_init, on the same stream (or on the stream this one was forked from, behind that kernel)
    const bool flags_are_reset = (st == c->st || forked_behind_init) && c->stat_flags_reset;
    c->stat_flags_reset = false;
    const int sl = c->slot;
    const bool skip_oor = (c->prm.flags & PG_FLAG_SKIP_OUT_OF_RANGE) != 0;
    pg_status se = ensure_stats_buffers(c);
    if (se != PG_OK) return se;
    uint8_t *oor = skip_oor ? c->oor.as<uint8_t>() : nullptr;
    const int range_only = c->prm.scaling != 1; // only the out-of-range flags are wanted
    int32_t *flags = c->stat_err[sl].as<int32_t>(); // [0] lowest failing read, [1] / [2] lengths of the wide / huge list
    PgLongState LS{};
    { pg_status sl_ = fill_long(c, LS, false); if (sl_ != PG_OK) return sl_; }
    if (!plan_done) {
        prof_begin(c, "k_read_plan", st);
        HIP_TRY(c, pg_launch_read_plan(st, c->B, needed, c->prm.pa_min, c->prm.pa_max, c->read_plan[sl].p, flags, c->stat_status[sl].as<int32_t>(), flags_are_reset, LS));
        prof_end(c, st);
    }
    const int win = (c->prm.flags & PG_FLAG_DEBUG_NARROW) ? 0 : 15;
    prof_begin(c, "k_read_stats", st);
    HIP_TRY(c, pg_launch_read_stats(st, c->B, c->read_plan[sl].p, c->med[sl].as<double>(), c->mad[sl].as<double>(),
                         c->stat_status[sl].as<int32_t>(), flags, win, c->wide_list[sl].as<uint32_t>(), flags + 1, oor, range_only,
                         c->gcal[sl].as<double>(), LS));
    prof_end(c, st);
    // the rare reads (in-range interval wider than 1024 codes; usually none): their workers ride in the launch of the sample-offset
    // scan when that comes next on the same stream (rare_pending); otherwise a launch of their own, here
    PgRareArgs &A = c->rare;
    A.B = c->B; A.plan = c->read_plan[sl].as<PgStatRec>(); A.med = c->med[sl].as<double>(); A.mad = c->mad[sl].as<double>(); A.gcal = c->gcal[sl].as<double>();
    A.status = c->stat_status[sl].as<int32_t>(); A.err = flags; A.win = win; A.wide_list = c->wide_list[sl].as<uint32_t>(); A.wide_count = flags + 1;
    A.scratch = c->huge_scratch.as<uint32_t>(); A.oor = oor; A.range_only = range_only; A.wide_blocks = c->wide_blocks;
    if (rare_with_scan && st == c->st) c->rare_pending = true;
    else {
        prof_begin(c, "k_read_stats_rare", st);
        HIP_TRY(c, pg_launch_read_stats_rare(st, A));
        prof_end(c, st);
    }
    return PG_OK;
}

static void fill_walk(pg_ctx *c, PgWalkParams &W, PgWalkOut &O) {
    W.k = c->prm.kmer_size; W.sig_move_offset = c->prm.sig_move_offset; W.print_margin = c->prm.signal_print_margin;
    W.max_dur = c->prm.max_dur; W.min_dur = c->prm.min_dur; W.pick_margin = c->prm.kmer_pick_margin; W.allow_rna = c->prm.allow_rna;
    W.short_ok = (c->prm.flags & PG_FLAG_SHORT_READS_OK) ? 1 : 0;
    W.no_generic = c->batch_all_matches ? 1 : 0;
    W.n_codes = c->n_codes; W.table_t = c->table_t.as<int32_t>(); W.table_u = c->table_t.as<int32_t>() + c->n_codes;
    for (int x = 0; x < 2; x++) { W.aff_ok[x] = c->tab_ok[x]; W.aff_lo[x] = c->tab_lo[x]; W.aff_hi[x] = c->tab_hi[x]; W.aff_delta[x] = c->tab_delta[x]; }
    O.m_start = c->m_start.as<uint32_t>(); O.m_len = c->m_len.as<uint32_t>(); O.m_base = c->m_base.as<uint8_t>();
    O.m_tix = c->m_tix.as<uint32_t>() + PG_TIX_FRONT(c->prm.kmer_pick_margin); O.ev_slot = c->ev_slot.as<uint32_t>();
    O.meta = c->meta.as<PgReadMeta>();
    O.status = c->status.as<int32_t>();
    O.err = c->errflag.as<unsigned long long>(); O.layout_err = c->errflag.as<int32_t>() + 2; O.gen_count = c->errflag.as<uint32_t>() + 4;
    O.blk_read = c->blk_read.as<uint32_t>(); O.gen_flag = c->gen_flag.as<uint32_t>(); O.gen_list = c->gen_list.as<uint32_t>();
    O.cum = c->cum.as<uint32_t>(); O.btot = c->btot.as<uint32_t>(); O.batch_id = c->batch_id; O.tile_read = c->tile_read.as<uint32_t>();
    O.oor = (c->prm.flags & PG_FLAG_SKIP_OUT_OF_RANGE) ? c->oor.as<uint8_t>() : nullptr;
}

// the long-read split of the statistics (PgLongState) for the batch in statistics slot c->slot; ensure = size the buffers for this batch
static pg_status fill_long(pg_ctx *c, PgLongState &LS, bool ensure) {
    LS = PgLongState{};
    if (getenv("PGMOVE_NO_LONG_SPLIT")) return PG_OK; // (tests, A/B: every read by one wave)
    if (ensure) {
        const char *ov = getenv("PGMOVE_LONG_HELPERS"); // (tests: a cap below what the batch wants)
        uint32_t want = ov ? (uint32_t)strtoul(ov, nullptr, 10) : c->long_want;
        if (want > (1u << 22)) want = 1u << 22;
        if (want > c->long_cap || !c->long_tab.p) {
            const uint32_t cap = want + want / 4 + 64;
            HIP_TRY(c, c->long_tab.ensure((size_t)cap * 8));
            const size_t before = c->long_hist.cap;
            HIP_TRY(c, c->long_hist.ensure((size_t)cap * PG_LONG_WORDS * 4));
            if (c->long_hist.cap != before) {
                // zero between batches: the last slice of a read leaves it so. The statistics that add into it may run on the second
                // stream, which is not ordered behind this stream's work (it waits for the staging copies at most): the fill must have
                // FINISHED before anything of this batch is queued there. Growth is rare (the first long batch, or one that wants more
                // helpers than any before), so the host waits for it (advisor r05: a fill of cap x 4512 bytes raced the first slices' adds).
                HIP_TRY(c, hipMemsetAsync(c->long_hist.p, 0, c->long_hist.cap, c->st));
                HIP_TRY(c, hipStreamSynchronize(c->st));
            }
            c->long_cap = cap;
        }
        // this batch's helpers: what it is expected to want plus a margin (a batch without long reads launches 64 helper workgroups, not the
        // thousands an earlier batch needed); PGMOVE_LONG_HELPERS: exactly that many
        c->long_use = ov ? std::min<uint32_t>(c->long_cap, want) : std::min<uint32_t>(c->long_cap, want + want / 4 + 64);
    }
    LS.tab = c->long_tab.as<uint2>(); LS.hist = c->long_hist.as<uint32_t>(); LS.cap = c->long_use;
    LS.cnt = c->long_ring.as<int32_t>() + 4u * (c->long_seq & 3u); LS.cnt_next = c->long_ring.as<int32_t>() + 4u * ((c->long_seq + 2u) & 3u);
    return PG_OK;
}

static void fill_sort(pg_ctx *c, PgSortBufs &S, uint32_t n_tiles) {
    S.keys[0] = c->sk[0].as<uint32_t>(); S.keys[1] = c->sk[1].as<uint32_t>();
    S.vals[0] = c->sv[0].as<uint32_t>(); S.vals[1] = c->sv[1].as<uint32_t>();
    S.hist = c->hist.as<uint32_t>(); S.wcnt = c->wcnt.as<uint32_t>(); S.totals = c->totals.as<uint32_t>();
    S.dbase = c->dbase.as<uint32_t>(); S.count = c->scount.as<uint32_t>(); S.n_tiles = n_tiles;
}

// which gather for many kept events: by the mean kept window of the last settled batch, else by a guess from the duration filter (dwells
// lean towards min_dur). 0 = k_gather_wave (a lane per pair of OUTPUT samples: short windows, k = 9's mean of 12), 1 = k_gather_evpair (a
// lane per pair of samples of ONE window: one load per lane, but a lane-slot lost per odd window -- pays from ~16 samples up: sample_limit
// 5000's mean of 28: 218 -> 197 us; k = 9: 837 -> 904). Every choice is correct for every window; PGMOVE_GATHER_LANES overrides
// (measurements: 4 / 8 / 16 = round 3's k_gather_chunks with that many lanes per event).
static int gather_lanes(const pg_ctx *c) {
    const char *ov = getenv("PGMOVE_GATHER_LANES"); // read per collect: a test can switch the form between two jobs of one process
    if (ov) return atoi(ov);
    const uint32_t mw = c->win_hint ? c->win_hint : c->prm.min_dur + (c->prm.max_dur > c->prm.min_dur ? (c->prm.max_dur - c->prm.min_dur) / 8 : 0) + 2 * c->prm.signal_print_margin;
    return mw > 16 ? 1 : 0;
}

// direct ranking: will (nearly) every tile place events? Then the placing kernel built for that takes over (k_rank_emit2), and it wants
// the prefix of the block sums. The bound is the same as the chunked gather's: more than 64 * 4096 events may be kept.
// (PGMOVE_DENSE_MIN=n lowers the bound: tests and the fuzzers run small jobs through the kernels built for many kept events)
static uint64_t dense_min() { const char *e = getenv("PGMOVE_DENSE_MIN"); return e ? strtoull(e, nullptr, 10) : 64ull * 4096; }
static bool dense_direct(const pg_ctx *c, uint64_t n_ops) {
    return c->prm.n_slots <= PG_DIRECT_MAX_SLOTS && std::min<uint64_t>(n_ops, (uint64_t)c->prm.n_slots * c->prm.sample_limit) > dense_min() && !getenv("PGMOVE_EMIT1");
}

static void fill_part(pg_ctx *c, PgPartBufs &P, uint64_t n_ops) {
    P.elemA = c->part_elem.as<uint4>(); P.loA = c->part_lodig.as<uint16_t>(); P.hist = c->hist.as<uint32_t>(); P.totals = c->totals.as<uint32_t>(); P.rbase = c->part_rbase.as<uint32_t>();
    P.tile_region = c->part_tile_region.as<uint32_t>(); P.n_tilesB = c->part_ntiles.as<uint32_t>(); P.histB = c->part_histB.as<uint32_t>();
    P.Bp = c->part_Bp.as<uint32_t>(); P.hi_bits = c->part_hi; P.lo_bits = c->part_lo; P.tilesB_cap = pg_part_tiles_cap(n_ops, c->part_hi);
}

// the kept events' lengths and reads as arrays of their own (the device holds 16-byte records): for the download, the device view, the model
static pg_status ensure_unpacked(pg_ctx *c) {
    if (c->unpacked) return PG_OK;
    HIP_TRY(c, c->ev_len.ensure((c->cur_n_kept + 1) * 4)); HIP_TRY(c, c->ev_read.ensure((c->cur_n_kept + 1) * 4));
    HIP_TRY(c, pg_launch_unpack_recs(c->st, c->ev_rec[c->rslot].as<PgKeptRec>(), c->cur_n_kept, c->ev_len.as<uint32_t>(), c->ev_read.as<uint32_t>()));
    HIP_TRY(c, hipStreamSynchronize(c->st));
    c->unpacked = true;
    return PG_OK;
}

pg_status pg_count(pg_ctx *c, const pg_batch *b, uint64_t *counts_out, int32_t counts_location) {
    if (!c || !b) return PG_ERR_INVALID_ARG;
    if (b->struct_size != sizeof(pg_batch)) return fail(c, PG_ERR_INVALID_ARG, "pg_batch.struct_size mismatch");
    HIP_TRY(c, hipSetDevice(c->device));
    pg_status s = download_last(c, true);
    if (s != PG_OK) return s;
    for (auto &hb : c->batches)
        if (hb.in_ctx) { // a deferred finish left this batch's samples in the buffer the next collect writes: park them first
            HIP_TRY(c, hb.dsamples.ensure(hb.n_samples * 8ull));
            HIP_TRY(c, hipMemcpyAsync(hb.dsamples.p, c->samples.p, hb.n_samples * 8ull, hipMemcpyDeviceToDevice, c->st));
            hb.in_ctx = false; c->merged_valid = false; c->fin_dev = nullptr;
        }
    if (c->have_batch_result) { c->reads_before += c->B.n_reads; c->have_batch_result = false; }
    c->have_count = false; c->rare_pending = false; c->plan_done = false;
    const uint32_t n = b->n_reads;

    double tmark_ = timing_on() ? wall_now() : 0.0;
    c->batch_is_host = b->location == PG_LOC_HOST;
    c->batch_all_matches = (b->flags & PG_BATCH_ALL_MATCHES) != 0 && !(c->prm.flags & PG_FLAG_DEBUG_SPLIT_WALK);
    if (b->location == PG_LOC_HOST) {
        if (c->gather_side) { // the staging buffers still hold the batch whose gather runs on the second stream
            HIP_TRY(c, hipStreamWaitEvent(c->st, c->ev_gathered[c->gather_side_slot], 0));
            c->gather_side = false;
        }
        s = stage_host_batch(c, b);
        if (s != PG_OK) return s;
        PG_TMARK("count: staging copies queued");
        if (timing_on()) { HIP_TRY(c, hipStreamSynchronize(c->st)); PG_TMARK("count: staging copies done (sync)"); }
    } else if (b->location == PG_LOC_DEVICE) {
        PgDevBatch &B = c->B;
        B.sig = b->sig; B.sig_off = b->sig_off; B.dig = b->digitisation; B.off = b->offset; B.range = b->range;
        B.qstart = b->query_start; B.tstart = b->target_start; B.tend = b->target_end; B.seq = b->seq; B.seq_off = b->seq_off;
        B.op_n = b->op_n; B.op_t = b->op_t; B.op_off = b->op_off;
        if (!B.sig_off || !B.seq_off || !B.op_off) return fail(c, PG_ERR_INVALID_ARG, "batch offsets missing");
        // op_off[n]: from the caller (pg_batch.n_ops; verified on the device by k_batch_init, and k_walk refuses reads whose ops end
        // behind it), else read back on the context's stream -- never remembered from an earlier batch: an allocator hands the
        // same address to the next batch of the same shape
        uint64_t no = b->n_ops;
        if (no == 0 && n > 0) {
            HIP_TRY(c, hipMemcpyAsync(&no, b->op_off + n, 8, hipMemcpyDeviceToHost, c->st));
            HIP_TRY(c, hipStreamSynchronize(c->st));
        }
        B.n_reads = n;
        B.n_ops = no;
    } else return fail(c, PG_ERR_INVALID_ARG, "pg_batch.location must be PG_LOC_HOST or PG_LOC_DEVICE");
    if (((uintptr_t)c->B.sig & 15) != 0) return fail(c, PG_ERR_INVALID_ARG, "sig must be 16-byte aligned");
    if (((uintptr_t)c->B.seq & 3) != 0 || ((uintptr_t)c->B.op_n & 15) != 0) return fail(c, PG_ERR_INVALID_ARG, "seq must be 4-byte aligned and op_n 16-byte aligned");
    const uint64_t N = c->B.n_ops;
    if (N >= 0x7fffffffull) return fail(c, PG_ERR_INVALID_ARG, "more than 2^31 ss ops in one batch; split the batch");

    // work buffers
    const uint64_t Nn = N ? N : 1;
    const bool direct = c->prm.n_slots <= PG_DIRECT_MAX_SLOTS;
    // +32 entries: k_events reads/writes these per-op arrays with 16-byte vectors that may overrun n_ops
    HIP_TRY(c, c->m_start.ensure((Nn + 32) * 4)); HIP_TRY(c, c->m_len.ensure((Nn + 32) * 4)); HIP_TRY(c, c->m_base.ensure(Nn + 32));
    HIP_TRY(c, c->m_tix.ensure((Nn + 64 + c->prm.kmer_size + 2 * PG_TIX_FRONT(c->prm.kmer_pick_margin)) * 4));
    HIP_TRY(c, c->ev_slot.ensure((Nn + 32) * 4));
    HIP_TRY(c, c->blk_read.ensure((Nn / 64 + 2) * 4)); HIP_TRY(c, c->cum.ensure((Nn / 4 + 64) * 4)); HIP_TRY(c, c->btot.ensure((Nn / 256 + 2) * 4));
    HIP_TRY(c, c->gen_list.ensure((n + 1) * 4ull)); HIP_TRY(c, c->tile_read.ensure((Nn / PG_SORT_TILE + 8) * 4));
    { // gen_flag entries are compared with the batch's serial number: fresh memory must not hold one by accident
        const size_t before = c->gen_flag.cap;
        HIP_TRY(c, c->gen_flag.ensure((n + 1) * 4ull));
        if (c->gen_flag.cap != before) HIP_TRY(c, hipMemsetAsync(c->gen_flag.p, 0, c->gen_flag.cap, c->st));
    }
    if (++c->batch_id == 0) { c->batch_id = 1; HIP_TRY(c, hipMemsetAsync(c->gen_flag.p, 0, c->gen_flag.cap, c->st)); }
    HIP_TRY(c, c->meta.ensure((n + 1) * sizeof(PgReadMeta))); HIP_TRY(c, c->status.ensure((n + 1) * 4ull));
    HIP_TRY(c, c->read_needed.ensure(n + 2ull));
    const uint32_t n_tiles = pg_tiles(Nn, direct);
    uint32_t ndig;
    if (direct) {
        ndig = 2; while (ndig < c->prm.n_slots) ndig <<= 1;
        if (dense_direct(c, N)) { HIP_TRY(c, c->part_Bp.ensure((Nn / 256 + 4) * 4)); HIP_TRY(c, c->chunk_part[0].ensure(PG_CHUNK_PART_N * 8)); if (c->side_enabled) HIP_TRY(c, c->chunk_part[1].ensure(PG_CHUNK_PART_N * 8)); }
    }
    else if (c->part_mode) {
        ndig = 1u << c->part_hi;
        const uint32_t tcap = pg_part_tiles_cap(Nn, c->part_hi);
        HIP_TRY(c, c->part_elem.ensure((size_t)tcap * PG_SORT_TILE * 16)); HIP_TRY(c, c->part_lodig.ensure((size_t)tcap * PG_SORT_TILE * 2)); HIP_TRY(c, c->part_rbase.ensure((ndig + 2) * 4ull));
        HIP_TRY(c, c->part_tile_region.ensure((tcap + 1) * 4ull)); HIP_TRY(c, c->part_ntiles.ensure(16));
        HIP_TRY(c, c->part_histB.ensure(((size_t)tcap << c->part_lo) * 4)); HIP_TRY(c, c->part_Bp.ensure((Nn / 256 + 4) * 4));
        HIP_TRY(c, c->chunk_part[0].ensure(PG_CHUNK_PART_N * 8)); if (c->side_enabled) HIP_TRY(c, c->chunk_part[1].ensure(PG_CHUNK_PART_N * 8));
    } else {
        const uint32_t passes = (c->key_bits + PG_RANK_MAX_BITS - 1) / PG_RANK_MAX_BITS;
        ndig = 1u << ((c->key_bits + passes - 1) / passes);
        for (int i = 0; i < 2; i++) { HIP_TRY(c, c->sk[i].ensure(Nn * 4)); HIP_TRY(c, c->sv[i].ensure(Nn * 4)); }
    }
    const uint32_t n_tiles_hist = c->part_mode ? pg_tiles(Nn, true) : n_tiles; // k_events writes a slot's counts of four tiles as one 16-byte store
    HIP_TRY(c, c->hist.ensure((size_t)n_tiles_hist * ndig * 4));
    if (!direct && !c->part_mode) HIP_TRY(c, c->wcnt.ensure((size_t)n_tiles * ndig * 16)); // per-wave counts: only the radix sort keeps them
    HIP_TRY(c, c->totals.ensure(ndig * 4ull)); HIP_TRY(c, c->dbase.ensure(ndig * 4ull));

    PG_TMARK("count: work buffers");
    c->full_before_batch = c->full_slots == c->prm.n_slots && c->prm.n_slots > 0;
    c->slot ^= 1; // this batch's statistics buffers
    c->rslot = c->side_enabled ? c->slot : 0;
    const bool skip_oor = (c->prm.flags & PG_FLAG_SKIP_OUT_OF_RANGE) != 0; // statistics first: the walk needs their verdict
    const bool eager_stats = skip_oor || (c->prm.scaling == 1 && !(c->prm.flags & PG_FLAG_LAZY_STATS));
    const bool overlap = !skip_oor && (c->prm.flags & PG_FLAG_OVERLAP) != 0;
    if (eager_stats && overlap) {
        // The statistics stream only needs the batch to be resident: after the staging copies of a host batch, or
        // after the producer of a device batch when the caller drives us on its own stream. A device batch on the
        // context's own stream must be complete when pg_count is called, so nothing on `st` has to be awaited and
        // the statistics of this batch overlap the tail (plan/emit/scan/gather) of the previous one.
        if (c->batch_is_host || (c->user_stream && !(b->flags & PG_BATCH_RESIDENT))) { // (PG_BATCH_RESIDENT: the caller vouches that nothing on its stream produces the batch)
            HIP_TRY(c, hipEventRecord(c->ev_fork, c->st));
            HIP_TRY(c, hipStreamWaitEvent(c->st2, c->ev_fork, 0));
        }
        // ... and the buffers of this slot to be free: the gather of the batch that used them two batches ago
        if (c->slot_used[c->slot]) HIP_TRY(c, hipStreamWaitEvent(c->st2, c->ev_gathered[c->slot], 0));
    }
    // eager statistics on the main stream (or forked from it later): k_batch_init also writes the statistics record of every
    // read -- k_read_plan's work, one kernel boundary less. The second-stream mode plans on its own stream; lazy statistics
    // plan behind the emit kernel (they need its flags).
    c->plan_in_init = eager_stats && !overlap && n > 0;
    if (c->plan_in_init) { pg_status se = ensure_stats_buffers(c); if (se != PG_OK) return se; }
    PgWalkParams W{}; PgWalkOut O{};
    fill_walk(c, W, O);
    const bool force_generic = (c->prm.flags & PG_FLAG_DEBUG_SPLIT_WALK) != 0;
    PgLongState LS{};
    ++c->long_seq; // this batch's entry of the long-read counter ring
    { pg_status sl_ = fill_long(c, LS, true); if (sl_ != PG_OK) return sl_; }
    prof_begin(c, "k_batch_init", c->st);
    HIP_TRY(c, pg_launch_batch_init(c->st, n, c->read_needed.as<uint8_t>(), c->running.as<uint64_t>(), c->prm.n_slots,
                         c->zero_running ? 1 : 0, overlap ? nullptr : c->stat_err[c->slot].as<int32_t>(), c->B, c->prm.pa_min, c->prm.pa_max,
                         c->plan_in_init ? c->read_plan[c->slot].p : nullptr, c->plan_in_init ? c->stat_status[c->slot].as<int32_t>() : nullptr,
                         W, O, force_generic ? 1 : 0, LS));
    prof_end(c, c->st);
    c->stat_flags_reset = !overlap;
    c->zero_running = false;

    if (skip_oor) {
        pg_status s2 = launch_stats(c, c->st, nullptr, false, c->plan_in_init); if (s2 != PG_OK) return s2;
        HIP_TRY(c, pg_launch_apply_oor(c->st, c->B, O));
    }
    // the generic walk over the list k_batch_init has just built; not launched when the caller vouches for a batch of matches only
    if (!c->batch_all_matches) {
        prof_begin(c, "k_walk", c->st);
        HIP_TRY(c, pg_launch_walk(c->st, c->B, W, O));
        prof_end(c, c->st);
    }

    PgSortBufs S{};
    fill_sort(c, S, n_tiles);
    prof_begin(c, "k_events", c->st);
    HIP_TRY(c, pg_launch_events(c->st, c->B, W, O, c->prm.n_slots, (direct || c->part_mode) ? S.hist : nullptr, c->part_mode ? c->part_hi : 0u, c->part_lo));
    prof_end(c, c->st);
    uint64_t *acc_copy = (counts_out && counts_location == PG_LOC_DEVICE) ? counts_out : nullptr; // written by the counting kernels
    if (direct) {
        prof_begin(c, "rank_scan", c->st);
        // pg_submit (base = this context's running counts): the sample_limit cut rides in the tile scan's launch
        const bool fuse_plan = c->in_submit;
        HIP_TRY(c, pg_launch_rank_direct_count(c->st, O.ev_slot, N, c->prm.n_slots, S, c->acc_cnt.as<uint64_t>(), c->running.as<uint64_t>(), c->prm.sample_limit,
                                    c->tile_last.as<int32_t>(), acc_copy, fuse_plan ? c->keep.as<uint64_t>() : nullptr, c->ev_off.as<uint64_t>(),
                                    c->plan_totals[c->rslot].as<uint64_t>(), c->errflag.as<uint32_t>() + 6, &c->plan_done,
                                    O.btot, dense_direct(c, N) ? c->part_Bp.as<uint32_t>() : nullptr, dense_direct(c, N) ? c->chunk_part[c->rslot].as<uint64_t>() : nullptr));
        prof_end(c, c->st);
    } else if (c->part_mode) {
        // partitioned ranking, pass A and the counts of pass B (pg_place.hip): everything pg_count's result needs
        PgPartBufs P{};
        fill_part(c, P, Nn);
        if (N) {
            prof_begin(c, "part_tile_scan", c->st);
            HIP_TRY(c, pg_launch_part_tile_scan(c->st, P, N, O.btot, c->chunk_part[c->rslot].as<uint64_t>()));
            prof_end(c, c->st);
            prof_begin(c, "k_part_bases", c->st);
            HIP_TRY(c, pg_launch_part_bases(c->st, P, c->B, O));
            prof_end(c, c->st);
            prof_begin(c, "k_part_scatter", c->st);
            HIP_TRY(c, pg_launch_part_scatter(c->st, P, O.ev_slot, N, c->B, W, O));
            prof_end(c, c->st);
            // pg_submit (base = this context's running counts): the sample_limit cut rides in the region scan's launch, as in the direct ranking
            PgRegionCutArgs cut{};
            const bool fuse_cut = c->in_submit && !acc_copy && !getenv("PGMOVE_NO_FUSED_CUT");
            if (fuse_cut) {
                const size_t before = c->region_state.cap;
                HIP_TRY(c, c->region_state.ensure(((size_t)(1u << c->part_hi) + 2) * 8));
                if (c->region_state.cap != before) HIP_TRY(c, hipMemsetAsync(c->region_state.p, 0, c->region_state.cap, c->st));
                if (++c->region_epoch > pg_region_cut_epochs()) { c->region_epoch = 1; HIP_TRY(c, hipMemsetAsync(c->region_state.p, 0, c->region_state.cap, c->st)); }
                HIP_TRY(c, c->keep32.ensure(c->prm.n_slots * 4ull));
                cut.running = c->running.as<uint64_t>(); cut.keep = c->keep.as<uint64_t>(); cut.ev_off = c->ev_off.as<uint64_t>(); cut.totals = c->plan_totals[c->rslot].as<uint64_t>();
                cut.keep32 = c->keep32.as<uint32_t>(); cut.state = c->region_state.as<uint64_t>(); cut.limit = c->prm.sample_limit; cut.epoch = c->region_epoch;
            }
            prof_begin(c, "region_counts", c->st, true);
            HIP_TRY(c, pg_launch_region_counts(c->st, P, c->prm.n_slots, c->acc_cnt.as<uint64_t>(), acc_copy, fuse_cut ? &cut : nullptr));
            prof_end(c, c->st);
            c->plan_done = fuse_cut;
        } else {
            HIP_TRY(c, hipMemsetAsync(c->acc_cnt.p, 0, c->prm.n_slots * 8ull, c->st));
            if (acc_copy) HIP_TRY(c, hipMemsetAsync(acc_copy, 0, c->prm.n_slots * 8ull, c->st));
            HIP_TRY(c, hipMemsetAsync(c->part_ntiles.p, 0, 4, c->st));
        }
    } else {
        prof_begin(c, "sort_events", c->st, true);
        if (N) HIP_TRY(c, pg_launch_sort_events(c->st, O.ev_slot, N, c->key_bits, S, &c->sorted_idx));
        else { c->sorted_idx = 0; HIP_TRY(c, hipMemsetAsync(c->scount.p, 0, 8, c->st)); }
        prof_end(c, c->st);
        prof_begin(c, "slot_bounds", c->st, true);
        HIP_TRY(c, pg_launch_slot_bounds(c->st, S.keys[c->sorted_idx], S.count, N, c->slot_start.as<uint32_t>(), c->slot_end.as<uint32_t>(),
                              c->prm.n_slots, c->acc_cnt.as<uint64_t>(), acc_copy));
        prof_end(c, c->st);
    }

    // Statistics only need the signal: in eager mode they run on the second stream, overlapping the
    // latency-bound walk/rank chain above; pg_collect joins the two streams before the gather.
    c->stats_in_flight = false;
    c->stats_deferred = eager_stats && !skip_oor && !overlap && (c->prm.flags & PG_FLAG_DEFER_STATS) != 0;
    // PG_FLAG_OVERLAP_TAIL: the statistics fork off HERE, behind the counting kernels, onto the second stream; the sample_limit
    // cut, the emit kernel and the offset scan of pg_collect -- small launches that keep a fraction of the chip busy -- run
    // next to the streaming kernel, and pg_collect joins the two streams in front of the gather (the only consumer of med/MAD).
    const bool tail = eager_stats && !skip_oor && !overlap && !c->stats_deferred && (c->prm.flags & PG_FLAG_OVERLAP_TAIL) != 0;
    if (eager_stats && !skip_oor && !c->stats_deferred) {
        hipStream_t ss = (overlap || tail) ? c->st2 : c->st;
        if (tail) {
            HIP_TRY(c, hipEventRecord(c->ev_fork, c->st));
            HIP_TRY(c, hipStreamWaitEvent(c->st2, c->ev_fork, 0));
        }
        pg_status s2 = launch_stats(c, ss, nullptr, tail, c->plan_in_init, /*rare_with_scan=*/ss == c->st);
        if (s2 != PG_OK) return s2;
        if (overlap || tail) HIP_TRY(c, hipEventRecord(c->ev_join[c->slot], c->st2));
        c->stats_in_flight = overlap || tail;
    }

    if (counts_out && counts_location != PG_LOC_DEVICE) { // a device output has been written by the counting kernels themselves
        HIP_TRY(c, hipMemcpyAsync(counts_out, c->acc_cnt.p, c->prm.n_slots * 8ull, hipMemcpyDeviceToHost, c->st));
        HIP_TRY(c, hipStreamSynchronize(c->st));
    }
    c->have_count = true;
    PG_TMARK("count: kernels queued");
    if (timing_on()) { HIP_TRY(c, hipStreamSynchronize(c->st)); PG_TMARK("count: kernels done (sync)"); }
    return PG_OK;
}

static pg_status collect_impl(pg_ctx *c, const uint64_t *base, int32_t base_location, const uint64_t *all_counts, uint32_t world, uint32_t rank);

pg_status pg_stats(pg_ctx *c) {
    if (!c) return PG_ERR_INVALID_ARG;
    if (!c->have_count) return fail(c, PG_ERR_STATE, "pg_stats without a preceding pg_count");
    if (!c->stats_deferred) return PG_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    c->stats_deferred = false;
    return launch_stats(c, c->st, nullptr, false, c->plan_in_init, true);
}

// pg_job.hip, RCCL exchange: pg_stats for a rank whose base is only known on the device. all_counts / world / rank as pg_collect_gathered
// takes them, the table complete on the context's stream. If every k-mer is complete below this rank (gmove.cpp:733-735: the reference
// would not have read this shard), the statistics launches that are queued here touch no sample: the decision is a flag on the device.
pg_status pgi_stats_gathered(pg_ctx *c, const uint64_t *all_counts, uint32_t world, uint32_t rank) {
    if (!c) return PG_ERR_INVALID_ARG;
    if (!c->have_count) return fail(c, PG_ERR_STATE, "pgi_stats_gathered without a preceding pg_count");
    if (!all_counts || rank >= world) return fail(c, PG_ERR_INVALID_ARG, "pgi_stats_gathered: all_counts / world / rank");
    if (!c->stats_deferred) return PG_OK;
    if (rank == 0 || c->prm.sample_limit == 0 || !c->plan_in_init) return pg_stats(c); // nothing below this rank / nothing is ever complete / no records yet
    HIP_TRY(c, hipSetDevice(c->device));
    c->stats_deferred = false;
    { pg_status se = ensure_stats_buffers(c); if (se != PG_OK) return se; }
    HIP_TRY(c, c->cancel_flag.ensure(16));
    HIP_TRY(c, pg_launch_stats_cancel_if_full(c->st, all_counts, rank, c->prm.n_slots, c->prm.sample_limit, c->cancel_flag.as<uint32_t>(),
                                              c->read_plan[c->slot].p, c->B.n_reads));
    c->cancel_pending = true;
    return launch_stats(c, c->st, nullptr, false, true, true);
}

pg_status pg_collect(pg_ctx *c, const uint64_t *base, int32_t base_location) { return collect_impl(c, base, base_location, nullptr, 0, 0); }

pg_status pg_collect_gathered(pg_ctx *c, const uint64_t *all_counts, uint32_t world, uint32_t rank) {
    if (!c) return PG_ERR_INVALID_ARG;
    if (!all_counts || world == 0 || rank >= world) return fail(c, PG_ERR_INVALID_ARG, "pg_collect_gathered: all_counts / world / rank");
    return collect_impl(c, nullptr, PG_LOC_DEVICE, all_counts, world, rank);
}

static pg_status collect_impl(pg_ctx *c, const uint64_t *base, int32_t base_location, const uint64_t *all_counts, uint32_t world, uint32_t rank) {
    if (!c) return PG_ERR_INVALID_ARG;
    if (!c->have_count) return fail(c, PG_ERR_STATE, "pg_collect without a preceding pg_count");
    HIP_TRY(c, hipSetDevice(c->device));
    double tmark_ = timing_on() ? wall_now() : 0.0;
    if (c->stats_deferred) { // PG_FLAG_DEFER_STATS and the caller did not place them with pg_stats
        c->stats_deferred = false;
        pg_status s2 = launch_stats(c, c->st, nullptr, false, c->plan_in_init, true);
        if (s2 != PG_OK) return s2;
    }
    const uint32_t ns = c->prm.n_slots;
    const uint64_t N = c->B.n_ops;
    const bool direct = ns <= PG_DIRECT_MAX_SLOTS;
    const uint64_t *d_base = c->running.as<uint64_t>();
    PgGathered G{}; // multi-GPU job: k_slot_plan sums the lower ranks' rows itself (and leaves the job's totals / freq.txt column)
    if (all_counts) {
        G.all_counts = all_counts; G.world = world; G.rank = rank; G.total = c->job_total.as<uint64_t>(); G.freq = c->job_freq.as<uint64_t>();
        c->have_job_totals = true;
        d_base = nullptr;
    } else if (base) {
        HIP_TRY(c, hipMemcpyAsync(c->base_stage.p, base, ns * 8ull,
                                  base_location == PG_LOC_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->st));
        d_base = c->base_stage.as<uint64_t>();
    }
    uint64_t *totals = c->plan_totals[c->rslot].as<uint64_t>();
    // capacity for the kept events of this batch: everything downstream is sized by this bound and reads the
    // actual counts from device memory, so the batch needs no host round trip
    const uint64_t ke_cap = std::min<uint64_t>(N, (uint64_t)ns * c->prm.sample_limit);
    {
        const size_t before = c->scan_scratch.cap;
        HIP_TRY(c, c->scan_scratch.ensure((std::max<uint64_t>(ke_cap, ns) / 4096 + 84) * 8));
        if (c->scan_scratch.cap != before) HIP_TRY(c, hipMemsetAsync(c->scan_scratch.p, 0, c->scan_scratch.cap, c->st)); // chained-scan state
    }
    if (!direct) HIP_TRY(c, c->keep32.ensure(ns * 4ull));
    const bool plan_in_scan = c->plan_done && !all_counts && !base;
    c->plan_done = false;
    if (!plan_in_scan) {
    prof_begin(c, "k_slot_plan", c->st, true);
    HIP_TRY(c, pg_launch_slot_plan(c->st, c->acc_cnt.as<uint64_t>(), d_base, c->running.as<uint64_t>(), c->prm.sample_limit, ns,
                        c->keep.as<uint64_t>(), c->ev_off.as<uint64_t>(), totals, direct ? c->hist.as<uint32_t>() : nullptr,
                        pg_tiles(N ? N : 1, true), direct ? nullptr : c->keep32.as<uint32_t>(), c->scan_scratch.as<uint64_t>(),
                        direct ? c->tile_last.as<int32_t>() : nullptr, G));
    prof_end(c, c->st);
    }

    const uint64_t win_cap = (uint64_t)c->prm.max_dur + 2ull * c->prm.signal_print_margin;
    const uint64_t samp_cap = ke_cap * win_cap;
    HIP_TRY(c, c->ev_rec[c->rslot].ensure((ke_cap + 1) * sizeof(PgKeptRec)));
    HIP_TRY(c, c->samp_off[c->rslot].ensure((ke_cap + 2) * 8));
    c->unpacked = false;

    PgWalkParams W{}; PgWalkOut O{};
    fill_walk(c, W, O);
    PgKeptOut K{};
    K.rec = c->ev_rec[c->rslot].as<PgKeptRec>();
    // the per-read "owns a kept event" flags are only consumed by the lazy statistics: no scattered byte stores otherwise
    K.read_needed = (c->prm.scaling == 1 && (c->prm.flags & PG_FLAG_LAZY_STATS) && !(c->prm.flags & PG_FLAG_SKIP_OUT_OF_RANGE)) ? c->read_needed.as<uint8_t>() : nullptr;
    // Many kept events (nearly every accepted event kept: large sample_limit, k = 9): the offset scan happens inside the gather's
    // workgroups (pg_place.hip). Few (the default limit: 10^5 events): the one-launch chained scan + the strided gather of round 2.
    const bool chunked = ke_cap > dense_min() && (win_cap + 1) * 4096 < (1ull << 32);
    bool sums_ready = false;
    // Two-stream mode, many kept events: the gather of this batch goes to a THIRD stream, behind the batch's statistics (second stream)
    // and its placing kernel, and the main stream goes on with the next batch's init / events / ranking / placing -- chains of dependent
    // round trips that leave the memory system idle (VALU 0.2-0.3 busy, 0.15-0.4 of the HBM rate) -- beside it. What the next batch's
    // chain writes and this gather reads exists per statistics slot (records, sample offsets, kept-event total, chunk sums); the sample
    // buffer exists once: gathers follow each other on their stream.
    // Measured (profiles/r04_side_gather.txt): 0-5 % at sample_limit 5000, nothing at k = 9, whatever share of the CUs the gather's stream
    // is given -- the chain's kernels wait inside occupied wave slots, they do not leave CUs free, and the gather needs its CUs (half of them:
    // +14 %). So it is opt-in (PGMOVE_GATHER_SIDE=1), and tests/test_gpu_parity.py runs the suite's cases through it once.
    const bool gather_side_on = c->side_enabled; // (PGMOVE_GATHER_SIDE, read when the context is created)
    const bool side = chunked && c->stats_in_flight && c->st2 && !c->user_stream && gather_side_on && (c->prm.flags & PG_FLAG_OVERLAP);
    if (side && !c->st3) { // like the statistics stream: a quarter of every XCD's CUs stays free of it, or the chain's 16-wave workgroups never find room
        const char *wh = getenv("PGMOVE_GATHER_CU_WITHHELD");
        hipDeviceProp_t prop;
        HIP_TRY(c, hipGetDeviceProperties(&prop, c->device));
        const int cus = prop.multiProcessorCount, withheld = wh ? atoi(wh) : cus / 4;
        bool masked = false;
        if (withheld > 0 && withheld < cus) {
            std::vector<uint32_t> mask((size_t)(cus + 31) / 32, 0u);
            for (int i = 0; i < cus - withheld; i++) mask[(size_t)i / 32] |= 1u << (i % 32);
            masked = hipExtStreamCreateWithCUMask(&c->st3, (uint32_t)mask.size(), mask.data()) == hipSuccess;
            if (!masked) (void)hipGetLastError();
        }
        if (!masked) {
            int prio_low = 0, prio_high = 0;
            HIP_TRY(c, hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
            HIP_TRY(c, hipStreamCreateWithPriority(&c->st3, hipStreamNonBlocking, prio_low));
        }
    }
    if (c->side_used[c->slot]) { // the gather that read this slot's records two batches ago
        HIP_TRY(c, hipStreamWaitEvent(c->st, c->ev_gathered[c->slot], 0));
        c->side_used[c->slot] = false;
    }
    if (!side && c->gather_side) { // this batch's gather runs on the main stream and writes the sample buffer the previous one is still filling
        HIP_TRY(c, hipStreamWaitEvent(c->st, c->ev_gathered[c->gather_side_slot], 0));
        c->gather_side = false;
    }
    if (chunked) HIP_TRY(c, c->chunk_part[c->rslot].ensure(PG_CHUNK_PART_N * 8)); // (partitioned ranking has it already, zeroed by its scan launch in pg_count)
    if (direct) {
        PgSortBufs S{};
        fill_sort(c, S, 0);
        prof_begin(c, "k_rank_emit", c->st);
        if (dense_direct(c, N)) HIP_TRY(c, pg_launch_rank_emit2(c->st, O.ev_slot, N, ns, S.hist, c->keep.as<uint64_t>(), c->ev_off.as<uint64_t>(), totals, c->B, W, O, K, c->part_Bp.as<uint32_t>()));
        else HIP_TRY(c, pg_launch_rank_direct_emit(c->st, O.ev_slot, N, ns, S, c->keep.as<uint64_t>(), c->ev_off.as<uint64_t>(), totals, c->B, W, O, K));
        prof_end(c, c->st);
    } else if (c->part_mode) {
        PgPartBufs P{};
        fill_part(c, P, N ? N : 1);
        if (N) {
            prof_begin(c, "k_region_place", c->st);
            HIP_TRY(c, pg_launch_region_place(c->st, P, ns, c->keep32.as<uint32_t>(), c->ev_off.as<uint64_t>(), O, K, chunked ? c->chunk_part[c->rslot].as<uint64_t>() : nullptr, chunked ? ke_cap : 0));
            prof_end(c, c->st);
            sums_ready = chunked;
        }
    } else {
        prof_begin(c, "k_kept_meta", c->st, true);
        HIP_TRY(c, pg_launch_kept_meta(c->st, c->sk[c->sorted_idx].as<uint32_t>(), c->sv[c->sorted_idx].as<uint32_t>(), c->scount.as<uint32_t>(), N,
                            c->slot_start.as<uint32_t>(), c->keep.as<uint64_t>(), c->ev_off.as<uint64_t>(), c->B, W, O, K,
                            c->sv[c->sorted_idx ^ 1].as<uint32_t>()));
        prof_end(c, c->st);
    }

    // lazy statistics: only the reads that own a kept event (flags written above)
    const bool lazy = c->prm.scaling == 1 && (c->prm.flags & PG_FLAG_LAZY_STATS) && !(c->prm.flags & PG_FLAG_SKIP_OUT_OF_RANGE);
    if (lazy) { pg_status s2 = launch_stats(c, c->st, c->read_needed.as<uint8_t>(), false, false, true); if (s2 != PG_OK) return s2; }

    uint64_t gather_cap = ke_cap;
    if (chunked) {
        // the chunk sums of the kept window lengths: accumulated by k_region_place (partitioned ranking), else one pass over the records;
        // the rare statistics ride in that pass's launch, or get their own
        if (!sums_ready) {
            // the coarse sums are ADDED to: zeroed by the tile scan's extra workgroup where there is one (dense direct ranking; partitioned
            // ranking, whose placing kernel has filled them by now), else here (the radix-sort ranking, PGMOVE_EMIT1)
            if (!(direct && dense_direct(c, N)) && !c->part_mode) HIP_TRY(c, hipMemsetAsync(c->chunk_part[c->rslot].p, 0, PG_CHUNK_PART_N * 8, c->st));
            prof_begin(c, "len_partials", c->st);
            HIP_TRY(c, pg_launch_len_partials(c->st, ke_cap, totals, c->ev_rec[c->rslot].as<PgKeptRec>(), c->chunk_part[c->rslot].as<uint64_t>(), c->rare_pending ? &c->rare : nullptr));
            prof_end(c, c->st);
        } else if (c->rare_pending) {
            prof_begin(c, "k_read_stats_rare", c->st);
            HIP_TRY(c, pg_launch_read_stats_rare(c->st, c->rare));
            prof_end(c, c->st);
        }
        c->rare_pending = false;
    } else {
        prof_begin(c, "scan_ev_len", c->st, /*bracket=*/(ke_cap + 4095) / 4096 > 64); // long inputs: three launches (pg_launch_scan_u32_u64)
        HIP_TRY(c, pg_launch_scan_u32_u64(c->st, reinterpret_cast<const uint32_t *>(c->ev_rec[c->rslot].p) + 2, 4, ke_cap, totals, c->samp_off[c->rslot].as<uint64_t>(), c->scan_scratch.as<uint64_t>(),
                                          c->rare_pending ? &c->rare : nullptr, totals + 2));
        c->rare_pending = false;
        prof_end(c, c->st);
    }

    if (samp_cap * 8 > c->samples.cap) {
        // grow once to the worst case if that is moderate; otherwise size exactly from the device total (one sync)
        // (moderate: an eighth of the device's memory -- k = 9 with the default max_dur has a 9 GB worst case for 1.5 GB of samples, and
        // sizing it exactly costs a host round trip per batch, which is what kept k = 9 batches from overlapping each other)
        size_t mem_free = 0, mem_total = 0;
        if (hipMemGetInfo(&mem_free, &mem_total) != hipSuccess) { (void)hipGetLastError(); mem_total = 0; }
        const uint64_t moderate = std::max<uint64_t>(4ull << 30, std::min<uint64_t>(mem_total / 8, mem_free / 2));
        bool sized = false;
        if (samp_cap * 8 <= moderate && !getenv("PGMOVE_SAMPLES_EXACT")) { // (PGMOVE_SAMPLES_EXACT: tests of the exact-size branch)
            // hipMemGetInfo races with every other context and process on the device (pg_job with one device listed twice, two ranks on one
            // GPU): a worst-case allocation that fails after all is not an error, the exact size below still fits
            sized = c->samples.ensure(samp_cap * 8 + 8) == hipSuccess;
            if (!sized) (void)hipGetLastError();
        }
        if (!sized) {
            uint64_t tot[3] = {0, 0, 0};
            HIP_TRY(c, hipMemcpyAsync(tot, totals, 24, hipMemcpyDeviceToHost, c->st));
            if (chunked) { // the kept samples' total is left by the GATHER there (its last chunk); in front of it: the sum of the coarse chunk sums
                uint64_t coarse[PG_CHUNK_COARSE];
                HIP_TRY(c, hipMemcpyAsync(coarse, c->chunk_part[c->rslot].as<uint64_t>() + PG_CHUNK_FINE, sizeof coarse, hipMemcpyDeviceToHost, c->st));
                HIP_TRY(c, hipStreamSynchronize(c->st));
                tot[2] = 0;
                for (uint64_t v : coarse) tot[2] += v;
            } else HIP_TRY(c, hipStreamSynchronize(c->st));
            HIP_TRY(c, c->samples.ensure((tot[2] + 1) * 8)); // all kept samples
            gather_cap = tot[0];
        }
    }
    hipStream_t gst = c->st;
    if (side) { // behind this batch's statistics (second stream) and behind the placing kernel + chunk sums of the main one
        HIP_TRY(c, hipEventRecord(c->ev_fork, c->st));
        HIP_TRY(c, hipStreamWaitEvent(c->st3, c->ev_fork, 0));
        HIP_TRY(c, hipStreamWaitEvent(c->st3, c->ev_join[c->slot], 0));
        gst = c->st3; c->stats_in_flight = false;
    }
    if (c->stats_in_flight) { HIP_TRY(c, hipStreamWaitEvent(c->st, c->ev_join[c->slot], 0)); c->stats_in_flight = false; }
    prof_begin(c, "k_gather", gst);
    if (chunked)
        HIP_TRY(c, pg_launch_gather_chunks(gst, c->B, ke_cap, totals, c->ev_rec[c->rslot].as<PgKeptRec>(), c->chunk_part[c->rslot].as<uint64_t>(), c->samp_off[c->rslot].as<uint64_t>(), totals + 2, c->prm.scaling,
                                           c->prm.pa_min, c->prm.pa_max, c->samples.as<double>(), c->prm.scaling == 1 ? c->gcal[c->slot].as<double>() : nullptr, gather_lanes(c),
                                           c->prm.scaling == 1 ? c->stat_err[c->slot].as<int32_t>() : nullptr));
    else
        HIP_TRY(c, pg_launch_gather(c->st, c->B, gather_cap, totals, c->ev_rec[c->rslot].as<PgKeptRec>(),
                     c->samp_off[c->rslot].as<uint64_t>(), c->prm.scaling, c->prm.pa_min, c->prm.pa_max, c->med[c->slot].as<double>(), c->mad[c->slot].as<double>(),
                     c->samples.as<double>(), c->prm.scaling == 1 ? c->gcal[c->slot].as<double>() : nullptr));
    prof_end(c, gst);
    // "this slot's statistics buffers have been read": for the statistics stream two batches on (and the side gather). With one stream nobody
    // waits for it, and a record ends the gather with 4.6 us in which the stream starts nothing (rocprofv3 kernel trace, tools/trace_gaps.py:
    // every other kernel-to-kernel gap of the chain is 0.0 us).
    if (c->st2 || c->st3) HIP_TRY(c, hipEventRecord(c->ev_gathered[c->slot], gst));
    if (side) { c->gather_side = true; c->gather_side_slot = c->slot; c->side_used[c->slot] = true; }
    PG_TMARK("collect: buffers + kernels queued");
    if (timing_on()) { HIP_TRY(c, hipStreamSynchronize(c->st)); PG_TMARK("collect: kernels done (sync)"); }
    c->slot_used[c->slot] = true;
    c->have_count = false; c->have_batch_result = true; c->downloaded = false; c->totals_known = false;
    return PG_OK;
}

// after a collect: wait for the batch, surface per-read errors, learn how many events/samples were kept
static pg_status settle_batch(pg_ctx *c) {
    if (!c->have_batch_result || c->totals_known) return PG_OK;
    // the other streams first: the record is packed on the main one, behind everything that writes what it reads (k_settle_pack: one launch
    // into host-mapped memory instead of four blocking copies of 8-24 bytes, ~80 -> ~10 us per call)
    if (c->st2) HIP_TRY(c, hipStreamSynchronize(c->st2));
    if (c->st3) HIP_TRY(c, hipStreamSynchronize(c->st3));
    HIP_TRY(c, pg_launch_settle_pack(c->st, c->errflag.as<uint32_t>(), c->stat_err[c->slot].as<int32_t>(), c->long_ring.as<int32_t>() + 4u * (c->long_seq & 3u), c->plan_totals[c->rslot].as<uint64_t>(),
                                     c->samp_off[c->rslot].as<uint64_t>(), c->samp_off[c->rslot].cap / 8, c->cancel_pending ? c->cancel_flag.as<uint32_t>() : nullptr, c->settle_host));
    HIP_TRY(c, hipStreamSynchronize(c->st));
    const PgSettlePack pk = *c->settle_host;
    pg_status s = check_read_errors(c, pk);
    if (s != PG_OK) { c->have_batch_result = false; c->downloaded = true; return s; }
    const uint64_t tot[2] = {pk.n_kept, pk.full_slots};
    c->cur_n_kept = tot[0]; c->full_slots = tot[1];
    const uint64_t n_samples = pk.n_samples;
    c->cur_n_samples = n_samples;
    if (c->cancel_pending) {
        if (pk.cancel[1]) c->stats_cancelled++;
        c->cancel_pending = false;
    }
    if (tot[0]) c->win_hint = (uint32_t)((n_samples + tot[0] - 1) / tot[0]);
    c->totals_known = true;
    return PG_OK;
}

pg_status pg_submit(pg_ctx *c, const pg_batch *b) {
    if (!c) return PG_ERR_INVALID_ARG;
    c->in_submit = true;
    pg_status s = pg_count(c, b, nullptr, PG_LOC_HOST);
    c->in_submit = false;
    if (s != PG_OK) return s;
    return pg_collect(c, nullptr, PG_LOC_HOST);
}

pg_status pg_job_totals_device(pg_ctx *c, const uint64_t **d_total, const uint64_t **d_freq) {
    if (!c) return PG_ERR_INVALID_ARG;
    if (!c->have_job_totals) return fail(c, PG_ERR_STATE, "pg_job_totals_device without a preceding pg_collect_gathered");
    if (d_total) *d_total = c->job_total.as<uint64_t>();
    if (d_freq) *d_freq = c->job_freq.as<uint64_t>();
    return PG_OK;
}

pg_status pg_sync(pg_ctx *c) {
    if (!c) return PG_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->st));
    if (c->st2) HIP_TRY(c, hipStreamSynchronize(c->st2));
    if (c->st3) HIP_TRY(c, hipStreamSynchronize(c->st3));
    return settle_batch(c);
}

pg_status pg_set_stream(pg_ctx *c, void *hip_stream) {
    if (!c) return PG_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->st));
    if (c->st2) HIP_TRY(c, hipStreamSynchronize(c->st2));
    if (c->st3) HIP_TRY(c, hipStreamSynchronize(c->st3));
    c->st = hip_stream ? (hipStream_t)hip_stream : c->own_st;
    c->user_stream = hip_stream != nullptr;
    return PG_OK;
}

// pg_job.hip: the stream the chain runs on (events of the job's communication stream are ordered against it)
void *pgi_stream(pg_ctx *c) { return c ? (void *)c->st : nullptr; }
// the job's k-mer-major sample stream on the device after pg_finish_deferred (null: it is on the host, or there is none): pg_job's device-side merge
const double *pgi_fin_dev(pg_ctx *c) { return c && c->merged_valid ? c->fin_dev : nullptr; }
// pg_job.hip, rank-level early-out: the batch counted with PG_FLAG_DEFER_STATS will keep no event (every k-mer is complete below this
// rank), so the collect that follows needs no statistics: none are queued (the gather reads a read's median / MAD for kept events only)
pg_status pgi_skip_stats(pg_ctx *c) {
    if (!c) return PG_ERR_INVALID_ARG;
    if (!c->have_count) return fail(c, PG_ERR_STATE, "pgi_skip_stats without a preceding pg_count");
    c->stats_deferred = false;
    return PG_OK;
}

int32_t pg_all_slots_full(pg_ctx *c) {
    if (!c) return 0;
    if (settle_batch(c) != PG_OK) return 0;
    return c->full_slots == c->prm.n_slots;
}

int32_t pg_poll(pg_ctx *c) {
    if (!c) return PG_ERR_INVALID_ARG;
    if (!c->have_batch_result || c->totals_known) return 1;
    if (hipSetDevice(c->device) != hipSuccess) return PG_ERR_HIP;
    if (hipStreamQuery(c->st) != hipSuccess || (c->st2 && hipStreamQuery(c->st2) != hipSuccess) || (c->st3 && hipStreamQuery(c->st3) != hipSuccess)) { (void)hipGetLastError(); return 0; } // hipErrorNotReady
    const pg_status s = settle_batch(c);
    return s == PG_OK ? 1 : s;
}

int32_t pg_all_slots_full_settled(const pg_ctx *c) { return c && c->full_slots == c->prm.n_slots; }
uint64_t pgi_full_slots_settled(const pg_ctx *c) { return c ? c->full_slots : 0; } // pg_job.hip: how many k-mers were complete after the last settled batch

pg_status pg_last_batch_device(pg_ctx *c, pg_device_view *v) {
    if (!c || !v) return PG_ERR_INVALID_ARG;
    if (!c->have_batch_result) return fail(c, PG_ERR_STATE, "no collected batch");
    { pg_status s0 = settle_batch(c); if (s0 != PG_OK) return s0; }
    { pg_status su = ensure_unpacked(c); if (su != PG_OK) return su; }
    v->n_events = c->cur_n_kept; v->n_samples = c->cur_n_samples;
    v->d_keep = c->keep.as<uint64_t>(); v->d_ev_off = c->ev_off.as<uint64_t>(); v->d_ev_len = c->ev_len.as<uint32_t>();
    v->d_ev_read = c->ev_read.as<uint32_t>(); v->d_samp_off = c->samp_off[c->rslot].as<uint64_t>(); v->d_samples = c->samples.as<double>();
    v->d_med = c->prm.scaling == 1 ? c->med[c->slot].as<double>() : nullptr; v->d_mad = c->prm.scaling == 1 ? c->mad[c->slot].as<double>() : nullptr;
    return PG_OK;
}

pg_status pg_finish(pg_ctx *c, pg_result *out) {
    if (!c || !out) return PG_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    pg_status s = download_last(c, false);
    if (s != PG_OK) return s;
    const uint32_t ns = c->prm.n_slots;
    if (c->merged_valid && c->want_samples && c->fin_dev && c->m_samples) { // deferred before, wanted now: one download into the merged view
        c->r_samples.resize(c->m_samples);
        HIP_TRY(c, hipMemcpy(c->r_samples.data(), c->fin_dev, c->m_samples * 8ull, hipMemcpyDeviceToHost));
        c->fin_dev = nullptr;
        for (auto &hb : c->batches) if (hb.on_device && c->batches.size() == 1) { hb.on_device = false; hb.in_ctx = false; hb.dsamples.release(); }
    }
    if (c->merged_valid) { // nothing was collected since the last call: the merged view is current
        out->n_slots = ns; out->reserved = 0; out->n_events = c->m_events; out->n_samples = c->m_samples; out->n_reads = c->m_reads;
        out->counts = c->r_counts.data(); out->ev_off = c->r_ev_off.data(); out->ev_len = c->r_ev_len.data();
        out->ev_read = c->r_ev_read.data(); out->samp_off = c->r_samp_off.data(); out->samples = c->fin_dev ? nullptr : c->r_samples.data();
        out->read_skipped = c->r_skipped.data();
        return PG_OK;
    }
    if (c->single_moved && !c->batches.empty()) { // give the first batch its arrays back (see the one-batch path below)
        HostBatchResult &h = c->batches[0];
        c->r_ev_off.swap(h.ev_off); c->r_samp_off.swap(h.samp_off); c->r_ev_len.swap(h.ev_len); c->r_ev_read.swap(h.ev_read);
        c->r_samples.swap(h.samples); c->r_skipped.swap(h.skipped);
    }
    c->single_moved = false;
    // merge the batches slot-major, batch-minor: reference order is PAF-line order inside every k-mer file
    uint64_t n_events = 0, n_samples = 0, n_reads = 0;
    for (auto &h : c->batches) { n_events += h.n_events; n_samples += h.n_samples; n_reads += h.n_reads; }
    if (c->batches.size() == 1) { // one batch: its arrays ARE the result (no per-event copy of up to GBs of samples)
        HostBatchResult &h = c->batches[0];
        c->fin_dev = nullptr;
        if (h.on_device && !c->want_samples) c->fin_dev = h.in_ctx ? c->samples.as<double>() : h.dsamples.as<double>(); // pg_fetch_samples reads them there
        else if (h.on_device) { // parked on the device (for a merge that never came, or by an earlier deferred finish): fetch them now
            h.samples.resize(h.n_samples);
            HIP_TRY(c, hipMemcpy(h.samples.data(), h.in_ctx ? c->samples.p : h.dsamples.p, h.n_samples * 8ull, hipMemcpyDeviceToHost));
            h.dsamples.release(); h.on_device = false; h.in_ctx = false;
        }
        c->r_counts.resize(ns);
        for (uint32_t sl = 0; sl < ns; sl++) c->r_counts[sl] = h.ev_off[sl + 1] - h.ev_off[sl];
        c->r_ev_off.swap(h.ev_off); c->r_samp_off.swap(h.samp_off); c->r_ev_len.swap(h.ev_len); c->r_ev_read.swap(h.ev_read);
        c->r_samples.swap(h.samples); c->r_skipped.swap(h.skipped);
        c->single_moved = true; // undone at the top of the next pg_finish (more batches may follow) and by pg_reset
        out->n_slots = ns; out->reserved = 0; out->n_events = n_events; out->n_samples = n_samples; out->n_reads = n_reads;
        out->counts = c->r_counts.data(); out->ev_off = c->r_ev_off.data(); out->ev_len = c->r_ev_len.data();
        out->ev_read = c->r_ev_read.data(); out->samp_off = c->r_samp_off.data(); out->samples = c->fin_dev ? nullptr : c->r_samples.data();
        out->read_skipped = c->r_skipped.data();
        c->merged_valid = true; c->m_events = n_events; c->m_samples = n_samples; c->m_reads = n_reads;
        return PG_OK;
    }
    c->fin_dev = nullptr;
    c->r_counts.assign(ns, 0); c->r_ev_off.assign(ns + 1, 0); c->r_samp_off.resize(n_events + 1);
    c->r_ev_len.resize(n_events); c->r_ev_read.resize(n_events);
    c->r_skipped.resize(n_reads);
    // pass 1 (cheap): where every slot's events and samples start in the merged arrays
    std::vector<uint64_t> slot_e(ns + 1, 0), slot_s(ns + 1, 0);
    for (uint32_t sl = 0; sl < ns; sl++) {
        uint64_t ne = 0, nsmp = 0;
        for (auto &h : c->batches) { const uint64_t a = h.ev_off[sl], b = h.ev_off[sl + 1]; ne += b - a; nsmp += h.samp_off[b] - h.samp_off[a]; }
        slot_e[sl + 1] = slot_e[sl] + ne; slot_s[sl + 1] = slot_s[sl] + nsmp;
        c->r_counts[sl] = ne; c->r_ev_off[sl] = slot_e[sl];
    }
    // the samples: merged on the device when every batch's are still there (the usual case: download_last), else on the host
    bool all_dev = true;
    for (auto &h : c->batches) if (h.n_samples && !h.on_device) all_dev = false;
    if (all_dev && n_samples) { // room for the merged copy and the segment list? Else the host merge after all
        size_t nseg = 0;
        for (uint32_t sl = 0; sl < ns; sl++) for (auto &h : c->batches) nseg += h.ev_off[sl + 1] > h.ev_off[sl];
        if (c->dseg.ensure(nseg * sizeof(PgSeg) + 16) != hipSuccess || c->dmerged.ensure(n_samples * 8ull) != hipSuccess) { (void)hipGetLastError(); c->dmerged.release(); all_dev = false; }
    }
    if (!(all_dev && !c->want_samples)) c->r_samples.resize(n_samples); // (a deferred finish leaves the device merge's result on the device)
    if (!all_dev)
        for (auto &h : c->batches)
            if (h.on_device) { // mixed (a pg_finish between batches, or a batch that found no room): this one through the host after all
                h.samples.resize(h.n_samples);
                HIP_TRY(c, hipMemcpy(h.samples.data(), h.dsamples.p, h.n_samples * 8ull, hipMemcpyDeviceToHost));
                h.dsamples.release(); h.on_device = false;
            }
    // pass 2: the copies (hundreds of MB at large limits), slot ranges side by side on a few threads
    auto merge_slots = [&](uint32_t s0, uint32_t s1) {
        for (uint32_t sl = s0; sl < s1; sl++) {
            uint64_t e = slot_e[sl], sp = slot_s[sl], rbase = 0;
            for (auto &h : c->batches) {
                const uint64_t a = h.ev_off[sl], b = h.ev_off[sl + 1];
                if (b > a) {
                    const uint64_t s_a = h.samp_off[a], s_b = h.samp_off[b];
                    if (!all_dev) memcpy(&c->r_samples[sp], &h.samples[s_a], (s_b - s_a) * sizeof(double)); // a slot's events of one batch are contiguous
                    for (uint64_t i = a; i < b; i++, e++) {
                        c->r_ev_len[e] = h.ev_len[i]; c->r_ev_read[e] = (uint32_t)(rbase + h.ev_read[i]); c->r_samp_off[e] = sp + (h.samp_off[i] - s_a);
                    }
                    sp += s_b - s_a;
                }
                rbase += h.n_reads;
            }
        }
    };
    {
        const unsigned nt = n_samples >= (8u << 20) ? 8u : 1u;
        if (nt == 1) merge_slots(0, ns);
        else {
            std::vector<std::thread> pool;
            uint32_t s0 = 0;
            for (unsigned t = 0; t < nt; t++) { // equal shares of the samples
                uint32_t s1 = s0;
                const uint64_t want = slot_s[ns] / nt * (t + 1);
                while (s1 < ns && (t + 1 == nt || slot_s[s1 + 1] <= want)) s1++;
                if (s1 > s0) pool.emplace_back(merge_slots, s0, s1);
                s0 = s1;
            }
            for (auto &th : pool) th.join();
        }
    }
    if (all_dev && n_samples) {
        std::vector<PgSeg> segs;
        segs.reserve((size_t)ns * c->batches.size());
        for (uint32_t sl = 0; sl < ns; sl++) {
            uint64_t sp2 = slot_s[sl];
            for (auto &h : c->batches) {
                const uint64_t a2 = h.ev_off[sl], b2 = h.ev_off[sl + 1];
                if (b2 > a2) {
                    const uint64_t s_a = h.samp_off[a2], s_b = h.samp_off[b2];
                    if (s_b > s_a) segs.push_back(PgSeg{h.dsamples.as<double>() + s_a, sp2, s_b - s_a});
                    sp2 += s_b - s_a;
                }
            }
        }
        HIP_TRY(c, c->dseg.ensure(segs.size() * sizeof(PgSeg) + 16)); HIP_TRY(c, c->dmerged.ensure(n_samples * 8ull));
        HIP_TRY(c, hipMemcpyAsync(c->dseg.p, segs.data(), segs.size() * sizeof(PgSeg), hipMemcpyHostToDevice, c->st));
        HIP_TRY(c, pg_launch_merge_segments(c->st, c->dseg.as<PgSeg>(), (uint32_t)segs.size(), c->dmerged.as<double>(), n_samples));
        HIP_TRY(c, hipStreamSynchronize(c->st));
        if (!c->want_samples) c->fin_dev = c->dmerged.as<double>();
        // one download of the merged samples, slices side by side (pageable destination, first touch included)
        const uint64_t bytes = c->want_samples ? n_samples * 8ull : 0;
        const unsigned parts = bytes >= (64ull << 20) ? 8u : 1u;
        std::vector<hipError_t> rc(parts, hipSuccess);
        std::vector<std::thread> pool;
        const uint64_t step = ((n_samples + parts - 1) / parts + 511) & ~511ull;
        for (unsigned t = 0; t < parts; t++)
            pool.emplace_back([&, t]() {
                const uint64_t a2 = std::min<uint64_t>(n_samples, t * step), b2 = std::min<uint64_t>(n_samples, a2 + step);
                if (b2 <= a2 || !bytes) return;
                rc[t] = hipSetDevice(c->device);
                if (rc[t] == hipSuccess) rc[t] = hipMemcpy(c->r_samples.data() + a2, c->dmerged.as<double>() + a2, (b2 - a2) * 8ull, hipMemcpyDeviceToHost);
            });
        for (auto &th : pool) th.join();
        for (hipError_t e2 : rc) HIP_TRY(c, e2);
    }
    const uint64_t e = slot_e[ns], sp = slot_s[ns];
    c->r_ev_off[ns] = e; c->r_samp_off[n_events] = sp;
    uint64_t rb = 0;
    for (auto &h : c->batches) { if (h.n_reads) memcpy(&c->r_skipped[rb], h.skipped.data(), h.n_reads); rb += h.n_reads; }
    out->n_slots = ns; out->reserved = 0; out->n_events = n_events; out->n_samples = n_samples; out->n_reads = n_reads;
    out->counts = c->r_counts.data(); out->ev_off = c->r_ev_off.data(); out->ev_len = c->r_ev_len.data();
    out->ev_read = c->r_ev_read.data(); out->samp_off = c->r_samp_off.data(); out->samples = c->fin_dev ? nullptr : c->r_samples.data();
    out->read_skipped = c->r_skipped.data();
    c->merged_valid = true; c->m_events = n_events; c->m_samples = n_samples; c->m_reads = n_reads;
    return PG_OK;
}

pg_status pg_finish_deferred(pg_ctx *c, pg_result *out) {
    if (!c || !out) return PG_ERR_INVALID_ARG;
    c->want_samples = false;
    const pg_status s = pg_finish(c, out);
    c->want_samples = true;
    return s;
}

pg_status pg_fetch_samples(pg_ctx *c, uint64_t first, uint64_t n, double *dst) {
    if (!c || (!dst && n)) return PG_ERR_INVALID_ARG;
    if (!c->merged_valid) return PG_ERR_STATE; // (no error text: several host threads may call this at once)
    if (first > c->m_samples || n > c->m_samples - first) return PG_ERR_INVALID_ARG;
    if (!n) return PG_OK;
    if (!c->fin_dev) { memcpy(dst, c->r_samples.data() + first, n * sizeof(double)); return PG_OK; } // the merge went through the host
    if (hipSetDevice(c->device) != hipSuccess || hipMemcpy(dst, c->fin_dev + first, n * 8ull, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return PG_ERR_HIP; }
    return PG_OK;
}

pg_status pg_text_device(pg_ctx *c, uint32_t ns, uint64_t ne, const uint64_t *d_ev_off, const uint64_t *d_samp_off, const double *d_samples, pg_text_result *out) {
    if (!c || !out) return PG_ERR_INVALID_ARG;
    if (ns && (!d_ev_off || !d_samp_off)) return fail(c, PG_ERR_INVALID_ARG, "pg_text_device: ev_off / samp_off missing");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, c->tx_len.ensure((ne + 1) * 4ull)); HIP_TRY(c, c->tx_off.ensure((ne + 2) * 8ull)); HIP_TRY(c, c->tx_flag.ensure(16));
    HIP_TRY(c, c->tx_slot_off.ensure((ns + 2) * 8ull));
    {
        const size_t before = c->scan_scratch.cap;
        HIP_TRY(c, c->scan_scratch.ensure((std::max<uint64_t>(ne, ns) / 4096 + 84) * 8));
        if (c->scan_scratch.cap != before) HIP_TRY(c, hipMemsetAsync(c->scan_scratch.p, 0, c->scan_scratch.cap, c->st));
    }
    prof_begin(c, "text", c->st, true);
    HIP_TRY(c, pg_launch_text_lens(c->st, d_samples, d_samp_off, ne, c->tx_len.as<uint32_t>(), c->tx_flag.as<uint32_t>()));
    HIP_TRY(c, pg_launch_scan_u32_u64(c->st, c->tx_len.as<uint32_t>(), 1, ne, nullptr, c->tx_off.as<uint64_t>(), c->scan_scratch.as<uint64_t>(), nullptr, c->tx_flag.as<uint64_t>() + 1));
    uint64_t head[2] = {0, 0}; // [0] flag, [1] total bytes
    HIP_TRY(c, hipMemcpyAsync(head, c->tx_flag.p, 16, hipMemcpyDeviceToHost, c->st));
    HIP_TRY(c, hipStreamSynchronize(c->st));
    if ((uint32_t)head[0]) { prof_end(c, c->st); return fail(c, PG_ERR_UNSUPPORTED, "pg_text: a kept sample is not finite or |sample| >= 4e7; format on the host"); }
    c->tx_bytes = ne ? head[1] : 0;
    HIP_TRY(c, c->tx_text.ensure(c->tx_bytes + 16));
    HIP_TRY(c, pg_launch_text_write(c->st, d_samples, d_samp_off, ne, c->tx_off.as<uint64_t>(), c->tx_text.as<char>(), d_ev_off, ns, c->tx_slot_off.as<uint64_t>()));
    prof_end(c, c->st);
    c->tx_slot_off_host.resize(ns + 1);
    HIP_TRY(c, hipMemcpyAsync(c->tx_slot_off_host.data(), c->tx_slot_off.p, (ns + 1) * 8ull, hipMemcpyDeviceToHost, c->st));
    HIP_TRY(c, hipStreamSynchronize(c->st));
    if (!ne) std::fill(c->tx_slot_off_host.begin(), c->tx_slot_off_host.end(), 0);
    out->n_slots = ns; out->reserved = 0; out->n_bytes = c->tx_bytes; out->slot_off = c->tx_slot_off_host.data();
    return PG_OK;
}

pg_status pg_text(pg_ctx *c, pg_text_result *out) {
    if (!c || !out) return PG_ERR_INVALID_ARG;
    pg_result R;
    pg_status s = pg_finish_deferred(c, &R);
    if (s != PG_OK) return s;
    if (R.n_samples && !c->fin_dev) return fail(c, PG_ERR_UNSUPPORTED, "pg_text: the job's samples were merged on the host; format them there");
    const uint32_t ns = c->prm.n_slots;
    const uint64_t ne = R.n_events;
    const uint64_t *d_samp_off, *d_ev_off;
    if (c->batches.size() == 1 && c->have_batch_result && c->cur_n_kept == ne) { d_samp_off = c->samp_off[c->rslot].as<uint64_t>(); d_ev_off = c->ev_off.as<uint64_t>(); } // still there
    else {
        HIP_TRY(c, c->tx_samp_off.ensure((ne + 1) * 8ull)); HIP_TRY(c, c->tx_ev_off.ensure((ns + 1) * 8ull));
        HIP_TRY(c, hipMemcpyAsync(c->tx_samp_off.p, R.samp_off, (ne + 1) * 8ull, hipMemcpyHostToDevice, c->st));
        HIP_TRY(c, hipMemcpyAsync(c->tx_ev_off.p, R.ev_off, (ns + 1) * 8ull, hipMemcpyHostToDevice, c->st));
        d_samp_off = c->tx_samp_off.as<uint64_t>(); d_ev_off = c->tx_ev_off.as<uint64_t>();
    }
    return pg_text_device(c, ns, ne, d_ev_off, d_samp_off, c->fin_dev, out);
}

pg_status pg_fetch_text(pg_ctx *c, uint64_t first, uint64_t n, char *dst) {
    if (!c || (!dst && n)) return PG_ERR_INVALID_ARG;
    if (first > c->tx_bytes || n > c->tx_bytes - first) return PG_ERR_INVALID_ARG;
    if (!n) return PG_OK;
    if (hipSetDevice(c->device) != hipSuccess || hipMemcpy(dst, c->tx_text.as<char>() + first, n, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return PG_ERR_HIP; }
    return PG_OK;
}

// launch + download + host finishing of the model reduction over device arrays in pg_result layout
static pg_status model_run(pg_ctx *c, uint32_t ns, const int any_kind[PG_MODEL_KINDS], const uint64_t *d_ev_off, const uint64_t *d_samp_off,
                           const uint32_t *d_ev_len, const double *d_samples, uint32_t flags, pg_model_result *out) {
    HIP_TRY(c, c->md_out.ensure((ns + 1) * sizeof(PgSlotModel))); HIP_TRY(c, c->md_dwell.ensure((ns + 1) * sizeof(PgSlotDwell)));
    HIP_TRY(c, c->md_class.ensure(pg_slot_model_scratch_bytes(ns)));
    prof_begin(c, "k_slot_model", c->st, true);
    HIP_TRY(c, pg_launch_slot_model(c->st, ns, any_kind, d_ev_off, d_samp_off, d_ev_len, d_samples, (flags & PG_MODEL_KEEP_FIRST) ? 0u : 1u,
                                    c->md_out.as<PgSlotModel>(), c->md_dwell.as<PgSlotDwell>(), c->md_class.p));
    prof_end(c, c->st);
    c->mo_raw.resize(ns); c->mo_dw.resize(ns);
    if (ns) {
        HIP_TRY(c, hipMemcpyAsync(c->mo_raw.data(), c->md_out.p, ns * sizeof(PgSlotModel), hipMemcpyDeviceToHost, c->st));
        HIP_TRY(c, hipMemcpyAsync(c->mo_dw.data(), c->md_dwell.p, ns * sizeof(PgSlotDwell), hipMemcpyDeviceToHost, c->st));
    }
    HIP_TRY(c, hipStreamSynchronize(c->st));
    c->mo_n.resize(ns); c->mo_s2lo.resize(ns); c->mo_s2hi.resize(ns); c->mo_dn.resize(ns); c->mo_lo.resize(ns); c->mo_hi.resize(ns);
    c->mo_origin.resize(ns); c->mo_s1.resize(ns); c->mo_med.resize(ns); c->mo_sd.resize(ns); c->mo_dmed.resize(ns);
    for (uint32_t i = 0; i < ns; i++) {
        const PgSlotModel &m = c->mo_raw[i]; const PgSlotDwell &d = c->mo_dw[i];
        if (d.flags & PG_MODEL_BAD_VALUE) return fail(c, PG_ERR_UNSUPPORTED, "pg_model: slot %u holds a non-finite sample or one with |x| >= 4e7", i);
        if (d.flags & PG_MODEL_BAD_COUNT) return fail(c, PG_ERR_UNSUPPORTED, "pg_model: slot %u holds more than 2^23 values", i);
        if (d.flags & PG_MODEL_BAD_SPREAD) return fail(c, PG_ERR_UNSUPPORTED, "pg_model: slot %u holds values further than 2^40 units of 1e-8 from its first one", i);
        const unsigned __int128 s2 = ((unsigned __int128)m.s2_hh << 40) + ((unsigned __int128)m.s2_hl << 21) + m.s2_ll;
        c->mo_n[i] = m.n; c->mo_lo[i] = m.mid_lo; c->mo_hi[i] = m.mid_hi; c->mo_origin[i] = m.origin; c->mo_s1[i] = m.s1;
        c->mo_s2lo[i] = (uint64_t)s2; c->mo_s2hi[i] = (uint64_t)(s2 >> 64);
        c->mo_med[i] = m.n ? (double)pg_model_median(m) : NAN;
        c->mo_sd[i] = m.n >= 2 ? (double)(pg_model_sstdev_units(m) / 1e8L) : NAN;
        c->mo_dn[i] = d.n; c->mo_dmed[i] = d.n ? ((double)d.mid_lo + (double)d.mid_hi) / 2.0 : NAN;
    }
    out->n_slots = ns; out->flags = flags; out->n_values = c->mo_n.data(); out->median = c->mo_med.data(); out->sstdev = c->mo_sd.data();
    out->mid_lo = c->mo_lo.data(); out->mid_hi = c->mo_hi.data(); out->origin = c->mo_origin.data(); out->sum1 = c->mo_s1.data();
    out->sum2_lo = c->mo_s2lo.data(); out->sum2_hi = c->mo_s2hi.data(); out->dwell_n = c->mo_dn.data(); out->dwell_median = c->mo_dmed.data();
    return PG_OK;
}

pg_status pg_model(pg_ctx *c, uint32_t flags, pg_model_result *out) {
    if (!c || !out) return PG_ERR_INVALID_ARG;
    pg_result R;
    // deferred: the small arrays come to the host, the kept samples stay where they are on the device (after pg_finish_deferred + pg_text
    // of the CLI a plain pg_finish here would download the whole k-mer-major stream and, for several batches, upload it again)
    pg_status s = pg_finish_deferred(c, &R);
    if (s != PG_OK) return s;
    const uint32_t ns = c->prm.n_slots;
    const uint64_t *d_ev_off, *d_samp_off; const uint32_t *d_ev_len; const double *d_samples;
    if (c->batches.size() == 1 && c->have_batch_result && c->cur_n_kept == R.n_events && c->cur_n_samples == R.n_samples) {
        // one batch: its kept events are still on the device, in the same order
        { pg_status su = ensure_unpacked(c); if (su != PG_OK) return su; }
        d_ev_off = c->ev_off.as<uint64_t>(); d_samp_off = c->samp_off[c->rslot].as<uint64_t>(); d_ev_len = c->ev_len.as<uint32_t>(); d_samples = c->samples.as<double>();
    } else { // several batches were merged on the host (slot-major): hand the merged arrays back
        HIP_TRY(c, c->md_ev_off.ensure((ns + 1) * 8ull)); HIP_TRY(c, c->md_samp_off.ensure((R.n_events + 1) * 8ull));
        HIP_TRY(c, c->md_ev_len.ensure(R.n_events * 4ull + 4));
        HIP_TRY(c, hipMemcpyAsync(c->md_ev_off.p, R.ev_off, (ns + 1) * 8ull, hipMemcpyHostToDevice, c->st));
        HIP_TRY(c, hipMemcpyAsync(c->md_samp_off.p, R.samp_off, (R.n_events + 1) * 8ull, hipMemcpyHostToDevice, c->st));
        if (R.n_events) HIP_TRY(c, hipMemcpyAsync(c->md_ev_len.p, R.ev_len, R.n_events * 4ull, hipMemcpyHostToDevice, c->st));
        if (c->fin_dev || !R.n_samples) d_samples = c->fin_dev; // the batches were merged on the device: the stream is there already
        else { // the merge went through the host (no room on the device at the time)
            HIP_TRY(c, c->md_samples.ensure(R.n_samples * 8ull + 8));
            HIP_TRY(c, hipMemcpyAsync(c->md_samples.p, R.samples, R.n_samples * 8ull, hipMemcpyHostToDevice, c->st));
            d_samples = c->md_samples.as<double>();
        }
        d_ev_off = c->md_ev_off.as<uint64_t>(); d_samp_off = c->md_samp_off.as<uint64_t>(); d_ev_len = c->md_ev_len.as<uint32_t>();
    }
    int any_kind[PG_MODEL_KINDS] = {0, 0, 0, 0}; // which of the kernels have work (the host holds the offsets since pg_finish)
    const uint64_t drop = (flags & PG_MODEL_KEEP_FIRST) ? 0 : 1;
    for (uint32_t i = 0; i < ns; i++) {
        const uint64_t all = R.samp_off[R.ev_off[i + 1]] - R.samp_off[R.ev_off[i]], nv = all > drop ? all - drop : 0;
        any_kind[pg_model_kind(nv, R.ev_off[i + 1] - R.ev_off[i])]++; // (counts: pg_launch_slot_model)
    }
    return model_run(c, ns, any_kind, d_ev_off, d_samp_off, d_ev_len, d_samples, flags, out);
}

pg_status pg_model_device(pg_ctx *c, uint32_t n_slots, const uint64_t *d_ev_off, const uint64_t *d_samp_off, const uint32_t *d_ev_len,
                          const double *d_samples, uint32_t flags, pg_model_result *out) {
    if (!c || !out) return PG_ERR_INVALID_ARG;
    if (n_slots && (!d_ev_off || !d_samp_off)) return fail(c, PG_ERR_INVALID_ARG, "pg_model_device: ev_off / samp_off missing");
    HIP_TRY(c, hipSetDevice(c->device));
    const int all_kinds[PG_MODEL_KINDS] = {1 << 20, 1 << 20, 1 << 20, 1 << 20}; // the offsets are not on the host: every kernel looks
    return model_run(c, n_slots, all_kinds, d_ev_off, d_samp_off, d_ev_len, d_samples, flags, out);
}

size_t pg_model_format(const pg_model_result *m, uint32_t slot, int32_t which, char *buf, size_t cap) {
    if (!m || !buf || cap == 0 || slot >= m->n_slots) return 0;
    int w = 0;
    buf[0] = 0;
    if (which == PG_MODEL_TEXT_DWELL) {
        if (m->dwell_n[slot] == 0) return 0;
        w = snprintf(buf, cap, "%.14Lg", (long double)m->dwell_median[slot]); // integers or halves: exact
    } else if (which == PG_MODEL_TEXT_MEDIAN || which == PG_MODEL_TEXT_SSTDEV) {
        if (m->n_values[slot] == 0) return 0; // datamash prints nothing for empty input
        PgSlotModel r{};
        r.n = m->n_values[slot]; r.mid_lo = m->mid_lo[slot]; r.mid_hi = m->mid_hi[slot]; r.s1 = m->sum1[slot];
        if (which == PG_MODEL_TEXT_MEDIAN) w = snprintf(buf, cap, "%.14Lg", pg_model_median(r));
        else if (r.n < 2) w = snprintf(buf, cap, "nan"); // 0/0 in datamash's sample variance
        else {
            const unsigned __int128 s2 = ((unsigned __int128)m->sum2_hi[slot] << 64) | m->sum2_lo[slot];
            const __int128 s1 = r.s1;
            const unsigned __int128 num = (unsigned __int128)r.n * s2 - (unsigned __int128)(s1 * s1);
            w = pg_model_sstdev_text(r.n, num, buf, cap); // "%.14Lg" of the standard deviation, its 14 digits decided in exact arithmetic
        }
    } else return 0;
    return (w < 0 || (size_t)w >= cap) ? 0 : (size_t)w;
}

pg_status pg_kernel_stats(pg_ctx *c, pg_kernel_stat *out, uint32_t cap, uint32_t *n_out) {
    if (!c || !n_out) return PG_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->st));
    if (c->st2) HIP_TRY(c, hipStreamSynchronize(c->st2));
    if (c->st3) HIP_TRY(c, hipStreamSynchronize(c->st3));
    prof_drain(c);
    uint32_t n = 0;
    for (auto &name : c->prof_names) {
        if (out && n < cap) { auto &a = c->prof_acc[name]; out[n].name = name.c_str(); out[n].launches = a.first; out[n].total_ms = a.second; }
        n++;
    }
    if (c->stats_cancelled) { // batches whose statistics the device's rank-level early-out cancelled (pgi_stats_gathered): launched, but no sample read
        if (out && n < cap) { out[n].name = "stats_cancelled_on_device"; out[n].launches = c->stats_cancelled; out[n].total_ms = 0.0; }
        n++;
    }
    if (c->long_reads_split) { // reads of the settled batches whose statistics were cut into slices (PgLongState)
        if (out && n < cap) { out[n].name = "long_reads_split"; out[n].launches = c->long_reads_split; out[n].total_ms = 0.0; }
        n++;
    }
    if (c->long_helpers_short) { // settled batches that wanted more helper waves than were launched (their surplus long reads ran on one wave each)
        if (out && n < cap) { out[n].name = "long_helpers_short_batches"; out[n].launches = c->long_helpers_short; out[n].total_ms = 0.0; }
        n++;
    }
    *n_out = n;
    return PG_OK;
}

pg_status pg_kernel_stats_reset(pg_ctx *c) {
    if (!c) return PG_ERR_INVALID_ARG;
    HIP_TRY(c, hipStreamSynchronize(c->st));
    if (c->st2) HIP_TRY(c, hipStreamSynchronize(c->st2));
    if (c->st3) HIP_TRY(c, hipStreamSynchronize(c->st3));
    prof_drain(c);
    c->prof_acc.clear(); c->prof_names.clear(); c->stats_cancelled = 0; c->long_reads_split = 0; c->long_helpers_short = 0;
    return PG_OK;
}

} // extern "C"
