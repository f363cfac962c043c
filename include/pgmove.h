/* pgmove.h -- C ABI of libpgmove: the MI355X (gfx950) implementation of poregen's `gmove` collector.
 *
 * The reference (hiruna72/poregen) has no plugin/FFI interface; its only seam for this path is the
 * internal C++ function
 *     void process_move_table_paf(char *move_table, std::map<std::string,FILE*>&, slow5_file_t **sp,
 *                                 opt_t*, std::map<std::string,uint64_t>&, std::vector<std::string>& kmers,
 *                                 char *fastq)                                   (src/gmove.cpp:76, 707-975)
 * called once from gmove() (src/gmove.cpp:515). This header is the boundary a maintainer would bind
 * at that seam: the host keeps all I/O (slow5lib / PAF / FASTQ parsing, the k-mer
 * list, output files) and hands batches of parsed reads to the device; the device does the per-read
 * ss walk, event filtering, deterministic first-`sample_limit` selection per k-mer, pA conversion,
 * med-MAD normalisation and the gather of kept windows.
 *
 * Conventions: plain C types only; no exceptions cross the ABI; every call returns a pg_status
 * (0 = ok, <0 = error, text via pg_last_error); the library never calls exit(). One pg_ctx per host
 * thread (like the reference, a context is not thread-safe). There is NO CPU fallback: pg_create
 * fails with PG_ERR_NO_DEVICE when no HIP device is usable.
 *
 * Entry point                      replaces (reference file:line)
 * -------------------------------  ---------------------------------------------------------------
 * pg_default_params                init_opt defaults                      src/poregen.cpp:209-237
 * pg_build_slot_tables             kmer_file_pointer_array/kmer_frequency_map keyed by k-mer string
 *                                                                         src/gmove.cpp:460-477
 * pg_create / pg_destroy           per-run state set up in gmove()        src/gmove.cpp:460-503
 * pg_submit                        one batch of while(getline) iterations src/gmove.cpp:732-969
 * pg_count + pg_collect            the same, split at the "count == sample_limit" test
 *                                  (src/gmove.cpp:925-927) so that several GPUs can exchange
 *                                  per-k-mer counts between the two halves
 * pg_finish                        the bytes the fprintf calls would have produced, as binary
 *                                                                         src/gmove.cpp:938-950
 * pg_finish_deferred /             the same with the samples left on the device, fetched range by range
 * pg_fetch_samples
 * pg_text / pg_fetch_text /        those bytes themselves: fprintf(f, "%.8f,") ... "%.8f;" on the device   src/gmove.cpp:938-944
 * pg_text_device
 * pg_all_slots_full                the early loop exit                    src/gmove.cpp:733-735
 * pg_model / pg_model_device /     the step behind gmove in the reference's pipeline: dump files -> tr | tail | datamash
 * pg_model_format                  median / sstdev per k-mer, awk | datamash median of the dwell times
 *                                                                         scripts/poregen.sh:54-85, 33-52
 * pg_job_create / _submit / _sync  a batch split over the node's GPUs from one process: the shape of the reference's only
 * / _finish / _all_slots_full /    parallel driver, work_db (a batch split over worker threads)   src/thread.c:119-132,
 * _model / _destroy                plugged in at the same seam as pg_submit                       src/gmove.cpp:515
 * pg_job_finish_deferred /         the job's output side, as the pg_ctx calls of the same names: the shards' kept samples
 * _fetch_samples / _text /         concatenated on the first device (peer copies over xGMI), text produced there
 * _fetch_text                                                                                     src/gmove.cpp:938-950
 * pg_set_stream / pg_sync /        (no counterpart: the reference is synchronous and single-threaded)
 * pg_runtime_init / pg_poll / pg_all_slots_full_settled / pg_last_batch_device / pg_kernel_stats*
 */
#ifndef PGMOVE_H
#define PGMOVE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t pg_status;
enum {
    PG_OK = 0,
    PG_ERR_NO_DEVICE = -1,    /* no usable HIP device / HIP runtime error at create */
    PG_ERR_INVALID_ARG = -2,  /* bad parameter or malformed batch layout */
    PG_ERR_INPUT = -3,        /* a read is outside the reference's defined behaviour (see pg_last_error) */
    PG_ERR_RNA_FLAG = -4,     /* RNA-oriented record without allow_rna (src/gmove.cpp:795-797) */
    PG_ERR_HIP = -5,          /* HIP runtime failure */
    PG_ERR_STATE = -6,        /* calls made in the wrong order */
    PG_ERR_UNSUPPORTED = -7   /* valid input this build cannot process (reported, never silently wrong) */
};

enum { PG_LOC_HOST = 0, PG_LOC_DEVICE = 1 };

/* pg_params.flags */
enum {
    PG_FLAG_LAZY_STATS = 1u << 0, /* compute median/MAD only for reads that contribute a kept event
                                     (legal: event acceptance is signal-independent in the PAF path);
                                     default is to touch every read's signal like the reference does */
    PG_FLAG_PROFILE = 1u << 1,    /* record HIP events around every kernel (pg_kernel_stats) */
    PG_FLAG_ONE_STREAM = 1u << 10, /* every kernel of a batch on ONE stream. Since round 3 the DEFAULT is the two-stream mode below
                                     (PG_FLAG_OVERLAP) whenever the context computes eager statistics and none of PG_FLAG_PROFILE /
                                     _LAZY_STATS / _SKIP_OUT_OF_RANGE / _DEFER_STATS / _OVERLAP_TAIL is set: batches that follow each other
                                     run 14-17 % faster (0.161 -> 0.139 ms per 50 000-read batch). This flag keeps the one-stream form:
                                     short jobs (a hardware queue takes 15-20 ms to create: the CLI sets it), clean per-kernel timings. */
    PG_FLAG_OVERLAP = 1u << 2,    /* two-stream mode, THE DEFAULT since round 3 (see PG_FLAG_ONE_STREAM for when it applies): the statistics
                                     kernels of batch i+1 run on a second stream next to the event / rank / emit chain of batches i and
                                     i+1; that stream is created with a quarter of the compute units (of every XCD) withheld, so that the
                                     chain's workgroups find room. Pays when batches follow each other (configs[1]: 0.162 -> 0.141 ms per
                                     batch). Every kernel then shares the chip: per-kernel timings (PG_FLAG_PROFILE, bench.py's roofline
                                     object) are taken on one stream. Setting the flag explicitly asks for the mode where it is not the
                                     default; PGMOVE_GATHER_SIDE=1 additionally puts the gather of batch i beside the chain of batch i+1
                                     (measured: 0-5 %, profiles/r04_side_gather.txt). */
    PG_FLAG_SHORT_READS_OK = 1u << 4, /* a read with fewer than k matched bases simply has no events (move-table front-end,
                                        where that is well defined); default: PG_ERR_INPUT, because the PAF path of the
                                        reference has undefined behaviour there (src/gmove.cpp:891) */
    PG_FLAG_SKIP_OUT_OF_RANGE = 1u << 5, /* a read with ANY sample outside [pa_min, pa_max] is skipped as a whole instead of
                                           zero-filling the sample: the SAM/BAM front-end (src/gmove.cpp:1149-1160). Event
                                           acceptance then depends on the signal, so the statistics pass runs first. */
    PG_FLAG_DEBUG_NARROW = 1u << 3, /* tests: shrink the exact MAD candidate window to one code so that the fallback search runs */
    PG_FLAG_OVERLAP_TAIL = 1u << 9, /* the statistics of a batch on a second stream, forked BEHIND the walk and the counting kernels:
                                     * the small launches of pg_collect (sample_limit cut, emit, offset scan) run next to the
                                     * streaming kernel; the streams join in front of the gather. Ignored with PG_FLAG_LAZY_STATS,
                                     * PG_FLAG_SKIP_OUT_OF_RANGE, PG_FLAG_OVERLAP and PG_FLAG_DEFER_STATS. */
    PG_FLAG_DEBUG_SPLIT_WALK = 1u << 8, /* tests / measurement: every read takes the generic wave-per-read walk (k_walk), also the reads of
                                        * matches only that the op-parallel event kernel (k_events) would handle */
    PG_FLAG_DEFER_STATS = 1u << 7, /* multi-GPU step: pg_count does not queue the per-read statistics (median/MAD of every read, which
                                     * do not depend on the exchange); pg_stats queues them -- between the ISSUE of the caller's
                                     * collective and the wait for it, so that the all_gather's latency hides behind the streaming
                                     * kernel -- or, if pg_stats is not called, pg_collect does. Ignored with PG_FLAG_LAZY_STATS,
                                     * PG_FLAG_SKIP_OUT_OF_RANGE and PG_FLAG_OVERLAP (their statistics are placed already). */
    PG_FLAG_STOP_WHEN_FULL = 1u << 6 /* the slots are the WHOLE k-mer list: the reference stops reading PAF lines once every k-mer is
                                     * complete (src/gmove.cpp:733-735), so a read behind the one that completes the last k-mer
                                     * is never looked at and cannot fail the job. With this flag a per-read input error is
                                     * reported only if the reference would have reached that read. Without it (a slice of the
                                     * list: the reference reads every line) every read of every batch counts. */
};

typedef struct pg_ctx pg_ctx;

/* Mirrors the gmove-relevant members of opt_t (src/poregen.h:48-79). */
typedef struct {
    uint32_t struct_size;         /* sizeof(pg_params), for ABI evolution */
    uint32_t kmer_size;           /* -k                       default 9 */
    uint32_t sig_move_offset;     /* -m                       default 0; must be <= kmer_size */
    uint32_t signal_print_margin; /* --margin                 default 0 */
    uint32_t sample_limit;        /* --sample_limit           default 100 */
    uint32_t max_dur;             /* --max_dur                default 70 */
    uint32_t min_dur;             /* --min_dur                default 5 */
    int32_t  kmer_pick_margin;    /* --kmer_pick_margin       default 2; must be >= 0 */
    int32_t  scaling;             /* --scaling: 0 none, 1 med-MAD (effective default 0) */
    int32_t  allow_rna;           /* --rna */
    double   pa_min;              /* --pa_min                 default 40.0 */
    double   pa_max;              /* --pa_max                 default 180.0 */
    uint32_t n_slots;             /* number of k-mers in the slice [index_start, index_end] */
    uint32_t flags;               /* PG_FLAG_* */
    int32_t  device;              /* HIP device ordinal */
    int32_t  reserved;
    /* code -> slot tables, int32[4^k], host memory, copied at pg_create; -1 = k-mer not in the slice.
     * The code of a k-mer is its base-4 value, first base most significant, A=0 C=1 G=2 T/U=3.
     * table_t is used for DNA-oriented records (sequence spelled with T), table_u for RNA-oriented
     * records (T->U applied, src/gmove.cpp:815-817). pg_build_slot_tables fills both. */
    const int32_t *table_t;
    const int32_t *table_u;
} pg_params;

/* One batch of reads in PAF line order, structure-of-arrays. All arrays are caller-owned. LIFETIME: pg_count / pg_submit
 * return with work still queued -- a device batch is read by kernels queued behind the call, a host batch is staged with
 * asynchronous copies out of the caller's memory -- so the arrays must stay valid and unmodified until pg_sync (or pg_finish /
 * pg_all_slots_full / the next pg_count / pg_submit on the context, which synchronise) has returned. `location` says whether
 * every pointer is a host or a device pointer (device pointers must belong to pg_params.device). sig must be 16-byte aligned.
 * A device batch must be complete on the context's stream (pg_set_stream) before the call, or be produced on that stream.
 *
 * Read r:  signal   sig[sig_off[r] .. sig_off[r+1])               slow5_rec_t.raw_signal
 *          digitisation/offset/range[r]                            slow5_rec_t fields
 *          query_start[r], target_start[r], target_end[r]          PAF columns 3, 8, 9
 *          seq[seq_off[r] .. seq_off[r+1])  = faidx_fetch_seq(tid, min(ts,te), max(ts,te)-1) as
 *                                             fetched (ASCII, no T->U); empty if the name is absent
 *          ss ops  op_n/op_t[op_off[r] .. op_off[r+1]): op_t 0=',' (match) 1='I' 2='D'
 */
typedef struct {
    uint32_t struct_size;
    int32_t  location;
    uint32_t n_reads;
    uint32_t n_ops;               /* device batches: op_off[n_reads] if the caller knows it (it sizes the work buffers without a
                                   * round trip to the device); 0 = the library reads it back (one synchronisation of the
                                   * context's stream per call). A wrong value is detected on the device and fails the batch
                                   * with PG_ERR_INVALID_ARG; no kernel touches memory behind n_ops. Host batches: ignored. */
    const int16_t  *sig;
    const uint64_t *sig_off;      /* [n_reads+1] */
    const double   *digitisation; /* [n_reads] */
    const double   *offset;
    const double   *range;
    const int32_t  *query_start;
    const int32_t  *target_start;
    const int32_t  *target_end;
    const uint8_t  *seq;
    const uint64_t *seq_off;      /* [n_reads+1] */
    const uint32_t *op_n;
    const uint8_t  *op_t;
    const uint64_t *op_off;       /* [n_reads+1] */
    uint32_t flags;               /* PG_BATCH_* */
    uint32_t reserved;
} pg_batch;
/* pg_batch.flags */
enum {
    PG_BATCH_ALL_MATCHES = 1u << 0 /* the caller vouches that every ss op of the batch is a match (what `reform` writes): the generic
                                    * wave-per-read walk is not launched at all. Verified on the device: a batch that holds an I, a D or
                                    * an unknown op after all fails with PG_ERR_INVALID_ARG (never a wrong result). */
    ,
    PG_BATCH_RESIDENT = 1u << 1    /* device batches of a context that runs on the CALLER's stream (pg_set_stream): the batch's arrays are complete
                                    * -- nothing queued on that stream produces them (a shard uploaded once and synchronised, as in a multi-GPU
                                    * job's steady state). Without it the statistics stream waits for everything the caller's stream holds in
                                    * front of pg_count, which includes the previous batch's gather: the statistics of a batch then start a whole
                                    * chain later than on the context's own stream (156 -> 13x us per step of the one-rank RCCL step). */
};

/* Host-side view of everything collected so far, in reference order: for slot s, its kept events are
 * e in [ev_off[s], ev_off[s+1]); event e has ev_len[e] samples at samples[samp_off[e] ..], and came
 * from read ev_read[e] (0-based index over all reads submitted to this context, in order).
 * Memory is owned by the context and valid until the next pg_submit/pg_collect/pg_reset/pg_destroy. */
typedef struct {
    uint32_t n_slots;
    uint32_t reserved;
    uint64_t n_events;
    uint64_t n_samples;
    uint64_t n_reads;             /* reads submitted so far */
    const uint64_t *counts;       /* [n_slots]   freq.txt values (<= sample_limit) */
    const uint64_t *ev_off;       /* [n_slots+1] */
    const uint32_t *ev_len;       /* [n_events]  */
    const uint32_t *ev_read;      /* [n_events]  */
    const uint64_t *samp_off;     /* [n_events+1] */
    const double   *samples;      /* [n_samples] */
    const uint8_t  *read_skipped; /* [n_reads] 1 = silently skipped (fetched length < k, gmove.cpp:806-808) */
} pg_result;

typedef struct {
    const char *name;     /* kernel name */
    uint64_t launches;
    double   total_ms;    /* HIP-event time on the launching stream */
} pg_kernel_stat;

void        pg_default_params(pg_params *p);
const char *pg_last_error(const pg_ctx *ctx); /* ctx may be NULL: error of the last failed pg_create */
const char *pg_version(void);

/* Fill int32[4^k] tables from the slice's k-mer strings (slot i = kmers[i]). K-mers containing
 * characters outside ACGTU, or both T and U, can never equal a fetched window and get no entry.
 * Returns PG_ERR_INVALID_ARG on duplicate k-mers or k > 13. */
pg_status pg_build_slot_tables(uint32_t kmer_size, const char *const *kmers, uint32_t n_slots,
                               int32_t *table_t, int32_t *table_u);

/* Brings the HIP runtime up on `device` (the first call in a process takes 0.1-0.2 s): a host that has other start-up work -- file
 * indices, parsing -- calls this on a thread of its own first; pg_create afterwards finds the runtime ready. Optional. */
pg_status pg_runtime_init(int32_t device);
pg_status pg_create(const pg_params *params, pg_ctx **out);
void      pg_destroy(pg_ctx *ctx);
pg_status pg_reset(pg_ctx *ctx); /* forget all reads/events, keep parameters and buffers */

/* count + collect with base = events accepted by earlier batches of this context */
pg_status pg_submit(pg_ctx *ctx, const pg_batch *batch);

/* Phase 1: walk/filter/rank one batch; writes the batch's accepted-event count per slot (uncapped)
 * to counts_out (uint64[n_slots]) in host or device memory. */
pg_status pg_count(pg_ctx *ctx, const pg_batch *batch, uint64_t *counts_out, int32_t counts_location);
/* Phase 2: keep the events whose rank (base[slot] + rank inside the batch) is < sample_limit and
 * gather their windows. base = uint64[n_slots], number of accepted events that precede this batch in
 * reference order (earlier batches, lower ranks of a multi-GPU job); NULL = the context's own running
 * count. The batch pointers given to pg_count must still be valid. */
pg_status pg_collect(pg_ctx *ctx, const uint64_t *base, int32_t base_location);
/* Between pg_count and pg_collect of a context created with PG_FLAG_DEFER_STATS: queue the statistics of the counted batch
 * (src/gmove.cpp:754-771 for every read) on the context's stream now. A no-op when nothing was deferred. */
pg_status pg_stats(pg_ctx *ctx);
/* Phase 2 of a multi-GPU job, straight from the all-gather: all_counts = uint64[world][n_slots] in DEVICE memory, row g =
 * pg_count's output of rank g (the receive buffer of the all_gather). base = sum of the rows below `rank`, computed on
 * the device on the context's stream: the step needs no host arithmetic and no extra kernels of the caller.
 * Replaces nothing in the reference (which is single-process); the order semantics are those of pg_collect. */
pg_status pg_collect_gathered(pg_ctx *ctx, const uint64_t *all_counts, uint32_t world, uint32_t rank);

/* After pg_collect_gathered: device pointers (uint64[n_slots], stable for the life of the context, contents rewritten by
 * every pg_collect_gathered on the context's stream) to the accepted events of ALL ranks per slot and to the job's freq.txt
 * values min(total, sample_limit) (src/gmove.cpp:945-953, 523-534) -- produced by the kernel that sums the lower ranks' rows,
 * so a multi-GPU step needs no reduction of its own. Either output pointer may be NULL. */
pg_status pg_job_totals_device(pg_ctx *ctx, const uint64_t **d_total, const uint64_t **d_freq);

pg_status pg_sync(pg_ctx *ctx);                      /* wait for all device work of the context */
/* Run the context's main chain on a caller-owned HIP stream (hipStream_t passed as void*), e.g. PyTorch's current
 * stream, so that collectives issued by the caller between pg_count and pg_collect are ordered without host
 * synchronisation. NULL restores the context's own stream. */
pg_status pg_set_stream(pg_ctx *ctx, void *hip_stream);
pg_status pg_finish(pg_ctx *ctx, pg_result *out);     /* sync, copy results to host, merge batches */
/* pg_finish without the samples' trip to the host: every array of pg_result is filled except `samples` (NULL whenever the kept samples
 * are still on the device: one batch, or several merged there). pg_fetch_samples then copies any range [first, first + n) of the
 * job's k-mer-major sample stream (the indices samp_off counts in) into a host buffer of the caller's; after pg_finish_deferred it
 * may be called from SEVERAL host threads at once, so that a writer formats and writes one k-mer range while the next one is on its
 * way -- the reference prints as it goes (src/gmove.cpp:938-944); the CLI's dump writers do this. Valid until the next
 * pg_submit / pg_count / pg_reset. A later pg_finish still hands the samples out as a whole. */
pg_status pg_finish_deferred(pg_ctx *ctx, pg_result *out);
pg_status pg_fetch_samples(pg_ctx *ctx, uint64_t first, uint64_t n, double *dst);

/* The dump files' TEXT, produced on the device: for every slot the bytes the reference's fprintf calls write into dump/<KMER> --
 * "%.8f," per sample, "%.8f;" for an event's last one (src/gmove.cpp:938-944) -- as one buffer in HBM, slot after slot; slot s is
 * bytes [slot_off[s], slot_off[s+1]). Without -d only (the ':' of src/gmove.cpp:960-962 depend on the reads: a host job). The digits
 * are printf's (correctly rounded, ties to even on the binary value). PG_ERR_UNSUPPORTED -- never a wrong digit -- if a kept sample is
 * not finite or |sample| >= 4e7, or if the samples of several batches had to be merged on the host: format there (pg_finish).
 * Calls pg_finish_deferred first; pg_fetch_text copies a byte range to the host and may be called from several threads at once.
 * Valid until the next pg_submit / pg_count / pg_reset / pg_text. */
typedef struct {
    uint32_t n_slots, reserved;
    uint64_t n_bytes;
    const uint64_t *slot_off; /* [n_slots + 1], host memory owned by the context */
} pg_text_result;
pg_status pg_text(pg_ctx *ctx, pg_text_result *out);
pg_status pg_fetch_text(pg_ctx *ctx, uint64_t first, uint64_t n, char *dst);
/* The same from DEVICE arrays in pg_result layout (what pg_last_batch_device or poregen_amd.dist.gather_kept hand out; pg_job_text's
 * merged view): n_events kept events, d_ev_off[n_slots + 1], d_samp_off[n_events + 1], d_samples on pg_params.device. The text is the
 * context's (pg_fetch_text), the arrays stay the caller's. */
pg_status pg_text_device(pg_ctx *ctx, uint32_t n_slots, uint64_t n_events, const uint64_t *d_ev_off, const uint64_t *d_samp_off,
                         const double *d_samples, pg_text_result *out);
int32_t   pg_all_slots_full(pg_ctx *ctx);            /* 1 when every slot holds sample_limit events (waits for the device) */
/* The same as of the last batch the context has already waited for (pg_submit / pg_count wait for the PREVIOUS batch): no wait.
 * A host that parses batch i+1 while batch i is on the device asks this after submitting i+1 and learns about batch i. */
int32_t   pg_all_slots_full_settled(const pg_ctx *ctx);
/* Has the device finished the last submitted batch? 1 = yes (the batch is then settled: its per-read errors are returned as a
 * negative pg_status, pg_all_slots_full_settled speaks about it), 0 = still running. Never waits. */
int32_t   pg_poll(pg_ctx *ctx);

/* device-resident view of the LAST collected batch (for callers that keep results on the GPU) */
typedef struct {
    uint64_t n_events, n_samples;
    const uint64_t *d_keep;     /* [n_slots] kept events of this batch per slot */
    const uint64_t *d_ev_off;   /* [n_slots+1] */
    const uint32_t *d_ev_len;
    const uint32_t *d_ev_read;  /* read index inside the batch */
    const uint64_t *d_samp_off;
    const double   *d_samples;
    const double   *d_med;      /* [n_reads] (scaling==1; NaN where not computed) */
    const double   *d_mad;
} pg_device_view;
pg_status pg_last_batch_device(pg_ctx *ctx, pg_device_view *out);

/* Per-k-mer model over everything collected so far, computed on the device from the kept samples: what the
 * reference's pipeline gets by reading the dump files back as text (scripts/poregen.sh:54-85: `tr ';,' '\n' < file |
 * tail -n +2 | datamash median 1` and `... | datamash sstdev 1`; :33-52: awk comma counts | datamash median).
 * The values are those of the "%.8f" TEXT gmove writes (src/gmove.cpp:941-944), i.e. integers of 1e-8 units:
 *   - the first value of every file is dropped (`tail -n +2`) unless PG_MODEL_KEEP_FIRST is given;
 *   - median = datamash's: middle value, or the mean of the two middle values;  sstdev = sqrt(sum (x-mean)^2/(n-1));
 *   - dwell  = median over the file's ';'-separated fields of the number of commas: samples-1 per event, plus one 0
 *     for the empty field behind the last ';' (files without events have no fields at all).
 * Calls pg_finish first. Arrays are owned by the context, valid until the next pg_model/pg_reset/pg_destroy.
 * PG_ERR_UNSUPPORTED (never a wrong number) if a slot holds a non-finite sample or |sample| >= 4e7, more than 2^23
 * values, or values further than 2^40 units (10995.1 pA) from its first one. */
enum { PG_MODEL_KEEP_FIRST = 1u << 0 };
typedef struct {
    uint32_t n_slots;
    uint32_t flags;
    const uint64_t *n_values;     /* [n_slots] values that count (0: the fields below are NaN / 0) */
    const double   *median;       /* [n_slots] */
    const double   *sstdev;       /* [n_slots] NaN when n_values < 2 */
    const int64_t  *mid_lo;       /* [n_slots] the two middle order statistics, exact, in 1e-8 units */
    const int64_t  *mid_hi;
    const int64_t  *origin;       /* [n_slots] exact moments of d = value - origin (1e-8 units): */
    const int64_t  *sum1;         /*           sum d                                              */
    const uint64_t *sum2_lo;      /*           sum d*d, low and high 64 bits                      */
    const uint64_t *sum2_hi;
    const uint64_t *dwell_n;      /* [n_slots] fields awk sees (kept events + 1), 0 for an empty file */
    const double   *dwell_median; /* [n_slots] */
} pg_model_result;
enum { PG_MODEL_TEXT_MEDIAN = 0, PG_MODEL_TEXT_SSTDEV = 1, PG_MODEL_TEXT_DWELL = 2 };
pg_status pg_model(pg_ctx *ctx, uint32_t flags, pg_model_result *out);
/* The same reduction over caller-owned DEVICE arrays in pg_result layout: ev_off uint64[n_slots+1], samp_off
 * uint64[ev_off[n_slots]+1], ev_len uint32[], samples double[] -- e.g. the kept events of all ranks of a multi-GPU job
 * brought together on the writing rank (poregen_amd/dist.py gather_kept). Uses the context's stream and result buffers;
 * the arrays must be complete before the call (no ordering with other streams is implied). */
pg_status pg_model_device(pg_ctx *ctx, uint32_t n_slots, const uint64_t *d_ev_off, const uint64_t *d_samp_off,
                          const uint32_t *d_ev_len, const double *d_samples, uint32_t flags, pg_model_result *out);
/* The number as datamash prints it ("%.14Lg" of its long double; "nan" for sstdev of one value; empty string when the
 * slot has no value at all, like datamash on empty input). Returns the length written (excluding the NUL), 0 on error.
 * Caveat (parity of this call is UNPINNED: datamash is not in the image): the 14 digits of the sample standard deviation are
 * decided here in exact integer arithmetic on the "%.8f" values. datamash rounds a long double that it computed with rounded
 * intermediate sums; on a file whose exact sstdev lies within a few 1e-19 (relative) of a 14th-digit rounding boundary -- the
 * fuzzer found one 5e-9 of a 14th-digit unit below it -- the real pipeline (scripts/poregen.sh:54-85) may print the neighbouring
 * digit. Undecidable without datamash; the exact value is what this returns. */
size_t pg_model_format(const pg_model_result *m, uint32_t slot, int32_t which, char *buf, size_t cap);

/* ---- one job over several GPUs of one node, driven from ONE host process ---------------------------------------------
 * The reference is a single process (src/main.c:64-103 -> gmove(), src/gmove.cpp:213) whose only parallel driver is
 * work_db (src/thread.c:119-132: a batch split over worker threads, each working a contiguous range); pg_job is that
 * shape with GPUs as the workers, at the same seam as pg_submit (src/gmove.cpp:515, 732-969): a batch's reads are cut into
 * n contiguous PAF-ordered shards, device i works shard i on its own host thread, and the "count == sample_limit" test
 * (src/gmove.cpp:925-927) is resolved by ONE exchange per batch -- an ncclAllGather (RCCL over xGMI) of every shard's
 * uint64[n_slots] accepted-event counts, in place in each device's receive buffer, whose rows below a shard (plus the
 * running total of the earlier batches, kept as row 0 of the same buffer) are its base (pg_collect_gathered). The per-read
 * statistics are queued behind the issue of the collective and hide it. pg_job_finish returns the job's per-k-mer streams
 * in reference order (slot, then batch, then shard): byte for byte what one context fed the same batches returns.
 * pg_job_submit takes a host batch and cuts it; pg_job_submit_shards takes one batch per device, host or device-resident. A device may be listed more than once (several shards on one GPU); the exchange then goes through
 * host memory (RCCL needs distinct devices), as it does when librccl cannot be loaded and PG_JOB_EXCHANGE_RCCL is not set. */
typedef struct pg_job pg_job;
enum {
    PG_JOB_EXCHANGE_AUTO = 0, /* RCCL when the listed devices are distinct and librccl loads, else through the host */
    PG_JOB_EXCHANGE_HOST = 1, /* always through host memory */
    PG_JOB_EXCHANGE_RCCL = 2  /* always RCCL: pg_job_create fails when it is unavailable or a device is listed twice */
};
/* params->device is ignored (devices[] decides); params->flags as for pg_create (PG_FLAG_DEFER_STATS is added). */
pg_status   pg_job_create(const pg_params *params, const int32_t *devices, uint32_t n_devices, uint32_t exchange, pg_job **out);
void        pg_job_destroy(pg_job *job);
const char *pg_job_last_error(const pg_job *job);       /* job may be NULL: error of the last failed pg_job_create */
/* Queues the whole batch on all devices and returns; the batch arrays must stay valid until pg_job_sync / the next
 * pg_job_submit / pg_job_finish has returned (see pg_batch). */
pg_status   pg_job_submit(pg_job *job, const pg_batch *host_batch);
/* The same step for shards that already are where they will be worked: shards[g] (g < n_devices, in PAF order) is rank g's part of the
 * batch as a pg_batch of its own -- PG_LOC_DEVICE arrays resident on devices[g] and complete before the call, or PG_LOC_HOST. Nothing is
 * cut or copied on the host and, with device shards, nothing crosses PCIe inside the step: a device-resident N-GPU step driven from the
 * C++ host (src/gmove.cpp:732-969 over N devices; lifetime of the arrays as pg_job_submit). Results: those of pg_job_submit on the
 * concatenation of the shards. A PG_LOC_DEVICE shard whose sig / sig_off / op_n / seq are not device memory on devices[g] is refused with
 * PG_ERR_INVALID_ARG (hipPointerGetAttributes, once per array and call) before anything is launched on it.
 * Where a rank's statistics are queued relative to the count exchange is a PROVISIONAL rule (csrc/pg_job_rule.h; no run on more than one
 * GPU exists yet): PGMOVE_JOB_STATS_RULE=front|behind|auto in the environment overrides it; results never depend on it. */
pg_status   pg_job_submit_shards(pg_job *job, const pg_batch *shards, uint32_t n_shards);
pg_status   pg_job_reset(pg_job *job);                  /* as pg_reset: the next submit starts a new job on the same devices and communicators */
pg_status   pg_job_sync(pg_job *job);                   /* wait for every device; surfaces per-read errors (lowest shard first) */
int32_t     pg_job_all_slots_full(pg_job *job);         /* src/gmove.cpp:733-735 for the job */
int32_t     pg_job_all_slots_full_settled(const pg_job *job); /* as pg_all_slots_full_settled */
int32_t     pg_job_poll(pg_job *job);                    /* as pg_poll, for every device of the job */
pg_status   pg_job_finish(pg_job *job, pg_result *out); /* merged view, owned by the job until the next submit / destroy */
/* As pg_finish_deferred / pg_fetch_samples / pg_text / pg_fetch_text for the job. The shards' kept samples are concatenated ON THE
 * JOB'S FIRST DEVICE -- one peer copy per shard that sits on another device (xGMI), one launch of (k-mer, batch, shard) segment copies:
 * north_star's "concatenate the buffers over xGMI" -- and stay there: pg_job_finish_deferred returns everything but `samples` (NULL),
 * pg_job_fetch_samples copies any range of the merged stream to the host (callable from several threads at once), pg_job_text
 * produces the dump files' bytes from it on that device (PG_ERR_UNSUPPORTED as pg_text), pg_job_model reduces it there.
 * pg_job_finish = the same merge + ONE download. (A shard whose own batches had to be merged on the host sends the job through the
 * host merge instead; the results are the same.) */
pg_status   pg_job_finish_deferred(pg_job *job, pg_result *out);
pg_status   pg_job_fetch_samples(pg_job *job, uint64_t first, uint64_t n, double *dst);
pg_status   pg_job_text(pg_job *job, pg_text_result *out);
pg_status   pg_job_fetch_text(pg_job *job, uint64_t first, uint64_t n, char *dst);
/* 1 when the last pg_job_create chose RCCL for this job's exchange, 0 = host memory */
int32_t     pg_job_uses_rccl(const pg_job *job);
/* The k-mer model of the whole job (see pg_model): the merged kept samples are reduced on the job's first device. */
pg_status   pg_job_model(pg_job *job, uint32_t flags, pg_model_result *out);
/* pg_kernel_stats of one shard's context (PG_FLAG_PROFILE in the job's params): what a rank-level early-out skipped shows up here */
pg_status   pg_job_kernel_stats(pg_job *job, uint32_t shard, pg_kernel_stat *out, uint32_t cap, uint32_t *n_out);

/* profiling (PG_FLAG_PROFILE): per-kernel launch counts and HIP-event times since the last reset */
pg_status pg_kernel_stats(pg_ctx *ctx, pg_kernel_stat *out, uint32_t cap, uint32_t *n_out);
pg_status pg_kernel_stats_reset(pg_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
