/* gmove_oracle.h -- CPU restatement of poregen's `gmove` collector (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle for the MI355X-native gmove hot path. It is a plain-C, single-threaded,
 * FP64 restatement of the reference algorithm, written from the reference's behaviour (file:line
 * citations are relative to /root/reference/ and are given on every function in gmove_oracle.c).
 * Nothing in the product (poregen_amd/, include/, the `poregen` CLI) links, imports or calls this
 * code. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and only
 * as the checker / the CPU baseline.
 *
 * PINNING STATUS: the reference holds NO expected outputs for gmove (test/test_gmove.sh only checks
 * exit codes and table==PAF==BAM self-consistency; its EXP_DIR does not exist). The reference is also
 * unbuildable in this image (slow5lib is an empty submodule, htslib 1.17 is downloaded by its
 * Makefile; no stand-ins were written). The oracle is therefore pinned by
 *   (1) the known-answer vectors recorded in SURVEY.md Appendix C (KA-1..KA-7), run on the reference's
 *       own input fixtures (copied as data under tests/golden/single_read/),
 *   (2) the reference tests' cross-format invariant (table path == PAF path at --kmer_pick_margin 0,
 *       test/test_gmove.sh:79-80,95-96), and
 *   (3) the reference tests' exit-status expectations (test_gmove.sh:50,58,66);
 *   (4) for the SAM/BAM path only the invariant BAM == table (test_gmove.sh:85-86,101-102): no known answers exist;
 *   (5) round 6, for orc_median / orc_madf and the zero-filled pA vector in front of them (gmove.cpp:142-184, 751-771) ONLY: the
 *       reference's own quickselect -- src/ksort.h:233-259, the one file of the path that compiles here -- built where it lies into
 *       oracle/_ref/libref_selection.so (ref_selection.c) and run on 216 seeded reads: tests/golden/ksmall_vectors.json.
 * No reference-held golden output exists for the path as a whole, so by the rules of this build: "parity unpinned" with
 * respect to reference-owned expected values, except the selection (5); see DESIGN.md 6.
 */
#ifndef GMOVE_ORACLE_H
#define GMOVE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Mirrors the gmove-relevant fields of opt_t (src/poregen.h:48-79) after gmove()'s option
 * reconciliation (src/gmove.cpp:428-440, 479-491). index_start/index_end are the final 1-based
 * closed slice into the k-mer list. */
typedef struct {
    uint32_t kmer_size;           /* -k, default 9 (poregen.h:38) */
    uint32_t sig_move_offset;     /* -m, default 0 */
    uint32_t kmer_start_offset;   /* -s, default 0 (table path only) */
    int32_t  scaling;             /* 0 none, 1 med-MAD (effective default 0: gmove.cpp:229,479-484) */
    uint32_t signal_print_margin; /* --margin, default 0 */
    uint32_t sample_limit;        /* default 100 */
    uint32_t index_start;         /* 1-based, closed */
    uint32_t index_end;           /* 1-based, closed */
    int32_t  delimit_files;       /* -d */
    uint32_t max_dur;             /* default 70 */
    uint32_t min_dur;             /* default 5 */
    double   pa_min;              /* default 40.0 */
    double   pa_max;              /* default 180.0 */
    int32_t  kmer_pick_margin;    /* default 2 (PAF path only) */
    int32_t  flag_rna;            /* --rna */
} orc_opt_t;

void orc_default_opt(orc_opt_t *opt); /* init_opt, src/poregen.cpp:209-237 (+ effective scaling 0) */

/* status codes of the per-read functions; negative = the reference would exit()/abort here */
enum {
    ORC_OK = 0,
    ORC_SKIPPED = 1,              /* read silently skipped (gmove.cpp:806-808, 616-619) */
    ORC_STOPPED = 2,              /* every k-mer of the FULL list is complete: loop breaks (gmove.cpp:733-735) */
    ORC_ERR_RNA_FLAG = -1,        /* RNA-oriented record without --rna (gmove.cpp:795-797) */
    ORC_ERR_BAD_SS = -2,          /* ss syntax error (gmove.cpp:834,838,867) */
    ORC_ERR_INTERNAL = -3,        /* "This should not have happened" (gmove.cpp:893) */
    ORC_ERR_ASSERT = -4,          /* assert() in the reference fails (gmove.cpp:752, 589-590) */
    ORC_ERR_UNDEFINED = -5        /* the reference has undefined behaviour on this input */
};

typedef struct orc_state orc_state_t;

/* kmers: the full k-mer list in list order (generated or from --kmer_file). */
orc_state_t *orc_create(const orc_opt_t *opt, const char *const *kmers, size_t n_kmers);
void orc_destroy(orc_state_t *st);

/* generate_kmers(), src/poregen.cpp:248-267: lexicographic 4^k strings over ACGT / ACGU.
 * Returns a malloc'd array of malloc'd strings; free with orc_free_kmers. */
char **orc_generate_kmers(int k, int rna, size_t *n_out);
void orc_free_kmers(char **kmers, size_t n);

/* One iteration of the while(getline) loop of process_move_table_paf (gmove.cpp:732-969).
 * target_seq/target_len: the whole FASTQ sequence of PAF column 6, or NULL if the name is absent
 * (the oracle applies faidx_fetch_seq's [beg,end] clamping itself). */
int orc_paf_read(orc_state_t *st,
                 const int16_t *raw, uint64_t len_raw_signal,
                 double digitisation, double offset, double range,
                 int32_t query_start, int32_t target_start, int32_t target_end,
                 const char *target_seq, int64_t target_len,
                 const char *ss);

/* One iteration of the while(getline) loop of process_move_table_file (gmove.cpp:557-700). */
int orc_table_read(orc_state_t *st,
                   const int16_t *raw, uint64_t len_raw_signal,
                   double digitisation, double offset, double range,
                   int32_t fastq_len, const char *fastq_seq, int32_t stride,
                   const char *move_seq, uint64_t signal_len, int32_t trim_offset);

/* One iteration of the while(sam_read1) loop of process_move_table_bam (gmove.cpp:1080-1261).
 * moves[0..n_moves) are the mv:B:c values AFTER the leading stride element (bam_auxB_len - 1 of them);
 * fastq_seq is the record's SEQ with every non-ACGT letter already mapped to 'N' (gmove.cpp:1128-1134). */
int orc_bam_read(orc_state_t *st,
                 const int16_t *raw, uint64_t len_raw_signal,
                 double digitisation, double offset, double range,
                 int32_t fastq_len, const char *fastq_seq, int32_t stride,
                 const int8_t *moves, uint32_t n_moves, uint64_t signal_len, uint64_t trim_offset);

/* results (slot = index into the slice, 0-based: slot i is kmers[index_start-1+i]) */
size_t      orc_n_slots(const orc_state_t *st);
const char *orc_slot_kmer(const orc_state_t *st, size_t slot);
uint64_t    orc_slot_count(const orc_state_t *st, size_t slot);              /* freq.txt value */
const char *orc_slot_text(const orc_state_t *st, size_t slot, size_t *len);  /* exact dump/<KMER> bytes */
size_t      orc_slot_n_values(const orc_state_t *st, size_t slot);
const double *orc_slot_values(const orc_state_t *st, size_t slot);           /* all printed samples, in order */
const uint32_t *orc_slot_event_lens(const orc_state_t *st, size_t slot);     /* one per kept event */
uint64_t    orc_total_samples(const orc_state_t *st);   /* sum of len_raw_signal over processed reads */
uint64_t    orc_reads_seen(const orc_state_t *st);

/* med/MAD of the last read processed with scaling==1 (for unit tests of the selection) */
void orc_last_medmad(const orc_state_t *st, double *med, double *mad);

/* standalone statistics (gmove.cpp:142-184): sorted(x)[n/2]; 1.4826*sorted(|x-med|)[n/2] */
double orc_median(const double *x, size_t n);
double orc_madf(const double *x, size_t n, double med);

/* writes output_dir/freq.txt and output_dir/dump/<KMER> (gmove.cpp:460-473, 523-534) */
int orc_write_outputs(const orc_state_t *st, const char *output_dir);

#ifdef __cplusplus
}
#endif
#endif
