/* gmove_oracle.c -- CPU restatement of poregen's `gmove` collector (TEST INFRASTRUCTURE ONLY).
 *
 * See gmove_oracle.h for scope and pinning status. Every function cites the reference lines it
 * follows (paths relative to /root/reference/). The code is deliberately sequential and literal:
 * one read at a time, a zero-initialised double vector per read, exact order statistics, a
 * per-event filter chain and "%.8f" text appended per kept sample, exactly in reference order.
 */
#include "gmove_oracle.h"

#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

/* ------------------------------------------------------------------------------------------ */
/* small growable buffers                                                                     */

typedef struct { char *p; size_t n, cap; } cbuf_t;
typedef struct { double *p; size_t n, cap; } dbuf_t;
typedef struct { uint32_t *p; size_t n, cap; } ubuf_t;
typedef struct { int32_t *p; size_t n, cap; } ibuf_t;

static void *xrealloc(void *p, size_t sz) {
    void *q = realloc(p, sz ? sz : 1);
    if (!q) { fprintf(stderr, "[oracle] out of memory\n"); abort(); }
    return q;
}
static void cbuf_put(cbuf_t *b, const char *s, size_t n) {
    if (b->n + n + 1 > b->cap) { b->cap = (b->n + n + 1) * 2; b->p = (char *)xrealloc(b->p, b->cap); }
    memcpy(b->p + b->n, s, n); b->n += n; b->p[b->n] = 0;
}
static void dbuf_push(dbuf_t *b, double v) {
    if (b->n == b->cap) { b->cap = b->cap ? b->cap * 2 : 64; b->p = (double *)xrealloc(b->p, b->cap * sizeof(double)); }
    b->p[b->n++] = v;
}
static void ubuf_push(ubuf_t *b, uint32_t v) {
    if (b->n == b->cap) { b->cap = b->cap ? b->cap * 2 : 16; b->p = (uint32_t *)xrealloc(b->p, b->cap * sizeof(uint32_t)); }
    b->p[b->n++] = v;
}
static void ibuf_push(ibuf_t *b, int32_t v) {
    if (b->n == b->cap) { b->cap = b->cap ? b->cap * 2 : 16; b->p = (int32_t *)xrealloc(b->p, b->cap * sizeof(int32_t)); }
    b->p[b->n++] = v;
}

/* ------------------------------------------------------------------------------------------ */
/* state                                                                                      */

typedef struct {
    const char *kmer;  /* points into st->kmers */
    uint64_t count;    /* kmer_frequency_map value (gmove.cpp:474-477) */
    int open;          /* FILE* still non-NULL (gmove.cpp:946-949) */
    cbuf_t text;       /* bytes fprintf'd to dump/<kmer> */
    dbuf_t values;     /* the doubles that were printed */
    ubuf_t ev_len;     /* samples per kept event */
} orc_slot_t;

typedef struct { const char *kmer; size_t slot; } orc_key_t;

struct orc_state {
    orc_opt_t opt;
    char **kmers; size_t n_kmers;     /* full list (gmove.cpp:394-426) */
    orc_slot_t *slots; size_t n_slots; /* the slice [index_start-1, index_end) (gmove.cpp:460-477) */
    orc_key_t *keys;                   /* slice k-mers sorted by string: the std::map lookup */
    size_t num_kmers_complete;         /* gmove.cpp:731 */
    uint64_t total_samples, reads_seen;
    double last_med, last_mad;
};

void orc_default_opt(orc_opt_t *o) {
    /* init_opt (src/poregen.cpp:209-237), defaults src/poregen.h:30-43; scaling is 0 in effect
     * because gmove() overwrites it from a local initialised to 0 (src/gmove.cpp:229,479-484). */
    memset(o, 0, sizeof(*o));
    o->kmer_size = 9; o->sig_move_offset = 0; o->kmer_start_offset = 0; o->scaling = 0;
    o->signal_print_margin = 0; o->sample_limit = 100; o->index_start = 1; o->index_end = 500;
    o->delimit_files = 0; o->max_dur = 70; o->min_dur = 5; o->pa_min = 40.0; o->pa_max = 180.0;
    o->kmer_pick_margin = 2; o->flag_rna = 0;
}

static void gen_rec(const char *set, char *prefix, int depth, int k, char **out, size_t *n) {
    /* generate_kmers (src/poregen.cpp:248-267): depth-first, alphabet order => lexicographic */
    if (depth == k) { prefix[k] = 0; out[*n] = strdup(prefix); (*n)++; return; }
    for (int i = 0; i < 4; i++) { prefix[depth] = set[i]; gen_rec(set, prefix, depth + 1, k, out, n); }
}
char **orc_generate_kmers(int k, int rna, size_t *n_out) {
    size_t total = 1; for (int i = 0; i < k; i++) total *= 4;
    char **out = (char **)xrealloc(NULL, total * sizeof(char *));
    char *prefix = (char *)xrealloc(NULL, (size_t)k + 1);
    size_t n = 0;
    gen_rec(rna ? "ACGU" : "ACGT", prefix, 0, k, out, &n); /* gmove.cpp:231-232,421-425 */
    free(prefix);
    *n_out = n;
    return out;
}
void orc_free_kmers(char **kmers, size_t n) { for (size_t i = 0; i < n; i++) free(kmers[i]); free(kmers); }

static int key_cmp(const void *a, const void *b) {
    return strcmp(((const orc_key_t *)a)->kmer, ((const orc_key_t *)b)->kmer);
}

orc_state_t *orc_create(const orc_opt_t *opt, const char *const *kmers, size_t n_kmers) {
    /* gmove.cpp:460-477: one open file + one zero counter per k-mer of the slice */
    if (opt->index_start < 1 || opt->index_end > n_kmers || opt->index_end + 1 < opt->index_start) return NULL; /* SURVEY A.7 */
    orc_state_t *st = (orc_state_t *)calloc(1, sizeof(*st));
    st->opt = *opt;
    st->n_kmers = n_kmers;
    st->kmers = (char **)xrealloc(NULL, n_kmers * sizeof(char *));
    for (size_t i = 0; i < n_kmers; i++) st->kmers[i] = strdup(kmers[i]);
    st->n_slots = (size_t)opt->index_end - (opt->index_start - 1);
    st->slots = (orc_slot_t *)calloc(st->n_slots ? st->n_slots : 1, sizeof(orc_slot_t));
    st->keys = (orc_key_t *)calloc(st->n_slots ? st->n_slots : 1, sizeof(orc_key_t));
    for (size_t i = 0; i < st->n_slots; i++) {
        st->slots[i].kmer = st->kmers[opt->index_start - 1 + i];
        st->slots[i].open = 1;
        cbuf_put(&st->slots[i].text, "", 0);
        st->keys[i].kmer = st->slots[i].kmer; st->keys[i].slot = i;
    }
    qsort(st->keys, st->n_slots, sizeof(orc_key_t), key_cmp);
    for (size_t i = 1; i < st->n_slots; i++)
        if (strcmp(st->keys[i].kmer, st->keys[i - 1].kmer) == 0) { orc_destroy(st); return NULL; } /* duplicate k-mers: the
            reference would open the same path twice and leak the first FILE*; treated as invalid input */
    st->last_med = st->last_mad = NAN;
    return st;
}

void orc_destroy(orc_state_t *st) {
    if (!st) return;
    for (size_t i = 0; i < st->n_slots; i++) { free(st->slots[i].text.p); free(st->slots[i].values.p); free(st->slots[i].ev_len.p); }
    free(st->slots); free(st->keys);
    for (size_t i = 0; i < st->n_kmers; i++) free(st->kmers[i]);
    free(st->kmers); free(st);
}

/* kmer_frequency_map.find(kmer) (gmove.cpp:922, 647): exact string match within the slice */
static orc_slot_t *find_slot(orc_state_t *st, const char *kmer) {
    size_t lo = 0, hi = st->n_slots;
    while (lo < hi) {
        size_t mid = lo + (hi - lo) / 2;
        int c = strcmp(kmer, st->keys[mid].kmer);
        if (c == 0) return &st->slots[st->keys[mid].slot];
        if (c < 0) hi = mid; else lo = mid + 1;
    }
    return NULL;
}

/* ------------------------------------------------------------------------------------------ */
/* order statistics                                                                           */

/* ks_ksmall_double(n, arr, kk) (src/ksort.h:233-259): the kk-th smallest element, 0-based.
 * Any exact selection returns the same VALUE; this is a plain Hoare quick-select with
 * median-of-three pivoting on a scratch copy. */
static double kth_smallest(double *a, size_t n, size_t kk) {
    /* Wirth's selection: partition around a[k] until the k-th position is fixed */
    ptrdiff_t l = 0, m = (ptrdiff_t)n - 1, k = (ptrdiff_t)kk;
    while (l < m) {
        double x = a[k];
        ptrdiff_t i = l, j = m;
        do {
            while (a[i] < x) i++;
            while (x < a[j]) j--;
            if (i <= j) { double t = a[i]; a[i] = a[j]; a[j] = t; i++; j--; }
        } while (i <= j);
        if (j < k) l = i;
        if (k < i) m = j;
    }
    return a[k];
}

double orc_median(const double *x, size_t n) {
    /* calc_median (gmove.cpp:142-150): copy, ks_ksmall(n, copy, n/2) -- the UPPER median */
    double *copy = (double *)xrealloc(NULL, n * sizeof(double));
    memcpy(copy, x, n * sizeof(double));
    double m = kth_smallest(copy, n, n / 2);
    free(copy);
    return m;
}

double orc_madf(const double *x, size_t n, double med) {
    /* calc_madf (gmove.cpp:161-184): n==1 -> 0.0; else 1.4826 * median(|x - med|) */
    if (n == 1) return 0.0;
    double *absdiff = (double *)xrealloc(NULL, n * sizeof(double));
    for (size_t i = 0; i < n; i++) absdiff[i] = fabs(x[i] - med);
    double mad = orc_median(absdiff, n);
    free(absdiff);
    return mad * 1.4826;
}

/* gmove.cpp:754-776 (PAF) / 592-614 (table): pA convert with zero-fill, then optional med-MAD.
 * x has len_total zero-initialised entries; only the first n are filled/normalised. */
static void convert_and_scale(orc_state_t *st, double *x, const int16_t *raw, uint64_t n,
                              double digitisation, double offset, double range) {
    const orc_opt_t *o = &st->opt;
    for (uint64_t i = 0; i < n; i++) {
        double pA = (raw[i] + offset) * (range / digitisation); /* TO_PICOAMPS, src/poregen.h:30 */
        if (pA < o->pa_min || pA > o->pa_max) continue;          /* slot stays 0.0 */
        x[i] = pA;
    }
    if (o->scaling == 1) {
        double med = orc_median(x, n);
        double mad = orc_madf(x, n, med);
        mad = (mad > 1.0) ? mad : 1.0; /* gmove.cpp:771 */
        for (uint64_t i = 0; i < n; i++) x[i] = (x[i] - med) / mad;
        st->last_med = med; st->last_mad = mad;
    }
}

/* gmove.cpp:928-950 (PAF) / 653-685 (table): margin expand/clamp, print window, count, close */
static int emit_event(orc_state_t *st, orc_slot_t *s, const double *x, uint64_t len_raw_signal,
                      uint32_t raw_start_local, uint32_t raw_end_local) {
    const orc_opt_t *o = &st->opt;
    /* `raw_start_local - margin < 0` is an unsigned compare: never true (gmove.cpp:928-932) */
    if (o->signal_print_margin > raw_start_local) return ORC_ERR_UNDEFINED; /* wraps in the reference */
    raw_start_local -= o->signal_print_margin;
    if ((uint64_t)raw_end_local + o->signal_print_margin > len_raw_signal) raw_end_local = (uint32_t)len_raw_signal;
    else raw_end_local += o->signal_print_margin;
    if (raw_end_local <= raw_start_local || raw_end_local > len_raw_signal) return ORC_ERR_UNDEFINED; /* empty/invalid vector range */
    char tmp[512];
    size_t n = raw_end_local - raw_start_local, k;
    for (k = 0; k < n - 1; k++) {
        int w = snprintf(tmp, sizeof tmp, "%.8f,", x[raw_start_local + k]);
        cbuf_put(&s->text, tmp, (size_t)w); dbuf_push(&s->values, x[raw_start_local + k]);
    }
    int w = snprintf(tmp, sizeof tmp, "%.8f;", x[raw_start_local + k]);
    cbuf_put(&s->text, tmp, (size_t)w); dbuf_push(&s->values, x[raw_start_local + k]);
    ubuf_push(&s->ev_len, (uint32_t)n);
    s->count += 1;
    if (s->count == o->sample_limit) { s->open = 0; st->num_kmers_complete++; }
    return ORC_OK;
}

/* delimit_kmer_files (gmove.cpp:196-203): ':' to every still-open file of the slice */
static void delimit(orc_state_t *st) {
    for (size_t i = 0; i < st->n_slots; i++) if (st->slots[i].open) cbuf_put(&st->slots[i].text, ":", 1);
}

/* pick_this_kmer (gmove.cpp:204-211) */
static int pick_this_kmer(const ibuf_t *P, int left_pos, int kmer_length, int margin, int *undefined) {
    for (size_t i = 0; i < P->n; i++) {
        if (left_pos + kmer_length + margin <= P->p[i]) {
            if (i == 0) { *undefined = 1; return 0; } /* would read indel_pos[-1] */
            if (P->p[i - 1] <= left_pos - margin) return 1;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* PAF path: one loop iteration of process_move_table_paf (gmove.cpp:732-969)                  */

#define ORC_MAX_LEN_KMER 2000 /* gmove.cpp:32 */

int orc_paf_read(orc_state_t *st, const int16_t *raw, uint64_t len_raw_signal,
                 double digitisation, double offset, double range,
                 int32_t query_start, int32_t target_start, int32_t target_end,
                 const char *target_seq, int64_t target_len, const char *ss) {
    const orc_opt_t *o = &st->opt;
    if (st->num_kmers_complete == st->n_kmers) return ORC_STOPPED;              /* gmove.cpp:733-735 */
    if (!((uint64_t)(int64_t)query_start < len_raw_signal)) return ORC_ERR_ASSERT; /* gmove.cpp:752 */
    st->reads_seen++; st->total_samples += len_raw_signal;

    double *x = (double *)calloc(len_raw_signal ? len_raw_signal : 1, sizeof(double)); /* gmove.cpp:751 */
    convert_and_scale(st, x, raw, len_raw_signal, digitisation, offset, range);       /* gmove.cpp:754-776 */

    int rc = ORC_OK;
    size_t start_raw = (size_t)query_start;
    size_t start_kmer = (size_t)target_start, end_kmer = (size_t)target_end;            /* gmove.cpp:779-780 */
    size_t cap = ORC_MAX_LEN_KMER;
    int *st_raw_idx = (int *)xrealloc(NULL, sizeof(int) * cap);
    int *end_raw_idx = (int *)xrealloc(NULL, sizeof(int) * cap);
    for (size_t i = 0; i < cap; i++) st_raw_idx[i] = end_raw_idx[i] = -1;
    ibuf_t P = {0, 0, 0};
    cbuf_t refined = {0, 0, 0};
    char *fastq_seq = NULL;

    size_t st_k = start_kmer, end_k = end_kmer;
    int rna = start_kmer > end_kmer ? 1 : 0;                                            /* gmove.cpp:793 */
    if (rna) { st_k = end_kmer; end_k = start_kmer; }
    if (rna && o->flag_rna == 0) { rc = ORC_ERR_RNA_FLAG; goto done; }                  /* gmove.cpp:795-798 */

    /* faidx_fetch_seq(m_fai, tid, st_k, end_k-1, &fastq_len) (gmove.cpp:805): htslib 1.17
     * faidx_adjust_position: name absent -> len=-2; beg/end clamped into the sequence; the result is
     * [beg, end] inclusive. */
    int fastq_len;
    {
        int beg = (int)st_k, end = (int)(end_k - 1);
        if (!target_seq) fastq_len = -2;
        else {
            int64_t b = beg, e = end;
            if (e < b) b = e;
            if (b < 0) b = 0; else if (target_len <= b) b = target_len;
            if (e < 0) e = 0; else if (target_len <= e) e = target_len - 1;
            int64_t n = e + 1 - b; if (n < 0) n = 0;
            fastq_seq = (char *)xrealloc(NULL, (size_t)n + 1);
            memcpy(fastq_seq, target_seq + b, (size_t)n); fastq_seq[n] = 0;
            fastq_len = (int)n;
        }
    }
    if (fastq_len < (int)o->kmer_size) { rc = ORC_SKIPPED; goto done; }                 /* gmove.cpp:806-808 */
    size_t fetched_size = (size_t)fastq_len;
    if (rna) for (int i = 0; i < fastq_len; i++) if (fastq_seq[i] == 'T') fastq_seq[i] = 'U'; /* gmove.cpp:815-817 */

    size_t i_k = 0, i_raw = start_raw, i_k_raw = 0, num_deletion = 0;                   /* gmove.cpp:799-800,822 */
    if (rna == 0) ibuf_push(&P, (int32_t)(0 - (int64_t)st_k));                          /* gmove.cpp:824-826 */
    cbuf_put(&refined, "", 0);

    char buff[11]; int i_buff = 0;                                                      /* gmove.cpp:831 */
    for (const char *c = ss; *c; c++) {
        if (*c == ',' || *c == 'I' || *c == 'D') {
            if (i_buff <= 0) { rc = ORC_ERR_BAD_SS; goto done; }                         /* gmove.cpp:834 */
            buff[i_buff] = 0;
            int num = atoi(buff);
            if (num < 0) { rc = ORC_ERR_BAD_SS; goto done; }
            i_buff = 0; buff[0] = 0;
            if (*c == 'I') { i_raw += (size_t)num; ibuf_push(&P, (int32_t)(i_k - num_deletion)); }
            else if (*c == 'D') { ibuf_push(&P, (int32_t)(i_k - num_deletion)); i_k += (size_t)num; num_deletion += (size_t)num; }
            else {
                size_t src = rna ? (size_t)fastq_len - i_k - 1 : i_k;                   /* gmove.cpp:849-853 */
                if (src >= fetched_size) { rc = ORC_ERR_UNDEFINED; goto done; }          /* ss overruns the fetched sequence */
                cbuf_put(&refined, &fastq_seq[src], 1);
                end_raw_idx[i_k_raw] = (int)i_raw; i_raw += (size_t)num;                /* window START */
                st_raw_idx[i_k_raw] = (int)i_raw; i_k++;                                /* window END   */
                i_k_raw++;
            }
            if (i_k >= cap) {                                                            /* gmove.cpp:858-865 */
                st_raw_idx = (int *)xrealloc(st_raw_idx, sizeof(int) * cap * 2);
                end_raw_idx = (int *)xrealloc(end_raw_idx, sizeof(int) * cap * 2);
                for (size_t i = cap; i < cap * 2; i++) st_raw_idx[i] = end_raw_idx[i] = -1;
                cap *= 2;
            }
        } else {
            if (!isdigit((unsigned char)*c)) { rc = ORC_ERR_BAD_SS; goto done; }         /* gmove.cpp:867 */
            if (i_buff >= 10) { rc = ORC_ERR_UNDEFINED; goto done; }                     /* buff[11] overflow */
            buff[i_buff++] = *c;
        }
    }
    fastq_len = (int)refined.n;                                                          /* gmove.cpp:872 */
    if (rna) {                                                                           /* gmove.cpp:877-884 */
        for (size_t i = 0; i < P.n; i++) P.p[i] = fastq_len - P.p[i];
        ibuf_push(&P, (int32_t)(0 - (int64_t)st_k));
        for (size_t a = 0, b = P.n; a + 1 < b; a++, b--) { int32_t t = P.p[a]; P.p[a] = P.p[b - 1]; P.p[b - 1] = t; }
        for (size_t a = 0, b = refined.n; a + 1 < b; a++, b--) { char t = refined.p[a]; refined.p[a] = refined.p[b - 1]; refined.p[b - 1] = t; }
    }
    ibuf_push(&P, (int32_t)((int64_t)end_k + o->kmer_pick_margin));                     /* gmove.cpp:885 */

    if ((uint32_t)fastq_len < o->kmer_size) { rc = ORC_ERR_UNDEFINED; goto done; }       /* unsigned wrap at gmove.cpp:891 */
    {
        char kmer[64];
        if (o->kmer_size >= sizeof kmer) { rc = ORC_ERR_UNDEFINED; goto done; }
        for (size_t i = 0; i <= (size_t)((uint32_t)fastq_len - o->kmer_size); i++) {   /* gmove.cpp:891 */
            size_t e = i + o->sig_move_offset;
            if (e >= cap) { rc = ORC_ERR_UNDEFINED; goto done; }
            if (end_raw_idx[e] == -1) {
                if (st_raw_idx[e] != -1) { rc = ORC_ERR_INTERNAL; goto done; }          /* gmove.cpp:893 */
                continue;
            }
            int left = rna ? (int)((size_t)fastq_len - i - o->kmer_size) : (int)i;      /* gmove.cpp:899-908 */
            memcpy(kmer, refined.p + left, o->kmer_size); kmer[o->kmer_size] = 0;
            int undefined = 0;
            int pick = pick_this_kmer(&P, left, (int)o->kmer_size, o->kmer_pick_margin, &undefined);
            if (undefined) { rc = ORC_ERR_UNDEFINED; goto done; }
            if (!pick) continue;
            uint32_t raw_start_local = (uint32_t)end_raw_idx[e];                        /* gmove.cpp:913-914 */
            uint32_t raw_end_local = (uint32_t)st_raw_idx[e];
            if (raw_end_local - raw_start_local > o->max_dur) continue;                 /* gmove.cpp:916-921 */
            if (raw_end_local - raw_start_local < o->min_dur) continue;
            orc_slot_t *s = find_slot(st, kmer);
            if (!s) continue;                                                           /* gmove.cpp:922-924 */
            if (s->count == o->sample_limit) continue;                                  /* gmove.cpp:925-927 */
            rc = emit_event(st, s, x, len_raw_signal, raw_start_local, raw_end_local);
            if (rc != ORC_OK) goto done;
            if (i + o->kmer_size > fetched_size) break;                                 /* gmove.cpp:951-953 */
        }
    }
    if (o->delimit_files == 1) delimit(st);                                              /* gmove.cpp:960-962 */
done:
    free(x); free(st_raw_idx); free(end_raw_idx); free(P.p); free(refined.p); free(fastq_seq);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* table path: one loop iteration of process_move_table_file (gmove.cpp:557-700)               */

int orc_table_read(orc_state_t *st, const int16_t *raw, uint64_t len_total,
                   double digitisation, double offset, double range,
                   int32_t fastq_len, const char *fastq_seq, int32_t stride,
                   const char *move_seq, uint64_t signal_len, int32_t trim_offset) {
    const orc_opt_t *o = &st->opt;
    if (st->num_kmers_complete == st->n_kmers) return ORC_STOPPED;                       /* gmove.cpp:558-560 */
    if (len_total != signal_len) return ORC_ERR_ASSERT;                                  /* gmove.cpp:589 */
    if (!((uint64_t)(int64_t)trim_offset < len_total)) return ORC_ERR_ASSERT;            /* gmove.cpp:590 */
    st->reads_seen++; st->total_samples += len_total;
    uint64_t len_raw_signal = len_total - (uint64_t)trim_offset;                         /* gmove.cpp:591 */
    double *x = (double *)calloc(len_total ? len_total : 1, sizeof(double));
    convert_and_scale(st, x, raw + trim_offset, len_raw_signal, digitisation, offset, range); /* gmove.cpp:592-614 */

    int rc = ORC_OK;
    size_t move_len = strlen(move_seq), seq_size = strlen(fastq_seq);
    if (fastq_len < 10) { rc = ORC_SKIPPED; goto done; }                                 /* gmove.cpp:616-619 */
    {
        uint32_t move_count = 0; size_t move_idx = 0, start_move_idx = 0;
        while (move_count < o->sig_move_offset + 1) {                                    /* gmove.cpp:623-629 */
            if (move_idx >= move_len) { rc = ORC_ERR_UNDEFINED; goto done; }             /* runs off the string */
            if (move_seq[move_idx] == '1') { move_count++; start_move_idx = move_idx; }
            move_idx++;
        }
        move_idx = start_move_idx + 1;
        size_t seq_start = o->kmer_start_offset;
        char kmer[64];
        if (o->kmer_size >= sizeof kmer) { rc = ORC_ERR_UNDEFINED; goto done; }
        for (; move_idx <= move_len; move_idx++) {                                       /* gmove.cpp:632 (move_seq[move_len] is NUL) */
            if (move_seq[move_idx] != '1') continue;
            if (seq_start > seq_size) { rc = ORC_ERR_UNDEFINED; goto done; }             /* substr would throw */
            size_t kl = seq_size - seq_start < o->kmer_size ? seq_size - seq_start : o->kmer_size;
            memcpy(kmer, fastq_seq + seq_start, kl); kmer[kl] = 0;                       /* gmove.cpp:634 */
            uint32_t raw_start_local = (uint32_t)(start_move_idx * (size_t)stride);      /* gmove.cpp:636-639 */
            uint32_t raw_end_local = (uint32_t)(move_idx * (size_t)stride);
            start_move_idx = move_idx;
            seq_start++;
            if (raw_end_local - raw_start_local > o->max_dur) continue;
            if (raw_end_local - raw_start_local < o->min_dur) continue;
            orc_slot_t *s = find_slot(st, kmer);
            if (!s) continue;
            if (s->count == o->sample_limit) continue;
            rc = emit_event(st, s, x, len_raw_signal, raw_start_local, raw_end_local);
            if (rc != ORC_OK) goto done;
            if (seq_start + o->kmer_size > seq_size) break;                              /* gmove.cpp:687-689 */
        }
    }
    if (o->delimit_files == 1) delimit(st);                                              /* gmove.cpp:692-694 */
done:
    free(x);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* SAM/BAM path: one loop iteration of process_move_table_bam (gmove.cpp:1080-1261)            */

int orc_bam_read(orc_state_t *st, const int16_t *raw, uint64_t len_total,
                 double digitisation, double offset, double range,
                 int32_t fastq_len, const char *fastq_seq, int32_t stride,
                 const int8_t *moves, uint32_t n_moves, uint64_t signal_len, uint64_t trim_offset) {
    const orc_opt_t *o = &st->opt;
    if (st->num_kmers_complete == st->n_kmers) return ORC_STOPPED;                       /* gmove.cpp:1081-1083 */
    if (len_total != signal_len) return ORC_ERR_ASSERT;                                  /* gmove.cpp:1146 */
    if (!(trim_offset < len_total)) return ORC_ERR_ASSERT;                               /* gmove.cpp:1147 */
    st->reads_seen++; st->total_samples += len_total;
    uint64_t len_raw_signal = len_total - trim_offset;                                   /* gmove.cpp:1148 */
    double *x = (double *)calloc(len_total ? len_total : 1, sizeof(double));
    int rc = ORC_OK;
    /* gmove.cpp:1149-1160: ANY out-of-range sample skips the whole read (no zero-fill on this path) */
    for (uint64_t i = 0; i < len_raw_signal; i++) {
        double pA = (raw[i + trim_offset] + offset) * (range / digitisation);
        if (pA < o->pa_min || pA > o->pa_max) { rc = ORC_SKIPPED; goto done; }
        x[i] = pA;
    }
    if (o->scaling == 1) {                                                               /* gmove.cpp:1162-1176 */
        double med = orc_median(x, len_raw_signal);
        double mad = orc_madf(x, len_raw_signal, med);
        mad = (mad > 1.0) ? mad : 1.0;
        for (uint64_t i = 0; i < len_raw_signal; i++) x[i] = (x[i] - med) / mad;
        st->last_med = med; st->last_mad = mad;
    }
    if (fastq_len < 10) { rc = ORC_SKIPPED; goto done; }                                 /* gmove.cpp:1178-1181 */
    {
        /* bam_auxB2i(mv_array, q + 1) is moves[q] for q < n_moves and 0 beyond the array (htslib 1.17 sam.c) */
        const size_t move_len = (size_t)n_moves + 1;                                     /* bam_auxB_len includes the stride */
        size_t seq_size = strlen(fastq_seq);
        uint32_t move_count = 0; size_t move_idx = 0, start_move_idx = 0;
        while (move_count < o->sig_move_offset + 1) {                                    /* gmove.cpp:1185-1192 */
            if (move_idx >= n_moves) { rc = ORC_ERR_UNDEFINED; goto done; }              /* would spin forever on zeros */
            if (moves[move_idx] == 1) { move_count++; start_move_idx = move_idx; }
            move_idx++;
        }
        move_idx = start_move_idx + 1;
        size_t seq_start = o->kmer_start_offset;
        char kmer[64];
        if (o->kmer_size >= sizeof kmer) { rc = ORC_ERR_UNDEFINED; goto done; }
        for (; move_idx <= move_len; move_idx++) {                                       /* gmove.cpp:1195 */
            int8_t value = move_idx < n_moves ? moves[move_idx] : 0;
            if (value != 1) continue;
            if (seq_start > seq_size) { rc = ORC_ERR_UNDEFINED; goto done; }             /* substr would throw */
            size_t kl = seq_size - seq_start < o->kmer_size ? seq_size - seq_start : o->kmer_size;
            memcpy(kmer, fastq_seq + seq_start, kl); kmer[kl] = 0;                       /* gmove.cpp:1198 */
            uint32_t raw_start_local = (uint32_t)(start_move_idx * (size_t)stride);      /* gmove.cpp:1199-1202 */
            uint32_t raw_end_local = (uint32_t)(move_idx * (size_t)stride);
            start_move_idx = move_idx;
            seq_start++;
            if (raw_end_local - raw_start_local > o->max_dur) continue;
            if (raw_end_local - raw_start_local < o->min_dur) continue;
            orc_slot_t *s = find_slot(st, kmer);
            if (!s) continue;
            if (s->count == o->sample_limit) continue;
            rc = emit_event(st, s, x, len_raw_signal, raw_start_local, raw_end_local);
            if (rc != ORC_OK) goto done;
            if (seq_start + o->kmer_size > seq_size) break;                              /* gmove.cpp:1248-1250 */
        }
    }
    if (o->delimit_files == 1) delimit(st);                                              /* gmove.cpp:1253-1255 */
done:
    free(x);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* accessors and output                                                                       */

size_t orc_n_slots(const orc_state_t *st) { return st->n_slots; }
const char *orc_slot_kmer(const orc_state_t *st, size_t s) { return st->slots[s].kmer; }
uint64_t orc_slot_count(const orc_state_t *st, size_t s) { return st->slots[s].count; }
const char *orc_slot_text(const orc_state_t *st, size_t s, size_t *len) { if (len) *len = st->slots[s].text.n; return st->slots[s].text.p; }
size_t orc_slot_n_values(const orc_state_t *st, size_t s) { return st->slots[s].values.n; }
const double *orc_slot_values(const orc_state_t *st, size_t s) { return st->slots[s].values.p; }
const uint32_t *orc_slot_event_lens(const orc_state_t *st, size_t s) { return st->slots[s].ev_len.p; }
uint64_t orc_total_samples(const orc_state_t *st) { return st->total_samples; }
uint64_t orc_reads_seen(const orc_state_t *st) { return st->reads_seen; }
void orc_last_medmad(const orc_state_t *st, double *med, double *mad) { *med = st->last_med; *mad = st->last_mad; }

int orc_write_outputs(const orc_state_t *st, const char *output_dir) {
    /* gmove.cpp:383-392 (dump dir), 460-473 (one file per slice k-mer), 525-534 (freq.txt) */
    char path[4096];
    mkdir(output_dir, 0700);
    snprintf(path, sizeof path, "%s/dump", output_dir);
    mkdir(path, 0700);
    for (size_t i = 0; i < st->n_slots; i++) {
        snprintf(path, sizeof path, "%s/dump/%s", output_dir, st->slots[i].kmer);
        FILE *f = fopen(path, "w");
        if (!f) return -1;
        fwrite(st->slots[i].text.p, 1, st->slots[i].text.n, f);
        fclose(f);
    }
    snprintf(path, sizeof path, "%s/freq.txt", output_dir);
    FILE *f = fopen(path, "w");
    if (!f) return -1;
    for (size_t i = 0; i < st->n_slots; i++) fprintf(f, "%s\t%llu\n", st->slots[i].kmer, (unsigned long long)st->slots[i].count);
    fclose(f);
    return 0;
}
