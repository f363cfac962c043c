/* gmove_oracle_cli.c -- file front-end of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 *
 * `gmove_oracle [options] reads.slow5 event_alignment_file output_dir` restates gmove()'s option
 * handling and set-up (src/gmove.cpp:213-537) around gmove_oracle.c, with its own minimal readers
 * for ASCII SLOW5, FASTQ/FASTA, PAF and the 7-column move table, so that the product CLI (which has
 * independent parsers) can be diffed against it byte-for-byte on whole output directories.
 * SAM text input (gmove.cpp:1061-1266) is parsed here; binary BAM is not (exit status 3: convert with samtools view).
 */
#define _GNU_SOURCE
#include "gmove_oracle.h"

#include <dirent.h>
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

/* ---- ASCII SLOW5: '#'/'@' header lines, then TSV records ---------------- */
typedef struct { char *id; double dig, off, range; uint64_t len; int16_t *raw; } s5_rec_t;
typedef struct { s5_rec_t *r; size_t n, cap; } s5_t;

static int s5_load(const char *path, s5_t *out) {
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    char *line = NULL; size_t cap = 0; ssize_t got;
    while ((got = getline(&line, &cap, f)) != -1) {
        if (line[0] == '#' || line[0] == '@' || got < 2) continue;
        char *save = NULL;
        char *tok[8]; int nt = 0;
        for (char *p = strtok_r(line, "\t\n", &save); p && nt < 8; p = strtok_r(NULL, "\t\n", &save)) tok[nt++] = p;
        if (nt < 8) { fclose(f); return -2; }
        if (out->n == out->cap) { out->cap = out->cap ? out->cap * 2 : 16; out->r = (s5_rec_t *)realloc(out->r, out->cap * sizeof(s5_rec_t)); }
        s5_rec_t *r = &out->r[out->n++];
        r->id = strdup(tok[0]); r->dig = atof(tok[2]); r->off = atof(tok[3]); r->range = atof(tok[4]);
        r->len = strtoull(tok[6], NULL, 10);
        r->raw = (int16_t *)malloc((r->len ? r->len : 1) * sizeof(int16_t));
        char *q = tok[7];
        for (uint64_t i = 0; i < r->len; i++) { r->raw[i] = (int16_t)strtol(q, &q, 10); if (*q == ',') q++; }
    }
    free(line); fclose(f);
    return 0;
}
static s5_rec_t *s5_get(s5_t *s, const char *id) {
    for (size_t i = 0; i < s->n; i++) if (strcmp(s->r[i].id, id) == 0) return &s->r[i];
    return NULL;
}

/* ---- FASTQ / FASTA: name = first token after '@'/'>' (htslib faidx contract) -- */
typedef struct { char *name; char *seq; int64_t len; } fq_rec_t;
typedef struct { fq_rec_t *r; size_t n, cap; } fq_t;

static int fq_load(const char *path, fq_t *out) {
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    char *line = NULL; size_t cap = 0; ssize_t got;
    int state = 0; /* 0 expect header, 1 in sequence, 2 in quality */
    int fastq = 0; fq_rec_t *cur = NULL; int64_t qual_left = 0;
    while ((got = getline(&line, &cap, f)) != -1) {
        while (got > 0 && (line[got - 1] == '\n' || line[got - 1] == '\r')) line[--got] = 0;
        if (state == 2) { qual_left -= got; if (qual_left <= 0) state = 0; continue; }
        if (state == 0 || (state == 1 && !fastq && line[0] == '>')) {
            if (line[0] != '@' && line[0] != '>') continue;
            fastq = line[0] == '@';
            if (out->n == out->cap) { out->cap = out->cap ? out->cap * 2 : 16; out->r = (fq_rec_t *)realloc(out->r, out->cap * sizeof(fq_rec_t)); }
            cur = &out->r[out->n++];
            size_t e = 1; while (line[e] && line[e] != ' ' && line[e] != '\t') e++;
            cur->name = strndup(line + 1, e - 1); cur->seq = strdup(""); cur->len = 0;
            state = 1; continue;
        }
        if (state == 1) {
            if (fastq && line[0] == '+') { qual_left = cur->len; state = qual_left > 0 ? 2 : 0; continue; }
            cur->seq = (char *)realloc(cur->seq, (size_t)cur->len + (size_t)got + 1);
            memcpy(cur->seq + cur->len, line, (size_t)got + 1); cur->len += got;
        }
    }
    free(line); fclose(f);
    return 0;
}
static fq_rec_t *fq_get(fq_t *q, const char *name) {
    for (size_t i = 0; i < q->n; i++) if (strcmp(q->r[i].name, name) == 0) return &q->r[i];
    return NULL;
}

/* ---- option table: same order/meaning as long_options[] (gmove.cpp:49-72) -------------------- */
static struct option long_options[] = {
    {"kmer_size", required_argument, 0, 'k'}, {"sig_move_offset", required_argument, 0, 'm'},
    {"kmer_start_offset", required_argument, 0, 's'}, {"scaling", required_argument, 0, 0},
    {"margin", required_argument, 0, 0}, {"sample_limit", required_argument, 0, 0},
    {"file_limit", required_argument, 0, 0}, {"kmer_file", required_argument, 0, 0},
    {"index_start", required_argument, 0, 0}, {"index_end", required_argument, 0, 0},
    {"fastq", required_argument, 0, 0}, {"", no_argument, 0, 'd'},
    {"max_dur", required_argument, 0, 0}, {"min_dur", required_argument, 0, 0},
    {"pa_min", required_argument, 0, 0}, {"pa_max", required_argument, 0, 0},
    {"kmer_pick_margin", required_argument, 0, 0}, {"rna", no_argument, 0, 0},
    {"verbose", required_argument, 0, 'v'}, {"help", no_argument, 0, 'h'},
    {"version", no_argument, 0, 'V'}, {"debug-break", required_argument, 0, 0}, {0, 0, 0, 0}};

static int create_dir(const char *dir_name) { /* gmove.cpp:126-140 */
    struct stat stt;
    if (stat(dir_name, &stt) == -1) { if (mkdir(dir_name, 0700) == -1) return -2; }
    else {
        DIR *d = opendir(dir_name); size_t n = 0;
        if (d) { while (readdir(d)) n++; closedir(d); }
        if (n > 2) return -1;
    }
    return 0;
}

int main(int argc, char **argv) {
    orc_opt_t opt; orc_default_opt(&opt);
    uint32_t file_limit = 500; /* KMERS_TO_DUMP_LIMIT, poregen.h:34 */
    int signal_scale = 0, longindex = 0, c, help = 0;
    const char *kmer_file = NULL, *fastq = NULL;
    while ((c = getopt_long(argc, argv, "k:m:s:d", long_options, &longindex)) >= 0) { /* gmove.cpp:240-327 */
        if (c == 'k') { if (atoi(optarg) < 1) return 1; opt.kmer_size = (uint32_t)atoi(optarg); }
        else if (c == 'm') { if (atoi(optarg) < 0) return 1; opt.sig_move_offset = (uint32_t)atoi(optarg); }
        else if (c == 's') { if (atoi(optarg) < 1) return 1; opt.kmer_start_offset = (uint32_t)atoi(optarg); }
        else if (c == 'd') opt.delimit_files = 1;
        else if (c == 'v') {}
        else if (c == 'V') { printf("gmove 0.1.0\n"); return 0; }
        else if (c == 'h') help = 1;
        else if (c == 0 && longindex == 3) signal_scale = atoi(optarg);
        else if (c == 0 && longindex == 4) { if (atoi(optarg) < 0) return 1; opt.signal_print_margin = (uint32_t)atoi(optarg); }
        else if (c == 0 && longindex == 5) { if (atoi(optarg) < 0) return 1; opt.sample_limit = (uint32_t)atoi(optarg); }
        else if (c == 0 && longindex == 6) { if (atoi(optarg) < 0) return 1; file_limit = (uint32_t)atoi(optarg); opt.index_end = opt.index_start + file_limit - 1; }
        else if (c == 0 && longindex == 7) kmer_file = optarg;
        else if (c == 0 && longindex == 8) { if (atoi(optarg) < 1) return 1; opt.index_start = (uint32_t)atoi(optarg); file_limit = opt.index_end - opt.index_start + 1; }
        else if (c == 0 && longindex == 9) { if (atoi(optarg) < 1) return 1; opt.index_end = (uint32_t)atoi(optarg); file_limit = opt.index_end - opt.index_start + 1; }
        else if (c == 0 && longindex == 10) fastq = optarg;
        else if (c == 0 && longindex == 12) opt.max_dur = (uint32_t)atoi(optarg);
        else if (c == 0 && longindex == 13) opt.min_dur = (uint32_t)atoi(optarg);
        else if (c == 0 && longindex == 14) opt.pa_min = atof(optarg);
        else if (c == 0 && longindex == 15) opt.pa_max = atof(optarg);
        else if (c == 0 && longindex == 16) opt.kmer_pick_margin = atoi(optarg);
        else if (c == 0 && longindex == 17) opt.flag_rna = 1;
    }
    if (argc - optind != 3 || help) { fprintf(help ? stdout : stderr, "Usage: gmove_oracle reads.slow5 event_alignment_file output_dir\n"); return help ? 0 : 1; } /* gmove.cpp:330-336 */
    const char *slow5file = argv[optind], *move_table = argv[optind + 1], *output_dir = argv[optind + 2];

    int rcd = create_dir(output_dir);                                    /* gmove.cpp:374-392 */
    if (rcd < 0) { fprintf(stderr, "output dir %s: %s\n", output_dir, rcd == -1 ? "not empty" : "cannot create"); return 1; }
    char dump[4096]; snprintf(dump, sizeof dump, "%s/dump", output_dir);
    rcd = create_dir(dump);
    if (rcd < 0) return 1;

    char **kmers = NULL; size_t n_kmers = 0;                             /* gmove.cpp:394-426 */
    if (kmer_file) {
        FILE *f = fopen(kmer_file, "r");
        if (!f) return 1;
        char *line = NULL; size_t cap = 0; ssize_t got; size_t kcap = 0;
        while ((got = getline(&line, &cap, f)) != -1) {
            line[got - 1] = 0;
            if (got != (ssize_t)opt.kmer_size + 1) { fprintf(stderr, "kmer length mismatch\n"); return 1; } /* gmove.cpp:409-412 */
            if (n_kmers == kcap) { kcap = kcap ? kcap * 2 : 64; kmers = (char **)realloc(kmers, kcap * sizeof(char *)); }
            kmers[n_kmers++] = strdup(line);
        }
        free(line); fclose(f);
    } else kmers = orc_generate_kmers((int)opt.kmer_size, opt.flag_rna, &n_kmers);

    uint32_t num_kmers = (uint32_t)n_kmers;                              /* gmove.cpp:428-440 */
    if (file_limit < num_kmers) { /* slice as parsed */ }
    else if (file_limit > num_kmers - opt.index_start + 1) {
        if (opt.index_end > num_kmers) { file_limit = num_kmers - opt.index_start + 1; opt.index_end = opt.index_start + file_limit - 1; }
        else file_limit = opt.index_end - opt.index_start + 1;
    }
    if (signal_scale == 0) opt.scaling = 0; else if (signal_scale == 1) opt.scaling = 1; else return 1; /* gmove.cpp:479-491 */

    orc_state_t *st = orc_create(&opt, (const char *const *)kmers, n_kmers);
    if (!st) { fprintf(stderr, "invalid k-mer slice [%u,%u] of %zu\n", opt.index_start, opt.index_end, n_kmers); return 1; }
    if (orc_write_outputs(st, output_dir) != 0) return 1;                /* files exist from the start (gmove.cpp:460-473) */

    s5_t s5 = {0, 0, 0};
    if (s5_load(slow5file, &s5) != 0) { fprintf(stderr, "Error in opening file %s\n", slow5file); return 1; }

    size_t ml = strlen(move_table);                                      /* gmove.cpp:505-521 */
    const char *ext = ml >= 4 ? move_table + ml - 4 : "";
    int is_paf = strcmp(ext, ".paf") == 0, is_bam = strcmp(ext, ".bam") == 0, is_sam = strcmp(ext, ".sam") == 0;
    if (is_bam) { fprintf(stderr, "binary BAM input is not read by the oracle (use the SAM text of the same records)\n"); return 3; }
    fq_t fq = {0, 0, 0};
    if (is_paf) {
        if (!fastq) { fprintf(stderr, ".paf input requires an additional .fastq file\n"); return 1; } /* gmove.cpp:510-513 */
        if (fq_load(fastq, &fq) != 0) { fprintf(stderr, "Error in loading fastq index for %s\n", fastq); return 1; }
    }
    FILE *f = fopen(move_table, "r");
    if (!f) { fprintf(stderr, "Error in opening file %s\n", move_table); return 1; }
    char *line = NULL; size_t cap = 0; ssize_t got; int status = 0;
    while ((got = getline(&line, &cap, f)) != -1) {
        int rc;
        char *save = NULL;
        if (is_paf) {                                                    /* parse_paf_rec, gmove.cpp:977-1052 */
            char *col[12]; int nc = 0; char *ss = NULL;
            for (char *p = strtok_r(line, "\t\r\n", &save); p; p = strtok_r(NULL, "\t\r\n", &save)) {
                if (nc < 12) col[nc++] = p; else if (strncmp("ss:Z:", p, 5) == 0) ss = p + 5;
            }
            if (nc < 12) { status = 134; break; }                        /* assert(pch!=NULL) */
            if (!ss) { fprintf(stderr, "ss:Z: tag not found\n"); status = 1; break; }
            s5_rec_t *r = s5_get(&s5, col[0]);
            if (!r) { fprintf(stderr, "Error in when fetching the read\n"); status = 1; break; } /* gmove.cpp:746-749 */
            fq_rec_t *t = fq_get(&fq, col[5]);
            rc = orc_paf_read(st, r->raw, r->len, r->dig, r->off, r->range, atoi(col[2]), atoi(col[7]), atoi(col[8]),
                              t ? t->seq : NULL, t ? t->len : 0, ss);
        } else if (is_sam) {                                             /* sam_read1 + tags ns, ts, mv (gmove.cpp:1080-1134) */
            if (line[0] == '@') continue;
            char *col[11]; int nc = 0; char *mv = NULL; long long ns = -1, ts = -1;
            for (char *p = strtok_r(line, "\t\r\n", &save); p; p = strtok_r(NULL, "\t\r\n", &save)) {
                if (nc < 11) col[nc++] = p;
                else if (strncmp(p, "mv:B:c,", 7) == 0) mv = p + 7;
                else if (strncmp(p, "ns:i:", 5) == 0) ns = atoll(p + 5);
                else if (strncmp(p, "ts:i:", 5) == 0) ts = atoll(p + 5);
            }
            if (nc < 11) { status = 1; break; }
            if (ns < 0 || ts < 0 || !mv) { fprintf(stderr, "tag ns/ts/mv is not found\n"); status = 1; break; }  /* gmove.cpp:1086-1105 */
            size_t cap2 = strlen(mv) / 2 + 2, nm = 0;
            int8_t *vals = (int8_t *)malloc(cap2);
            for (char *q = mv; *q;) { vals[nm++] = (int8_t)strtol(q, &q, 10); if (*q == ',') q++; }
            if (nm == 0) { status = 1; free(vals); break; }
            int stride = vals[0];
            size_t sl = strlen(col[9]);
            char *seq = (char *)malloc(sl + 1);
            for (size_t i = 0; i < sl; i++) { char ch = col[9][i]; if (ch >= 'a' && ch <= 'z') ch -= 32; seq[i] = (ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T') ? ch : 'N'; }
            seq[sl] = 0;
            if (sl == 1 && col[9][0] == '*') { seq[0] = 0; sl = 0; }
            s5_rec_t *r = s5_get(&s5, col[0]);
            if (!r) { fprintf(stderr, "Error in when fetching the read\n"); status = 1; free(vals); free(seq); break; }
            rc = orc_bam_read(st, r->raw, r->len, r->dig, r->off, r->range, (int32_t)sl, seq, stride, vals + 1, (uint32_t)(nm - 1),
                              (uint64_t)ns, (uint64_t)ts);
            free(vals); free(seq);
        } else {                                                         /* gmove.cpp:570-577 */
            char *col[7]; int nc = 0;
            for (char *p = strtok_r(line, "\t", &save); p && nc < 7; p = strtok_r(NULL, "\t", &save)) col[nc++] = p;
            if (nc < 7) { status = 139; break; }
            s5_rec_t *r = s5_get(&s5, col[0]);
            if (!r) { fprintf(stderr, "Error in when fetching the read\n"); status = 1; break; }
            rc = orc_table_read(st, r->raw, r->len, r->dig, r->off, r->range, atoi(col[1]), col[2], atoi(col[3]), col[4],
                                strtoull(col[5], NULL, 10), atoi(col[6]));
        }
        if (rc == ORC_STOPPED) break;
        if (rc == ORC_ERR_ASSERT) { status = 134; break; }
        if (rc == ORC_ERR_UNDEFINED) { fprintf(stderr, "[oracle] input is undefined behaviour in the reference\n"); status = 70; break; }
        if (rc < 0) { status = 1; break; }
    }
    free(line); fclose(f);
    if (status == 0 && orc_write_outputs(st, output_dir) != 0) status = 1;
    fprintf(stderr, "[oracle] reads=%llu samples=%llu\n", (unsigned long long)orc_reads_seen(st), (unsigned long long)orc_total_samples(st));
    orc_destroy(st);
    return status;
}
