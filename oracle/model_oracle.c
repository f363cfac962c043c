/* model_oracle.c -- CPU restatement of the step BEHIND gmove in the reference's pipeline. TEST INFRASTRUCTURE ONLY
 * (tests/, __graft_entry__.smoke and bench.py's cpu_baseline may run it; the product never does).
 *
 * Restates, for one dump directory:
 *   scripts/poregen.sh:54-85  calculate_mean_stddev_all <dir> <out> <limit>
 *        mean=$(tr ';,' '\n' < "$file" | tail -n +2 | datamash median 1)
 *        stddev=$(tr ';,' '\n' < "$file" | tail -n +2 | datamash sstdev 1)
 *        if (( $(echo "$stddev > $limit" | bc -l) )); then stddev=$limit; fi
 *        echo -e "$(basename "$file")\t$mean\t$stddev" >> "$output_file"
 *   scripts/poregen.sh:33-52  calculate_dwell_times_medians <dir> <out>
 *        median=$(awk -F';' '{for(i=1;i<=NF;i++) print gsub(",", "", $i)}' "$file" | datamash median 1)
 *        echo -e "$filename\t$median" >> "$output_file"
 * in the order of the shell glob "$dir"/STAR (sorted names; STAR = the asterisk).
 *
 * PARITY UNPINNED: the arithmetic lives in GNU datamash, which is neither in /root/reference nor in this image, and
 * the reference pins no version and holds no expected output for this step. What is restated is datamash's published
 * algorithm (1.x, src/utils.c): values are parsed with strtold into long double; median_value() sorts and returns
 * values[n/2] or (values[n/2-1] + values[n/2]) / 2.0; variance_value() takes mean = sum/n, then
 * sum((x-mean)*(x-mean)) / (n - 1), both sums left to right in long double; sstdev = sqrtl of that; numbers are printed
 * with "%.14Lg". A single value gives 0/0: printed here as "nan" (glibc prints the x87 default NaN as "-nan"; datamash's
 * own test-suite strips the sign). Empty input prints nothing. `bc` compares the two decimal texts; "nan" and the empty
 * string do not exceed the limit.
 *
 * usage: model_oracle stats <dump_dir> <limit>      -> raw model lines on stdout
 *        model_oracle dwell <dump_dir>              -> dwell lines on stdout
 */
#define _GNU_SOURCE
#include <dirent.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int cmp_name(const void *a, const void *b) { return strcmp(*(char *const *)a, *(char *const *)b); }
static int cmp_ld(const void *a, const void *b) {
    const long double x = *(const long double *)a, y = *(const long double *)b;
    return (x > y) - (x < y);
}

static char *slurp(const char *path, size_t *len) {
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *b = malloc((size_t)n + 1);
    if (n && fread(b, 1, (size_t)n, f) != (size_t)n) { perror(path); exit(2); }
    b[n] = 0;
    fclose(f);
    *len = (size_t)n;
    return b;
}

/* datamash median 1 / sstdev 1 on `n` parsed values; ok = 0 when datamash would have stopped on a bad field */
static void datamash_text(long double *v, size_t n, int ok, int want_sd, char *out, size_t cap) {
    out[0] = 0;
    if (!ok || n == 0) return;
    if (!want_sd) {
        qsort(v, n, sizeof *v, cmp_ld);
        const long double m = (n & 1) ? v[n / 2] : (v[n / 2 - 1] + v[n / 2]) / 2.0L;
        snprintf(out, cap, "%.14Lg", m);
        return;
    }
    long double sum = 0;
    for (size_t i = 0; i < n; i++) sum += v[i];
    const long double mean = sum / n;
    sum = 0;
    for (size_t i = 0; i < n; i++) sum += (v[i] - mean) * (v[i] - mean);
    if (n < 2) { snprintf(out, cap, "nan"); return; }
    snprintf(out, cap, "%.14Lg", sqrtl(sum / (n - 1)));
}

static void stats_file(const char *dir, const char *name, const char *limit) {
    char path[4096];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    size_t len;
    char *b = slurp(path, &len);
    /* tr ';,' '\n' : lines; a last line without '\n' still counts */
    size_t cap = 16, n = 0, line_no = 0;
    long double *v = malloc(cap * sizeof *v);
    int ok = 1;
    size_t i = 0;
    while (i < len) {
        size_t j = i;
        while (j < len && b[j] != ';' && b[j] != ',' && b[j] != '\n') j++;
        line_no++;
        if (line_no >= 2) { /* tail -n +2 */
            char save = b[j];
            b[j] = 0;
            char *end;
            const long double x = strtold(b + i, &end);
            if (end == b + i || *end != 0) ok = 0; /* datamash: invalid numeric value / missing field */
            b[j] = save;
            if (n == cap) { cap *= 2; v = realloc(v, cap * sizeof *v); }
            v[n++] = x;
        }
        i = j + 1;
    }
    char med[64], sd[64];
    long double *w = malloc((n ? n : 1) * sizeof *w);
    memcpy(w, v, n * sizeof *v);
    datamash_text(w, n, ok, 0, med, sizeof med);
    datamash_text(v, n, ok, 1, sd, sizeof sd);
    const char *sd_out = sd;
    if (sd[0] && strcmp(sd, "nan") != 0 && strtold(sd, NULL) > strtold(limit, NULL)) sd_out = limit;
    printf("%s\t%s\t%s\n", name, med, sd_out);
    free(w); free(v); free(b);
}

static void dwell_file(const char *dir, const char *name) {
    char path[4096];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    size_t len;
    char *b = slurp(path, &len);
    size_t cap = 16, n = 0;
    long double *v = malloc(cap * sizeof *v);
    /* awk reads records separated by '\n'; every record is split on ';' (an empty record has NF = 0) */
    size_t i = 0;
    while (i < len) {
        size_t e = i;
        while (e < len && b[e] != '\n') e++;
        if (e > i) {
            size_t commas = 0;
            for (size_t j = i; j <= e; j++) {
                if (j == e || b[j] == ';') {
                    if (n == cap) { cap *= 2; v = realloc(v, cap * sizeof *v); }
                    v[n++] = (long double)commas;
                    commas = 0;
                } else if (b[j] == ',') commas++;
            }
        }
        i = e + 1;
    }
    char med[64];
    datamash_text(v, n, 1, 0, med, sizeof med);
    printf("%s\t%s\n", name, med);
    free(v); free(b);
}

int main(int argc, char **argv) {
    const int stats = argc == 4 && strcmp(argv[1], "stats") == 0, dwell = argc == 3 && strcmp(argv[1], "dwell") == 0;
    if (!stats && !dwell) { fprintf(stderr, "usage: model_oracle stats <dump_dir> <limit> | model_oracle dwell <dump_dir>\n"); return 2; }
    DIR *d = opendir(argv[2]);
    if (!d) { perror(argv[2]); return 2; }
    size_t cap = 1024, n = 0;
    char **names = malloc(cap * sizeof *names);
    struct dirent *de;
    while ((de = readdir(d))) {
        if (de->d_name[0] == '.') continue; /* the glob * skips dot files */
        if (n == cap) { cap *= 2; names = realloc(names, cap * sizeof *names); }
        names[n++] = strdup(de->d_name);
    }
    closedir(d);
    qsort(names, n, sizeof *names, cmp_name);
    for (size_t i = 0; i < n; i++) {
        if (stats) stats_file(argv[2], names[i], argv[3]);
        else dwell_file(argv[2], names[i]);
        free(names[i]);
    }
    free(names);
    return 0;
}
