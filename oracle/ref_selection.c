/* oracle/_ref/libref_selection.so -- the REFERENCE's own selection code, compiled where it lies.
 *
 * TEST INFRASTRUCTURE ONLY (tests/, tools/gen_ksmall_fixtures.py): nothing in the product links, loads or calls this.
 *
 * Of the reference's gmove path exactly one piece builds in this image without the libraries it lacks (slow5lib, htslib):
 * src/ksort.h, a self-contained header. This file instantiates it the way src/gmove.cpp:24-27 does --
 *     #include "ksort.h"  +  KSORT_INIT_GENERIC(double)
 * -- so that ks_ksmall_double below IS the reference's quickselect (src/ksort.h:233-259), not a restatement. It is built by
 * oracle/Makefile with -I/root/reference/src, only in the build container (the reference does not travel); the output goes to
 * oracle/_ref/ (git-ignored, not gpurun-ignored: the .so travels to the GPU box like any other built file).
 *
 * Around it, two things gmove.cpp keeps `static` and which therefore cannot be linked: calc_median / calc_madf
 * (src/gmove.cpp:142-184) and the pA conversion with zero fill (src/gmove.cpp:750-776, TO_PICOAMPS src/poregen.h:30). They are
 * restated here, a dozen lines, each with its reference line -- the selection they call is the reference's compiled code.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "ksort.h"            /* /root/reference/src/ksort.h, via -I */
KSORT_INIT_GENERIC(double)    /* src/gmove.cpp:27 */

/* src/gmove.cpp:142-150 */
static double ref_calc_median(const double *x, size_t n) {
    double *copy = (double *)malloc(n * sizeof(double));
    memcpy(copy, x, n * sizeof(double));
    double m = ks_ksmall_double(n, copy, n / 2);
    free(copy);
    return m;
}

/* src/gmove.cpp:161-184 (x != NULL, med given) */
static double ref_calc_madf(const double *x, size_t n, double med) {
    const double mad_scaling_factor = 1.4826;
    if (1 == n) return 0.0;
    double *absdiff = (double *)malloc(n * sizeof(double));
    for (size_t i = 0; i < n; i++) absdiff[i] = fabs(x[i] - med);
    const double mad = ref_calc_median(absdiff, n);
    free(absdiff);
    return mad * mad_scaling_factor;
}

double ref_median(const double *x, size_t n) { return ref_calc_median(x, n); }
double ref_madf(const double *x, size_t n, double med) { return ref_calc_madf(x, n, med); }

/* One read's statistics as process_move_table_paf computes them (src/gmove.cpp:751-771): a zero-initialised vector, pA where
 * pa_min <= pA <= pa_max, the UPPER median, 1.4826 * median(|x - med|), clamped to >= 1.0. out[0] = median, out[1] = calc_madf's
 * value, out[2] = the clamped MAD the normalisation divides by. */
void ref_read_medmad(const int16_t *raw, size_t n, double digitisation, double offset, double range, double pa_min, double pa_max, double *out) {
    double *x = (double *)calloc(n ? n : 1, sizeof(double));                 /* std::vector<double> raw_signal(len): zeros */
    for (size_t i = 0; i < n; i++) {
        double pA = ((raw[i]) + (offset)) * ((range) / (digitisation));     /* TO_PICOAMPS, src/poregen.h:30 */
        if (pA < pa_min || pA > pa_max) continue;                           /* src/gmove.cpp:756-758 */
        x[i] = pA;
    }
    const double med = ref_calc_median(x, n);                               /* :763 */
    double mad = ref_calc_madf(x, n, med);                                  /* :767 */
    out[0] = med; out[1] = mad;
    out[2] = (mad > 1.0) ? mad : 1.0;                                       /* :771 */
    free(x);
}
