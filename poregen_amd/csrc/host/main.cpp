// main.cpp -- `poregen` dispatcher (src/main.c:64-103): the gmove subtool (device path) and reform (host-only).
#include <cstdio>
#include <cstring>
#include <string>
#include <sys/resource.h>
#include <sys/time.h>
#include <cstdlib>
#include <unistd.h>

#ifdef PG_REFORM_ONLY // the sanitizer build of the host-only subtool (Makefile: asan): no device code linked
static int gmove_main(int, char **) { fprintf(stderr, "[poregen] this build holds reform only\n"); return 1; }
#else
int gmove_main(int argc, char **argv);
#endif
int reform_main(int argc, char **argv);

static double realtime() { struct timeval tp; gettimeofday(&tp, nullptr); return tp.tv_sec + tp.tv_usec * 1e-6; }
static double cputime() { struct rusage r; getrusage(RUSAGE_SELF, &r); return r.ru_utime.tv_sec + r.ru_stime.tv_sec + 1e-6 * (r.ru_utime.tv_usec + r.ru_stime.tv_usec); }
static long peakrss() { struct rusage r; getrusage(RUSAGE_SELF, &r); return r.ru_maxrss * 1024; }

static int usage(FILE *fp, int code) {
    fprintf(fp, "Usage: poregen <command> [options]\n\ncommand:\n         gmove      move k-mer signal samples into k-mer buckets (MI355X implementation)\n         reform     rewrite a SAM/BAM move table as TSV or as PAF with ss:Z:\n");
    return code;
}

int main(int argc, char **argv) {
    const double t0 = realtime();
    if (argc < 2) return usage(stderr, 1);
    int ret;
    if (strcmp(argv[1], "gmove") == 0) ret = gmove_main(argc - 1, argv + 1);
    else if (strcmp(argv[1], "reform") == 0) ret = reform_main(argc - 1, argv + 1);
    else if (strcmp(argv[1], "--version") == 0 || strcmp(argv[1], "-V") == 0) { fprintf(stdout, "poregen 0.1.0 (pgmove, gfx950)\n"); return 0; }
    else if (strcmp(argv[1], "--help") == 0 || strcmp(argv[1], "-h") == 0) return usage(stdout, 0);
    else { fprintf(stderr, "[poregen] Unrecognised command %s\n", argv[1]); return usage(stderr, 1); }
    fprintf(stderr, "[%s] Version: %s\n", __func__, "0.1.0");
    fprintf(stderr, "[%s] CMD:", __func__);
    for (int i = 0; i < argc; ++i) fprintf(stderr, " %s", argv[i]);
    fprintf(stderr, "\n[%s] Real time: %.3f sec; CPU time: %.3f sec; Peak RAM: %.3f GB\n\n", __func__, realtime() - t0, cputime(), peakrss() / 1024.0 / 1024.0 / 1024.0);
    // Everything this process owns is closed and flushed by now. Tearing the HIP runtime down call by call (static destructors,
    // hsa_shut_down, one hipFree per buffer) takes 0.1-0.2 s of a run that lasts 0.3-0.7 s; the operating system reclaims the
    // same resources at exit. POREGEN_CLEAN_EXIT=1 keeps the orderly teardown (sanitizer and leak-check runs).
    if (!getenv("POREGEN_CLEAN_EXIT")) { fflush(stdout); fflush(stderr); _exit(ret); }
    return ret;
}
