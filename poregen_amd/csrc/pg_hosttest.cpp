// pg_hosttest.cpp -- host-only build of the shared (host+device) arithmetic of libpgmove, so that the
// CPU test-suite can exercise exactly the code the kernels run (pg_select.h) without a GPU.
// Not part of the product path: nothing here is reachable from libpgmove's C ABI.
#include "pg_select.h"
#include <vector>
#include <cstdint>

extern "C" int pgt_plan(double dig, double off, double range, double pa_min, double pa_max, int32_t *out4) {
    PgReadPlan p = pg_make_plan(dig, off, range, pa_min, pa_max);
    out4[0] = p.c_lo; out4[1] = p.span; out4[2] = p.z0; out4[3] = p.status;
    return p.status;
}

// histogram + inclusive prefix on the host, then the shared selection
extern "C" int pgt_medmad(const int16_t *raw, uint64_t n, double dig, double off, double range,
                          double pa_min, double pa_max, double *med, double *mad, double *mad_raw) {
    PgReadPlan p = pg_make_plan(dig, off, range, pa_min, pa_max);
    if (p.status != 0) return p.status;
    std::vector<uint32_t> pre((size_t)(p.span > 0 ? p.span : 1), 0u);
    for (uint64_t i = 0; i < n; i++) {
        int idx = (int)raw[i] - p.c_lo;
        if (idx >= 0 && idx < p.span) pre[(size_t)idx]++;
    }
    for (int b = 1; b < p.span; b++) pre[(size_t)b] += pre[(size_t)b - 1];
    PgMedMad mm = pg_medmad_from_prefix(pre.data(), p, n, off, range / dig);
    *med = mm.med; *mad = mm.mad; *mad_raw = mm.mad_raw;
    return 0;
}
