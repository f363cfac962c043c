// pg_text.hip -- the dump files' TEXT on the device (round 3).
//
// The reference prints every sample of a kept event as it goes: fprintf(f, "%.8f,") ... "%.8f;" (src/gmove.cpp:938-944) -- its dominant
// cost whenever events are being kept. Until round 2 the CLI did that on the host (exact digits from pg_fixed8, 16 threads:
// 0.08 s of a 0.34 s job at sample_limit 5000). Here the bytes are produced where the doubles are:
//   k_text_lens    one thread per kept event: the length of its text (per sample: sign, integer digits, '.', 8 digits, separator)
//   (exclusive scan of those lengths -> where every event's text starts: pg_launch_scan_u32_u64)
//   k_text_write   one thread per kept event: the characters; per slot the offset of its first event = its file's range
// The digits are those of printf: pg_fixed8 (pg_model.h) is the correctly rounded (ties to even on the exact binary value) number of
// 1e-8 units for |x| < 4e7; anything else (never seen in pA or med-MAD units) raises a flag and the caller formats on the host.
// With -d (':' after every read in every open file) the text depends on the reads, not only on the events: the host writes those.
#include "../../include/pgmove.h"
#include "pg_dev.h"
#include "pg_model.h"

// characters of "%.8f" of x without the separator: [-] digits . 8 digits
__device__ __forceinline__ uint32_t f8_len(double x, bool &bad) {
    const int64_t u = pg_fixed8(x, bad);
    uint64_t ip = (uint64_t)(u < 0 ? -u : u) / 100000000ull;
    uint32_t nd = 1;
    while (ip >= 10) { ip /= 10; ++nd; }
    return nd + 9u + (__builtin_signbit(x) ? 1u : 0u); // "-0.00000000" for a negative value that rounds to zero, like printf
}

__global__ __launch_bounds__(256) void k_text_lens(const double *__restrict__ samples, const uint64_t *__restrict__ samp_off, uint64_t n_events,
                                                   uint32_t *__restrict__ tlen, uint32_t *__restrict__ flag) {
    const uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_events) return;
    const uint64_t a = samp_off[e], b = samp_off[e + 1];
    uint32_t len = 0; bool bad = false;
    for (uint64_t i = a; i < b; ++i) len += f8_len(samples[i], bad) + 1u;
    tlen[e] = len;
    if (bad) atomicOr(flag, 1u);
}

__global__ __launch_bounds__(256) void k_text_write(const double *__restrict__ samples, const uint64_t *__restrict__ samp_off, uint64_t n_events,
                                                    const uint64_t *__restrict__ toff, char *__restrict__ text) {
    const uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_events) return;
    const uint64_t a = samp_off[e], b = samp_off[e + 1];
    char *p = text + toff[e];
    for (uint64_t i = a; i < b; ++i) {
        const double x = samples[i];
        bool bad = false;
        const int64_t u = pg_fixed8(x, bad);
        const uint64_t mag = (uint64_t)(u < 0 ? -u : u);
        uint64_t ip = mag / 100000000ull;
        uint32_t fp = (uint32_t)(mag % 100000000ull);
        if (__builtin_signbit(x)) *p++ = '-';
        uint32_t nd = 1;
        for (uint64_t t = ip; t >= 10; t /= 10) ++nd;
        for (uint32_t k = nd; k-- > 0;) { p[k] = (char)('0' + ip % 10); ip /= 10; }
        p += nd;
        *p++ = '.';
        for (int k = 7; k >= 0; --k) { p[k] = (char)('0' + fp % 10); fp /= 10; }
        p += 8;
        *p++ = i + 1 == b ? ';' : ','; // src/gmove.cpp:941-944
    }
}

// where every slot's file starts in the text: the offset of its first event (slots without events: the next one's)
__global__ __launch_bounds__(256) void k_text_slot_off(const uint64_t *__restrict__ ev_off, const uint64_t *__restrict__ toff, uint32_t n_slots, uint64_t *__restrict__ slot_toff) {
    const uint32_t s = blockIdx.x * 256 + threadIdx.x;
    if (s <= n_slots) slot_toff[s] = toff[ev_off[s]];
}

hipError_t pg_launch_text_lens(hipStream_t st, const double *samples, const uint64_t *samp_off, uint64_t n_events, uint32_t *tlen, uint32_t *flag) {
    PG_HIP(hipMemsetAsync(flag, 0, 4, st));
    if (n_events) PG_LAUNCH(k_text_lens, dim3((uint32_t)((n_events + 255) / 256)), dim3(256), 0, st, samples, samp_off, n_events, tlen, flag);
    return hipSuccess;
}
hipError_t pg_launch_text_write(hipStream_t st, const double *samples, const uint64_t *samp_off, uint64_t n_events, const uint64_t *toff, char *text,
                                const uint64_t *ev_off, uint32_t n_slots, uint64_t *slot_toff) {
    if (n_events) PG_LAUNCH(k_text_write, dim3((uint32_t)((n_events + 255) / 256)), dim3(256), 0, st, samples, samp_off, n_events, toff, text);
    PG_LAUNCH(k_text_slot_off, dim3(n_slots / 256 + 1), dim3(256), 0, st, ev_off, toff, n_slots, slot_toff);
    return hipSuccess;
}
