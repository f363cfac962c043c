// mem_probe.hip -- the memory envelope of the k = 9 gather's access pattern, without the gather (round 4):
//   W   streaming stores of D doubles, 16 bytes per lane, 1 KB per wave instruction, consecutive
//   R   E "windows" of LEN int16 samples at random places of a SRC-byte source (each lane one sample, 2-byte loads), summed into a sink
//   RW  both in one kernel: every output pair reads its two samples from the random windows and stores 16 bytes
// usage: mem_probe [events=15500000] [len=12] [src_mb=400] [local_mb=0: windows confined to this many MB if > 0]
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/mem_probe tools/probe/mem_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_write(double *out, uint64_t npairs) {
    for (uint64_t p = (uint64_t)blockIdx.x * 256 + threadIdx.x; p < npairs; p += (uint64_t)gridDim.x * 256)
        *reinterpret_cast<double2 *>(out + 2 * p) = make_double2((double)p, 1.0);
}
// chunked like the gather: workgroup b owns pairs [b * per, (b + 1) * per)
__global__ __launch_bounds__(256) void k_write_chunked(double *out, uint64_t npairs, uint32_t per) {
    const uint64_t p0 = (uint64_t)blockIdx.x * per;
    for (uint32_t i = threadIdx.x; i < per && p0 + i < npairs; i += 256)
        *reinterpret_cast<double2 *>(out + 2 * (p0 + i)) = make_double2((double)i, 1.0);
}
template <bool STORE, bool LOAD>
__global__ __launch_bounds__(256) void k_rw(const int16_t *__restrict__ src, const uint32_t *__restrict__ starts /* per event, in samples */, uint32_t len, double *out, uint64_t npairs, uint32_t per, double *sink) {
    const uint64_t p0 = (uint64_t)blockIdx.x * per;
    double acc = 0;
    for (uint32_t i = threadIdx.x; i < per && p0 + i < npairs; i += 256) {
        const uint64_t s0 = 2 * (p0 + i);
        const uint64_t e0 = s0 / len, e1 = (s0 + 1) / len;
        int a = 1, b = 2;
        if (LOAD) { a = src[(uint64_t)starts[e0] + (s0 - e0 * len)]; b = src[(uint64_t)starts[e1] + (s0 + 1 - e1 * len)]; }
        const double x0 = ((double)a + 3.0) * 0.137, x1 = ((double)b + 3.0) * 0.137;
        if (STORE) *reinterpret_cast<double2 *>(out + s0) = make_double2(x0, x1); else acc += x0 + x1;
    }
    if (!STORE && acc == 1.2345e300) *sink = acc;
}
// the same with the gather's per-event metadata: a 16-byte record {start, len, read} instead of the 4-byte start, a random 32-byte
// calibration per event from a table of `nreads` entries (CAL), an 8-byte offset written per event (SOFF), the FP64 conversion (MATH)
template <bool CAL, bool SOFF, bool MATH>
__global__ __launch_bounds__(256) void k_meta(const int16_t *__restrict__ src, const uint4 *__restrict__ rec, const double *__restrict__ cal, uint32_t len, double *out, uint64_t *soff, uint64_t npairs, uint32_t per) {
    const uint64_t p0 = (uint64_t)blockIdx.x * per;
    if (SOFF) { const uint64_t ev0 = 2 * p0 / len, nev = 2ull * per / len; for (uint32_t i = threadIdx.x; i < nev; i += 256) soff[ev0 + i] = (ev0 + i) * len; }
    for (uint32_t i = threadIdx.x; i < per && p0 + i < npairs; i += 256) {
        const uint64_t s0 = 2 * (p0 + i);
        const uint64_t e0 = s0 / len, e1 = (s0 + 1) / len;
        const uint4 r0 = rec[e0], r1 = rec[e1];
        const int a = src[(uint64_t)r0.x + (s0 - e0 * len)], b = src[(uint64_t)r1.x + (s0 + 1 - e1 * len)];
        double4 c0 = make_double4(3.0, 0.137, 90.0, 11.0), c1 = c0;
        if (CAL) { c0 = *reinterpret_cast<const double4 *>(cal + 4ull * r0.w); c1 = *reinterpret_cast<const double4 *>(cal + 4ull * r1.w); }
        double x0 = ((double)a + c0.x) * c0.y, x1 = ((double)b + c1.x) * c1.y;
        if (MATH) { x0 = (x0 < 40.0 || x0 > 180.0) ? 0.0 : x0; x1 = (x1 < 40.0 || x1 > 180.0) ? 0.0 : x1; x0 = (x0 - c0.z) / c0.w; x1 = (x1 - c1.z) / c1.w; }
        *reinterpret_cast<double2 *>(out + s0) = make_double2(x0, x1);
    }
}
// how does the cost of the random window reads scale: with the number of LANE loads or with the bytes? S samples per lane, read as ONE
// load of 2 * S bytes (windows start at multiples of S samples here, so every load is aligned and inside one window), nothing stored
// (STORE false) or the S doubles stored as S / 2 16-byte stores (lane stride 8 * S bytes)
template <int S, bool STORE>
__global__ __launch_bounds__(256) void k_wide(const int16_t *__restrict__ src, const uint32_t *__restrict__ starts, uint32_t len, double *out, uint64_t nitems, uint32_t per, double *sink) {
    const uint64_t p0 = (uint64_t)blockIdx.x * per;
    double acc = 0;
    for (uint32_t i = threadIdx.x; i < per && p0 + i < nitems; i += 256) {
        const uint64_t s0 = (uint64_t)S * (p0 + i), e0 = s0 / len;
        const uint64_t idx = ((uint64_t)starts[e0] / S) * S + (s0 - e0 * len); // (len is a multiple of S in this mode)
        int v[S];
        if (S == 1) v[0] = src[idx];
        else if (S == 2) { const uint32_t x = *reinterpret_cast<const uint32_t *>(src + idx); v[0] = (short)x; v[1] = (int)x >> 16; }
        else if (S == 4) { const uint2 x = *reinterpret_cast<const uint2 *>(src + idx); v[0] = (short)x.x; v[1] = (int)x.x >> 16; v[2] = (short)x.y; v[3] = (int)x.y >> 16; }
        else { const uint4 x = *reinterpret_cast<const uint4 *>(src + idx); const uint32_t w[4] = {x.x, x.y, x.z, x.w};
               for (int q = 0; q < 4; ++q) { v[2 * q] = (short)w[q]; v[2 * q + 1] = (int)w[q] >> 16; } }
        double x[S];
        for (int q = 0; q < S; ++q) x[q] = ((double)v[q] + 3.0) * 0.137;
        if (STORE) { if (S == 1) out[s0] = x[0]; else for (int q = 0; q < S; q += 2) *reinterpret_cast<double2 *>(out + s0 + q) = make_double2(x[q], x[q + 1]); }
        else for (int q = 0; q < S; ++q) acc += x[q];
    }
    if (!STORE && acc == 1.2345e300) *sink = acc;
}
template <class F> static float timeit(F f, int reps = 5) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    float best = 1e9;
    for (int r = 0; r < reps; ++r) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
    return best;
}
int main(int argc, char **argv) {
    const uint64_t E = argc > 1 ? strtoull(argv[1], 0, 10) : 15500000ull;
    const uint32_t len = argc > 2 ? atoi(argv[2]) : 12;
    const uint64_t src_bytes = (argc > 3 ? strtoull(argv[3], 0, 10) : 400ull) << 20;
    const uint64_t local = (argc > 4 ? strtoull(argv[4], 0, 10) : 0ull) << 20;
    const uint64_t D = E * len, npairs = D / 2;
    int16_t *src; uint32_t *starts; double *out, *sink;
    CK(hipMalloc(&src, src_bytes + 4096)); CK(hipMemset(src, 1, src_bytes + 4096));
    CK(hipMalloc(&out, D * 8 + 64)); CK(hipMalloc(&sink, 8)); CK(hipMalloc(&starts, (E + 2) * 4));
    std::vector<uint32_t> h(E + 2);
    std::mt19937_64 rng(7);
    const uint64_t span = (local ? local : src_bytes) / 2 - len;
    for (auto &x : h) x = (uint32_t)(rng() % span);
    CK(hipMemcpy(starts, h.data(), (E + 2) * 4, hipMemcpyHostToDevice));
    const uint32_t per = 512 * len / 2 * 4; // pairs per workgroup: 2048 events' worth, as the gather's chunks
    const uint32_t grid = (uint32_t)((npairs + per - 1) / per);
    printf("events %llu len %u: %.2f GB of doubles out, %.2f GB source (windows inside %.0f MB), grid %u\n", (unsigned long long)E, len, D * 8 / 1e9, src_bytes / 1e9, (local ? local : src_bytes) / 1048576.0, grid);
    float t;
    t = timeit([&] { hipLaunchKernelGGL(k_write, dim3(8192), dim3(256), 0, 0, out, npairs); });
    printf("W  grid-stride stores          %.3f ms  %.2f TB/s\n", t, D * 8 / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL(k_write_chunked, dim3(grid), dim3(256), 0, 0, out, npairs, per); });
    printf("W  chunked stores              %.3f ms  %.2f TB/s\n", t, D * 8 / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL((k_rw<true, false>), dim3(grid), dim3(256), 0, 0, src, starts, len, out, npairs, per, sink); });
    printf("W  chunked, rw skeleton        %.3f ms  %.2f TB/s\n", t, D * 8 / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL((k_rw<false, true>), dim3(grid), dim3(256), 0, 0, src, starts, len, out, npairs, per, sink); });
    printf("R  random windows only         %.3f ms  %.1f G windows/s\n", t, E / t / 1e6);
    t = timeit([&] { hipLaunchKernelGGL((k_rw<true, true>), dim3(grid), dim3(256), 0, 0, src, starts, len, out, npairs, per, sink); });
    printf("RW random windows + stores     %.3f ms  %.2f TB/s of output\n", t, D * 8 / t / 1e9);
    if (len % 8 == 0) { // lane-load width sweep (needs windows of a multiple of 8 samples): mem_probe 11600000 16
        const uint64_t ns = D;
        #define WIDE(SS) { const uint32_t perw = per * 2 / SS; const uint32_t gridw = (uint32_t)((ns / SS + perw - 1) / perw); \
            float tr = timeit([&] { hipLaunchKernelGGL((k_wide<SS, false>), dim3(gridw), dim3(256), 0, 0, src, starts, len, out, ns / SS, perw, sink); }); \
            float tw = timeit([&] { hipLaunchKernelGGL((k_wide<SS, true>), dim3(gridw), dim3(256), 0, 0, src, starts, len, out, ns / SS, perw, sink); }); \
            printf("lane loads of %2d bytes (%d samples per lane): reads only %.3f ms, reads + stores %.3f ms\n", 2 * SS, SS, tr, tw); }
        WIDE(1) WIDE(2) WIDE(4) WIDE(8)
    }
    // metadata variants
    const uint32_t nreads = 50000;
    uint4 *rec; double *cal; uint64_t *soff;
    CK(hipMalloc(&rec, (E + 2) * 16)); CK(hipMalloc(&cal, nreads * 32)); CK(hipMalloc(&soff, (E + 2) * 8));
    { std::vector<uint4> hr(E + 2); for (uint64_t i = 0; i < E + 2; ++i) hr[i] = make_uint4(h[i], 0, len, (uint32_t)(rng() % nreads)); CK(hipMemcpy(rec, hr.data(), (E + 2) * 16, hipMemcpyHostToDevice));
      std::vector<double> hc(nreads * 4); for (uint32_t i = 0; i < nreads; ++i) { hc[4 * i] = -240.0; hc[4 * i + 1] = 0.137; hc[4 * i + 2] = 95.0; hc[4 * i + 3] = 12.5; } CK(hipMemcpy(cal, hc.data(), nreads * 32, hipMemcpyHostToDevice)); }
    t = timeit([&] { hipLaunchKernelGGL((k_meta<false, false, false>), dim3(grid), dim3(256), 0, 0, src, rec, cal, len, out, soff, npairs, per); });
    printf("RW + 16-byte records           %.3f ms\n", t);
    t = timeit([&] { hipLaunchKernelGGL((k_meta<true, false, false>), dim3(grid), dim3(256), 0, 0, src, rec, cal, len, out, soff, npairs, per); });
    printf("RW + records + random cal      %.3f ms\n", t);
    t = timeit([&] { hipLaunchKernelGGL((k_meta<true, true, false>), dim3(grid), dim3(256), 0, 0, src, rec, cal, len, out, soff, npairs, per); });
    printf("RW + records + cal + offsets   %.3f ms\n", t);
    t = timeit([&] { hipLaunchKernelGGL((k_meta<true, true, true>), dim3(grid), dim3(256), 0, 0, src, rec, cal, len, out, soff, npairs, per); });
    printf("RW + records + cal + offs + FP64 math  %.3f ms\n", t);
    t = timeit([&] { hipLaunchKernelGGL((k_meta<false, false, true>), dim3(grid), dim3(256), 0, 0, src, rec, cal, len, out, soff, npairs, per); });
    printf("RW + records + FP64 math (no cal table) %.3f ms\n", t);
    return 0;
}
