// scatter_probe.hip -- which side of the k-mer transposition should be the random one? (HISTORY.md 9.3)
// N events with short windows (8..17 int16 samples, back to back in the source signal) go to n_slots buckets in stable order; the
// output is double per sample, bucket-major. Three ways to move them:
//   A  destination order: per kept event read {src, len, out offset} coalesced, the window at random (a 128-B line per ~25 bytes),
//      write the doubles coalesced                                              (= k_gather today)
//   B  source order: windows coalesced, the output offset handed over in source order (best case: no look-up), ~100-byte runs
//      written at random places
//   B2 source order with the look-up: dst[g] coalesced, soff[dst[g]] at random (4 bytes out of a 64 MB table)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/scatter_probe tools/probe/scatter_probe.hip ; run: tools/probe/scatter_probe [N_millions]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static inline uint64_t mix(uint64_t x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }

__device__ __forceinline__ double conv(int raw) { return ((double)raw + 13.0) * 0.1373 ; }

// one event per 8 lanes; window as one 8-byte load per lane (two samples), output as one 16-byte store per lane
template <int MODE> __global__ __launch_bounds__(256) void k_move(const int16_t *__restrict__ sig, uint64_t total, uint32_t n,
        const uint32_t *__restrict__ src, const uint32_t *__restrict__ len, const uint64_t *__restrict__ soff,   // indexed by the order the kernel walks in
        const uint32_t *__restrict__ dst, const uint32_t *__restrict__ soff_rel, const uint64_t *__restrict__ slot_base_of_dst, // B2
        double *__restrict__ out) {
    const uint32_t sub = threadIdx.x & 7u;
    const uint64_t stride = (uint64_t)gridDim.x * 32;
    const uint32_t *sig32 = reinterpret_cast<const uint32_t *>(sig);
    for (uint64_t i = (uint64_t)blockIdx.x * 32 + (threadIdx.x >> 3); i < n; i += stride) {
        const uint32_t s = src[i], l = len[i];
        uint64_t o;
        if (MODE == 2) { const uint32_t e = dst[i]; o = (uint64_t)soff_rel[e] + slot_base_of_dst[e >> 6]; } // random 4-byte read + a small table
        else o = soff[i];
        const uint32_t odd = s & 1u; const uint64_t d0 = s >> 1;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const uint32_t t = 2 * sub + 16 * ps;
            if (t < l) {
                uint2 q = make_uint2(0, 0);
                if (2 * (d0 + (t >> 1)) + 3 < total) q = *reinterpret_cast<const uint2 *>(sig32 + d0 + (t >> 1));
                const int s0 = odd ? (int)q.x >> 16 : (int)(short)(q.x & 0xffffu);
                const int s1 = odd ? (int)(short)(q.y & 0xffffu) : (int)q.x >> 16;
                if (t + 1 < l) *reinterpret_cast<double2 *>(out + o + t) = make_double2(conv(s0), conv(s1));
                else out[o + t] = conv(s0);
            }
        }
    }
}

// mode A with XCD affinity: block b belongs to XCD b % 8 (workgroups go round the XCDs) and walks only that XCD's list of events
// [xstart[x], xstart[x + 1]) of arrays that are grouped by XCD: the events of the regions (= 1 / 256 of the slots each) r with r % 8 == x
__global__ __launch_bounds__(256) void k_move_xcd(const int16_t *__restrict__ sig, uint64_t total, const uint32_t *__restrict__ xstart,
        const uint32_t *__restrict__ src, const uint32_t *__restrict__ len, const uint64_t *__restrict__ soff, double *__restrict__ out) {
    const uint32_t sub = threadIdx.x & 7u, x = blockIdx.x & 7u, j = blockIdx.x >> 3, per = gridDim.x >> 3;
    const uint32_t *sig32 = reinterpret_cast<const uint32_t *>(sig);
    const uint32_t a = xstart[x], b = xstart[x + 1];
    for (uint64_t i = (uint64_t)a + (uint64_t)j * 32 + (threadIdx.x >> 3); i < b; i += (uint64_t)per * 32) {
        const uint32_t s = src[i], l = len[i];
        const uint64_t o = soff[i];
        const uint32_t odd = s & 1u; const uint64_t d0 = s >> 1;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const uint32_t t = 2 * sub + 16 * ps;
            if (t < l) {
                uint2 q = make_uint2(0, 0);
                if (2 * (d0 + (t >> 1)) + 3 < total) q = *reinterpret_cast<const uint2 *>(sig32 + d0 + (t >> 1));
                const int s0 = odd ? (int)q.x >> 16 : (int)(short)(q.x & 0xffffu);
                const int s1 = odd ? (int)(short)(q.y & 0xffffu) : (int)q.x >> 16;
                if (t + 1 < l) *reinterpret_cast<double2 *>(out + o + t) = make_double2(conv(s0), conv(s1));
                else out[o + t] = conv(s0);
            }
        }
    }
}

int main(int argc, char **argv) {
    const uint32_t N = (argc > 1 ? atoi(argv[1]) : 16) * 1000000u, n_slots = argc > 2 ? atoi(argv[2]) : 262144;
    const uint32_t lmin = argc > 3 ? atoi(argv[3]) : 8, lspan = argc > 4 ? atoi(argv[4]) : 10;
    std::vector<uint32_t> slot(N), len(N), src(N);
    uint64_t total = 0;
    for (uint32_t i = 0; i < N; ++i) { const uint64_t h = mix(i); slot[i] = (uint32_t)(h % n_slots); len[i] = lmin + (uint32_t)((h >> 32) % lspan); src[i] = (uint32_t)total; total += len[i] + (lspan > 10 ? 40 : 0); }
    // stable counting sort by slot -> dst[i]
    std::vector<uint32_t> start(n_slots + 1, 0), dst(N);
    for (uint32_t i = 0; i < N; ++i) start[slot[i] + 1]++;
    for (uint32_t s = 0; s < n_slots; ++s) start[s + 1] += start[s];
    { std::vector<uint32_t> cur(start.begin(), start.end() - 1); for (uint32_t i = 0; i < N; ++i) dst[i] = cur[slot[i]]++; }
    // destination-ordered arrays + sample offsets
    std::vector<uint32_t> d_src(N), d_len(N); std::vector<uint64_t> d_soff(N + 1), s_soff(N);
    for (uint32_t i = 0; i < N; ++i) { d_src[dst[i]] = src[i]; d_len[dst[i]] = len[i]; }
    d_soff[0] = 0; for (uint32_t e = 0; e < N; ++e) d_soff[e + 1] = d_soff[e] + d_len[e];
    for (uint32_t i = 0; i < N; ++i) s_soff[i] = d_soff[dst[i]];
    // B2: offsets relative to a base per 64 destination events
    std::vector<uint32_t> rel(N); std::vector<uint64_t> base((N + 63) / 64);
    for (uint32_t e = 0; e < N; ++e) { if ((e & 63) == 0) base[e >> 6] = d_soff[e]; rel[e] = (uint32_t)(d_soff[e] - base[e >> 6]); }
    const uint64_t n_out = d_soff[N];
    printf("N %u events, %u slots, %.1f M samples in (%.2f GB), %.1f M out (%.2f GB)\n", N, n_slots, total / 1e6, total * 2 / 1e9, n_out / 1e6, n_out * 8 / 1e9);
    std::vector<int16_t> sig(total + 16);
    for (uint64_t i = 0; i < total; ++i) sig[i] = (int16_t)(mix(i) & 1023);
    int16_t *g_sig; uint32_t *g_src, *g_len, *g_dsrc, *g_dlen, *g_dst, *g_rel; uint64_t *g_soff, *g_dsoff, *g_base; double *g_out;
    CK(hipMalloc(&g_sig, (total + 16) * 2)); CK(hipMalloc(&g_src, N * 4ull)); CK(hipMalloc(&g_len, N * 4ull)); CK(hipMalloc(&g_dsrc, N * 4ull)); CK(hipMalloc(&g_dlen, N * 4ull));
    CK(hipMalloc(&g_dst, N * 4ull)); CK(hipMalloc(&g_rel, N * 4ull)); CK(hipMalloc(&g_soff, N * 8ull)); CK(hipMalloc(&g_dsoff, (N + 1) * 8ull)); CK(hipMalloc(&g_base, base.size() * 8)); CK(hipMalloc(&g_out, (n_out + 2) * 8));
    CK(hipMemcpy(g_sig, sig.data(), (total + 16) * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(g_src, src.data(), N * 4ull, hipMemcpyHostToDevice)); CK(hipMemcpy(g_len, len.data(), N * 4ull, hipMemcpyHostToDevice));
    CK(hipMemcpy(g_dsrc, d_src.data(), N * 4ull, hipMemcpyHostToDevice)); CK(hipMemcpy(g_dlen, d_len.data(), N * 4ull, hipMemcpyHostToDevice)); CK(hipMemcpy(g_dst, dst.data(), N * 4ull, hipMemcpyHostToDevice));
    CK(hipMemcpy(g_rel, rel.data(), N * 4ull, hipMemcpyHostToDevice)); CK(hipMemcpy(g_soff, s_soff.data(), N * 8ull, hipMemcpyHostToDevice)); CK(hipMemcpy(g_dsoff, d_soff.data(), (N + 1) * 8ull, hipMemcpyHostToDevice));
    CK(hipMemcpy(g_base, base.data(), base.size() * 8, hipMemcpyHostToDevice));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<double> ref(n_out), got(n_out);
    // A-local: mode A with every window's source folded into a small area of the signal (argv[5] MB, default 27): what the gather would cost if
    // its random reads were served by the Infinity Cache / L2 instead of HBM (outputs differ from the reference run by construction)
    {
        const uint64_t area = (uint64_t)(argc > 5 ? atoi(argv[5]) : 27) * 1000000ull / 2; // samples
        std::vector<uint32_t> loc(N);
        for (uint32_t e = 0; e < N; ++e) loc[e] = (uint32_t)(d_src[e] % (area < total - 64 ? area : total - 64));
        uint32_t *g_loc; CK(hipMalloc(&g_loc, N * 4ull)); CK(hipMemcpy(g_loc, loc.data(), N * 4ull, hipMemcpyHostToDevice));
        for (int grid : {8192, 32768}) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(a));
                hipLaunchKernelGGL(k_move<0>, dim3(grid), dim3(256), 0, 0, g_sig, total, N, g_loc, g_dlen, g_dsoff, nullptr, nullptr, nullptr, g_out);
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best;
            }
            printf("mode A-local (sources inside %.0f MB)          grid %5d: %.3f ms\n", area * 2 / 1e6, grid, best);
        }
    }
    // A-banded: mode A walking the destination events band by band -- ranks [64 b, 64 b + 64) of EVERY slot, then the next band -- instead of
    // slot by slot. A slot's events are in source order, so a band of ranks is a band of the source: the windows the chip has in flight then
    // come from a narrow range of the signal (cache lines are used by their ~2.3 windows while they are resident) while a slot's 64 events
    // still leave as one run of ~14 KB. argv[6] = band width in ranks (default 64)
    {
        const uint32_t band = argc > 6 ? atoi(argv[6]) : 64;
        std::vector<uint32_t> bsrc, blen; std::vector<uint64_t> bsoff;
        bsrc.reserve(N); blen.reserve(N); bsoff.reserve(N);
        uint32_t maxc = 0; for (uint32_t sl = 0; sl < n_slots; ++sl) maxc = std::max(maxc, start[sl + 1] - start[sl]);
        for (uint32_t b0 = 0; b0 < maxc; b0 += band)
            for (uint32_t sl = 0; sl < n_slots; ++sl)
                for (uint32_t e = start[sl] + b0; e < start[sl + 1] && e < start[sl] + b0 + band; ++e) { bsrc.push_back(d_src[e]); blen.push_back(d_len[e]); bsoff.push_back(d_soff[e]); }
        uint32_t *g_bsrc, *g_blen; uint64_t *g_bsoff;
        CK(hipMalloc(&g_bsrc, N * 4ull)); CK(hipMalloc(&g_blen, N * 4ull)); CK(hipMalloc(&g_bsoff, N * 8ull));
        CK(hipMemcpy(g_bsrc, bsrc.data(), N * 4ull, hipMemcpyHostToDevice)); CK(hipMemcpy(g_blen, blen.data(), N * 4ull, hipMemcpyHostToDevice)); CK(hipMemcpy(g_bsoff, bsoff.data(), N * 8ull, hipMemcpyHostToDevice));
        for (int grid : {8192, 32768}) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipMemset(g_out, 0, n_out * 8));
                CK(hipEventRecord(a));
                hipLaunchKernelGGL(k_move<0>, dim3(grid), dim3(256), 0, 0, g_sig, total, N, g_bsrc, g_blen, g_bsoff, nullptr, nullptr, nullptr, g_out);
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best;
            }
            printf("mode A-banded (bands of %u ranks across all slots) grid %5d: %.3f ms\n", band, grid, best);
        }
    }
    // A-bucket: what the gather would see if pass A of the partition carried every event's int16 window into its region's bucket (region =
    // 1 / 256 of the slots, ~1.5 MB of windows at the K9 shape, laid out in source order inside the bucket): sources compact per region.
    // (a) walked as mode A (workgroups round the XCDs: every XCD's L2 sees every region), (b) with XCD affinity (k_move_xcd): a region's
    // windows are read by ONE XCD, whose 4 MB L2 holds them
    {
        const uint32_t R = 256, per_r = (n_slots + R - 1) / R;
        std::vector<uint64_t> rbytes(R + 1, 0);          // bucket sizes in samples, then bases
        for (uint32_t i = 0; i < N; ++i) rbytes[slot[i] / per_r + 1] += len[i] + (len[i] & 1); // (even lengths keep the 4-byte alignment)
        for (uint32_t r = 0; r < R; ++r) rbytes[r + 1] += rbytes[r];
        std::vector<uint64_t> fill(rbytes.begin(), rbytes.end() - 1);
        std::vector<uint32_t> bsrc_of(N);                // bucket position of source event i (source order inside the bucket)
        for (uint32_t i = 0; i < N; ++i) { const uint32_t r = slot[i] / per_r; bsrc_of[i] = (uint32_t)fill[r]; fill[r] += len[i] + (len[i] & 1); }
        std::vector<uint32_t> inv(N); for (uint32_t i = 0; i < N; ++i) inv[dst[i]] = i;   // destination event e came from source event inv[e]
        std::vector<uint32_t> asrc(N); for (uint32_t e = 0; e < N; ++e) asrc[e] = bsrc_of[inv[e]];
        uint32_t *g_asrc; CK(hipMalloc(&g_asrc, N * 4ull)); CK(hipMemcpy(g_asrc, asrc.data(), N * 4ull, hipMemcpyHostToDevice));
        for (int grid : {8192, 32768}) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(a));
                hipLaunchKernelGGL(k_move<0>, dim3(grid), dim3(256), 0, 0, g_sig, total, N, g_asrc, g_dlen, g_dsoff, nullptr, nullptr, nullptr, g_out);
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best;
            }
            printf("mode A-bucket (windows in 256 region buckets), workgroups round the XCDs   grid %5d: %.3f ms\n", grid, best);
        }
        // grouped by XCD: regions r with r % 8 == x, in region order, destination order inside
        std::vector<uint32_t> xsrc, xlen; std::vector<uint64_t> xsoff; std::vector<uint32_t> xstart(9, 0);
        xsrc.reserve(N); xlen.reserve(N); xsoff.reserve(N);
        for (uint32_t x = 0; x < 8; ++x) {
            for (uint32_t r = x; r < R; r += 8) {
                const uint32_t s0 = r * per_r, s1 = std::min(n_slots, (r + 1) * per_r);
                for (uint32_t e = start[s0]; e < start[s1]; ++e) { xsrc.push_back(asrc[e]); xlen.push_back(d_len[e]); xsoff.push_back(d_soff[e]); }
            }
            xstart[x + 1] = (uint32_t)xsrc.size();
        }
        uint32_t *g_xsrc, *g_xlen, *g_xstart; uint64_t *g_xsoff;
        CK(hipMalloc(&g_xsrc, N * 4ull)); CK(hipMalloc(&g_xlen, N * 4ull)); CK(hipMalloc(&g_xsoff, N * 8ull)); CK(hipMalloc(&g_xstart, 9 * 4));
        CK(hipMemcpy(g_xsrc, xsrc.data(), N * 4ull, hipMemcpyHostToDevice)); CK(hipMemcpy(g_xlen, xlen.data(), N * 4ull, hipMemcpyHostToDevice));
        CK(hipMemcpy(g_xsoff, xsoff.data(), N * 8ull, hipMemcpyHostToDevice)); CK(hipMemcpy(g_xstart, xstart.data(), 9 * 4, hipMemcpyHostToDevice));
        for (int grid : {2048, 8192, 32768}) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(a));
                hipLaunchKernelGGL(k_move_xcd, dim3(grid), dim3(256), 0, 0, g_sig, total, g_xstart, g_xsrc, g_xlen, g_xsoff, g_out);
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best;
            }
            printf("mode A-bucket with XCD affinity (a region's windows are read by one XCD)    grid %5d: %.3f ms\n", grid, best);
        }
    }
    for (int mode = 0; mode < 3; ++mode) {
        for (int grid : {8192, 32768}) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipMemset(g_out, 0, n_out * 8));
                CK(hipEventRecord(a));
                if (mode == 0) hipLaunchKernelGGL(k_move<0>, dim3(grid), dim3(256), 0, 0, g_sig, total, N, g_dsrc, g_dlen, g_dsoff, nullptr, nullptr, nullptr, g_out);
                if (mode == 1) hipLaunchKernelGGL(k_move<1>, dim3(grid), dim3(256), 0, 0, g_sig, total, N, g_src, g_len, g_soff, nullptr, nullptr, nullptr, g_out);
                if (mode == 2) hipLaunchKernelGGL(k_move<2>, dim3(grid), dim3(256), 0, 0, g_sig, total, N, g_src, g_len, nullptr, g_dst, g_rel, g_base, g_out);
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best;
            }
            CK(hipMemcpy(got.data(), g_out, n_out * 8, hipMemcpyDeviceToHost));
            if (mode == 0 && grid == 8192) ref = got;
            const bool same = got == ref;
            printf("mode %s grid %5d: %.3f ms  (%.2f TB/s on in+out bytes)  %s\n", mode == 0 ? "A  dest order, random window reads " : (mode == 1 ? "B  source order, random run writes " : "B2 source order + random soff read "),
                   grid, best, (total * 2 + n_out * 8) / (best * 1e-3) / 1e12, same ? "same" : "DIFFERENT");
        }
    }
    return 0;
}
