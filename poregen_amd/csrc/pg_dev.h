// pg_dev.h -- device-side helpers shared by the kernel files of libpgmove (pg_kernels.hip, pg_place.hip): wave primitives,
// the owner / window look-ups of a kept event, the lane-group gather, the checked-launch macros. Not installed.
#pragma once
#include "pg_internal.h"
#include <hip/hip_ext.h>

#define WAVE 64

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share an L2; observed, speed only -- MI355X_MICROARCH.md, "Workgroup
// dispatch"). The tile kernels write runs of ~128-250 bytes per (tile, digit) whose ends share 128-byte lines with the runs of the NEIGHBOURING
// tiles: dealt round-robin, the two halves of such a line are written through two different L2s as partial lines. This maps block b of n to
// work item (b % 8) * (n / 8) + b / 8, so that an XCD works on a CONTIGUOUS range of tiles and neighbouring runs meet in one L2; the n % 8
// items at the end keep their own number. A bijection of [0, n) for every n; correctness never depends on the placement.
#ifndef PG_XCD_REMAP
#define PG_XCD_REMAP 1
#endif
__device__ __forceinline__ uint32_t xcd_contiguous(uint32_t b, uint32_t n) {
#if PG_XCD_REMAP
    const uint32_t q = n >> 3;
    return b < (q << 3) ? (b & 7u) * q + (b >> 3) : b;
#else
    (void)n; return b;
#endif
}

// Inclusive wave64 scan with DPP row shifts + row broadcasts (no LDS traffic, 6 VALU ops): rows of 16 lanes
// are scanned with row_shr:1/2/4/8, then lane 15 of each row is broadcast into the next row (rows 1,3) and
// lane 31 into rows 2,3.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ uint32_t dpp_zero(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xF, true);
}
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
    v += dpp_zero<0x111, 0xF>(v); // row_shr:1
    v += dpp_zero<0x112, 0xF>(v); // row_shr:2
    v += dpp_zero<0x114, 0xF>(v); // row_shr:4
    v += dpp_zero<0x118, 0xF>(v); // row_shr:8
    v += dpp_zero<0x142, 0xA>(v); // row_bcast:15 -> rows 1 and 3
    v += dpp_zero<0x143, 0xC>(v); // row_bcast:31 -> rows 2 and 3
    return v;
}
__device__ __forceinline__ uint64_t wave_incl_scan_u64(uint64_t v) {
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        uint64_t t = __shfl_up(v, d, WAVE);
        if (lane >= d) v += t;
    }
    return v;
}

__device__ __forceinline__ void report_error(const PgWalkOut &O, uint32_t r, int code) {
    O.status[r] = code;
    atomicMax(O.err, ((unsigned long long)O.batch_id << 32) | (unsigned long long)(0xFFFFFFFFu - r)); // PgWalkOut::err
}

// the read that owns op index g (g < n_ops): the read of the 64-op block's first op, then along op_off (reads without ops
// are stepped over; a read of >= 64 ops ends the probe at once)
__device__ __forceinline__ uint32_t owner_of(const PgDevBatch &B, const PgWalkOut &O, uint64_t g) {
    uint32_t r = O.blk_read[g >> 6];
    if (r >= B.n_reads) r = B.n_reads - 1; // only with a broken op_off (the batch fails anyway): stay inside the arrays
    while (r + 1 < B.n_reads && B.op_off[r + 1] <= g) ++r;
    return r;
}
// ... by binary search (k_batch_init's op-parallel part runs next to the threads that write blk_read)
__device__ __forceinline__ uint32_t owner_search(const PgDevBatch &B, uint64_t g) {
    uint32_t lo = 0, hi = B.n_reads; // first r with op_off[r + 1] > g
    while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (B.op_off[mid + 1] > g) hi = mid; else lo = mid + 1; }
    return lo < B.n_reads ? lo : B.n_reads - 1;
}

#define PG_OP_N_LIMIT (1u << 24) // an op of 2^24 samples or more is refused (PGR_ERR_RANGE): 256 of them fit a 32-bit block sum

// what the emit kernels need of the read of a kept event
struct KeptRead { uint64_t o0, sig0; uint32_t qs, L; bool generic; };
__device__ __forceinline__ KeptRead kept_read(const PgWalkOut &O, uint32_t rd) {
    const PgReadMeta *mt = O.meta + rd;
    KeptRead k; k.o0 = mt->o0; k.sig0 = mt->sig0; k.qs = (uint32_t)mt->qs; k.L = mt->L; k.generic = O.gen_flag[rd] == O.batch_id;
    return k;
}
// window start and length of a kept event (gmove.cpp:854-855) at op index g of read rd: from the generic walk's arrays, or, for a
// direct read, op_n itself and the block sums k_events left. Returns false when the sample index leaves the reference's int range.
__device__ __forceinline__ bool kept_window(const PgDevBatch &B, const PgWalkParams &W, const PgWalkOut &O, const KeptRead &kr, uint64_t g, uint32_t &start, uint32_t &len) {
    const uint64_t ge = g + W.sig_move_offset; // the event's window is that of match i + sig_move_offset
    if (kr.generic) { start = O.m_start[ge]; len = O.m_len[ge]; return true; }
    len = B.op_n[ge];
    // P(x) = sum of op_n over [x & ~255, x) = cum at the 4-op group + the ops of the group in front of x; the window starts at
    // query_start + sum of op_n over [o0, ge) = P differences + whole blocks in between
    auto P = [&](uint64_t x) {
        uint32_t s = O.cum[x >> 2];
        const uint64_t y = x & ~3ull;
        const uint32_t a = B.op_n[y], b = B.op_n[y + 1 < B.n_ops ? y + 1 : y], c = B.op_n[y + 2 < B.n_ops ? y + 2 : y]; // (unconditional: 3 loads in flight)
        const uint32_t m = (uint32_t)(x & 3);
        s += (m > 0 ? a : 0u) + (m > 1 ? b : 0u) + (m > 2 ? c : 0u);
        return s;
    };
    const uint64_t b0 = kr.o0 >> 8, b1 = ge >> 8;
    const uint32_t t0 = O.btot[b0]; // unconditional: in flight with P's loads, not behind them (a read rarely ends in the block it starts in)
    const uint32_t p0 = P(kr.o0), pge = P(ge);
    uint64_t sum;
    if (b0 == b1) sum = (uint64_t)(pge - p0);
    else {
        sum = (uint64_t)(t0 - p0) + pge;
        for (uint64_t b = b0 + 1; b < b1; ++b) sum += O.btot[b];
    }
    const uint64_t st = (uint64_t)kr.qs + sum;
    start = (uint32_t)st;
    return st + len <= 0x7fffffffull;
}

// lanes holding the same digit (among valid lanes)
__device__ __forceinline__ uint64_t match_digit(uint32_t d, bool valid, int nbits) {
    uint64_t peers = __ballot(valid);
    for (int b = 0; b < nbits; ++b) {
        const bool bit = (d >> b) & 1u;
        const uint64_t mset = __ballot(valid && bit);
        peers &= bit ? mset : ~mset;
    }
    return peers;
}

#ifndef PG_GATHER8_MEAN
#define PG_GATHER8_MEAN 32
#endif
// one kept event by a group of G lanes (sub = lane within the group, g0 = the group's first lane), in two halves so that a caller
// can have the loads of several events in flight: gather_load -- the read's calibration and statistics by five lanes and the
// window's samples by every lane, all requested together; gather_finish -- conversion and 16-byte stores
#define PG_GATHER_PASSES 4 // windows of up to 2 * G * PASSES samples have all their loads in flight (longer ones: the loop in gather_finish)
template <int P = PG_GATHER_PASSES> struct GatherRegs { uint64_t h; uint2 q[P]; };
template <int G, int P = PG_GATHER_PASSES>
__device__ __forceinline__ void gather_load(const PgDevBatch &B, uint32_t sub, uint32_t rd, uint32_t len, uint64_t src, uint64_t total, int scaling,
                                            const double *__restrict__ med, const double *__restrict__ mad, const double *__restrict__ gcal, GatherRegs<P> &R) {
    const uint32_t *__restrict__ sig32 = reinterpret_cast<const uint32_t *>(B.sig);
    if (gcal) { // offset, scale, median, MAD of the read as one 32-byte record (written by the statistics kernels): four lanes, one transaction
        R.h = sub < 4u ? reinterpret_cast<const uint64_t *>(gcal)[4ull * rd + sub] : 0ull;
    } else {
        const uint64_t *arr = sub == 0 ? reinterpret_cast<const uint64_t *>(B.off + rd) : (sub == 1 ? reinterpret_cast<const uint64_t *>(B.range + rd)
                              : (sub == 2 ? reinterpret_cast<const uint64_t *>(B.dig + rd) : (sub == 3 ? reinterpret_cast<const uint64_t *>(med + rd) : reinterpret_cast<const uint64_t *>(mad + rd))));
        R.h = sub < (scaling ? 5u : 3u) ? *arr : 0ull;
    }
    const uint32_t odd = (uint32_t)(src & 1u);
    const uint64_t d0 = src >> 1; // dword that holds sample src
#pragma unroll
    for (int ps = 0; ps < P; ++ps) {
        const uint32_t t = 2 * sub + 2 * G * ps;
        const uint64_t d = d0 + (t >> 1);
        R.q[ps] = make_uint2(0u, 0u);
        if (t < len) {
#ifdef PG_GATHER_NT
            if (2 * d + 3 < total) { typedef unsigned pg_u2 __attribute__((ext_vector_type(2), aligned(4))); const pg_u2 x = __builtin_nontemporal_load(reinterpret_cast<const pg_u2 *>(sig32 + d)); R.q[ps] = make_uint2(x.x, x.y); }
#else
            if (2 * d + 3 < total) R.q[ps] = *reinterpret_cast<const uint2 *>(sig32 + d); // 8 bytes, 4-byte aligned
#endif
            else { // the last dwords of the batch: no read beyond the buffer
                const uint32_t a = (uint32_t)(uint16_t)B.sig[src + t], b2 = t + 1 < len ? (uint32_t)(uint16_t)B.sig[src + t + 1] : 0u;
                R.q[ps] = odd ? make_uint2(a << 16, b2) : make_uint2(a | (b2 << 16), 0u);
            }
        }
    }
}
template <int G, int P = PG_GATHER_PASSES>
__device__ __forceinline__ void gather_finish(const PgDevBatch &B, uint32_t sub, int g0, uint32_t len, uint64_t src, uint64_t dst, uint64_t total,
                                              int scaling, double pa_min, double pa_max, double *__restrict__ samples, const GatherRegs<P> &R, bool gcal) {
    const uint32_t *__restrict__ sig32 = reinterpret_cast<const uint32_t *>(B.sig);
    auto from = [&](uint64_t v, int k) { // the 64-bit value held by lane g0+k
        return (uint64_t)(uint32_t)__shfl((int)(uint32_t)v, g0 + k, WAVE) | ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(v >> 32), g0 + k, WAVE) << 32);
    };
    const uint32_t odd = (uint32_t)(src & 1u);
    const uint64_t d0 = src >> 1;
    const double offset = __longlong_as_double((long long)from(R.h, 0));
    // gcal: the record holds range / digitisation as the statistics used it (the same expression, PgStatRec::scale)
    const double scale = gcal ? __longlong_as_double((long long)from(R.h, 1)) : __longlong_as_double((long long)from(R.h, 1)) / __longlong_as_double((long long)from(R.h, 2));
    const double md = scaling ? __longlong_as_double((long long)from(R.h, gcal ? 2 : 3)) : 0.0;
    const double ma = scaling ? __longlong_as_double((long long)from(R.h, gcal ? 3 : 4)) : 1.0;
    auto conv = [&](int raw) {
        const double pA = ((double)raw + offset) * scale;             // TO_PICOAMPS, poregen.h:30
        double x = (pA < pa_min || pA > pa_max) ? 0.0 : pA;           // gmove.cpp:756-759
        // gmove.cpp:774. (Round 3 tried the division's reciprocal refinement once per event -- it depends on the divisor alone while
        // v_div_scale leaves the operands as they are -- and a multiplication + two FMAs per sample: bit-identical, 5 % SLOWER at k = 9
        // (1027 vs 971 us): the gather is bound by the 64-byte sectors its windows pull over the fabric, not by FP64 issue.)
        if (scaling) x = (x - md) / ma;
        return x;
    };
    auto emit2 = [&](uint32_t t, const uint2 &qq) { // this lane's two samples t, t+1 = halves of dwords d, d+1
        const int s0 = odd ? (int)qq.x >> 16 : (int)(short)(qq.x & 0xffffu);
        const int s1 = odd ? (int)(short)(qq.y & 0xffffu) : (int)qq.x >> 16;
        const double x0 = conv(s0);
        if (t + 1 < len) {
            const double x1 = conv(s1);
            *reinterpret_cast<double2 *>(samples + dst + t) = make_double2(x0, x1); // 16 bytes, 8-byte aligned (streaming "nt" stores: measured, slower -- 21.0 -> 23.5 us, 260 -> 368 us at 2.1 M events)
        } else samples[dst + t] = x0;
    };
#pragma unroll
    for (int ps = 0; ps < P; ++ps) { const uint32_t t = 2 * sub + 2 * G * ps; if (t < len) emit2(t, R.q[ps]); }
    for (uint32_t t = 2 * sub + 2 * G * P; t < len; t += 2 * G) { // very long windows (--margin, --max_dur)
        const uint64_t d = d0 + (t >> 1);
        uint2 qq;
        if (2 * d + 3 < total) qq = *reinterpret_cast<const uint2 *>(sig32 + d);
        else {
            const uint32_t a = (uint32_t)(uint16_t)B.sig[src + t], b2 = t + 1 < len ? (uint32_t)(uint16_t)B.sig[src + t + 1] : 0u;
            qq = odd ? make_uint2(a << 16, b2) : make_uint2(a | (b2 << 16), 0u);
        }
        emit2(t, qq);
    }
}
template <int G, int P = PG_GATHER_PASSES>
__device__ __forceinline__ void gather_one(const PgDevBatch &B, uint32_t sub, int g0, uint32_t rd, uint32_t len, uint64_t src, uint64_t dst,
                                           uint64_t total, int scaling, double pa_min, double pa_max,
                                           const double *__restrict__ med, const double *__restrict__ mad, const double *__restrict__ gcal, double *__restrict__ samples) {
    GatherRegs<P> R;
    gather_load<G, P>(B, sub, rd, len, src, total, scaling, med, mad, gcal, R);
    gather_finish<G, P>(B, sub, g0, len, src, dst, total, scaling, pa_min, pa_max, samples, R, gcal != nullptr);
}

// checked launches: a rejected launch (bad configuration, missing code object, a sticky earlier error) must fail the batch instead of
// leaving the previous batch's results in the buffers (hipGetLastError is per host thread and sticky: an unrelated earlier failure is
// cleared first). PG_FLAG_PROFILE: the first launch behind prof_begin carries the pair of events ITSELF (hipExtLaunchKernelGGL: the
// dispatch's own start / end time stamps, what rocprofv3's kernel trace reads) instead of standing between two recorded events
#define PG_LAUNCH(kernel, grid, block, shmem, stream, ...) do { (void)hipGetLastError(); \
    if (pg_prof_start) { hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, pg_prof_start, pg_prof_stop, 0, __VA_ARGS__); pg_prof_start = nullptr; pg_prof_stop = nullptr; } \
    else hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__); \
    const hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return e_; } while (0)
#define PG_HIP(expr) do { const hipError_t e_ = (expr); if (e_ != hipSuccess) return e_; } while (0)

