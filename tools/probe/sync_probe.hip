// sync_probe.hip -- what ordering two streams of one device against each other costs on this runtime (round 5: the kernel trace shows 4.6 us
// behind an event record and 5.9 us in front of a kernel that waits for another stream's event; every other kernel-to-kernel gap is 0.0).
// hipcc --offload-arch=gfx950 -O2 -o sync_probe tools/probe/sync_probe.hip && ./sync_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_work(float *p, int iters) { // a few microseconds on a handful of CUs
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    p[threadIdx.x] = v;
}
__global__ void k_set(unsigned long long *flag, unsigned long long v) { if (threadIdx.x == 0) __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// waits (bounded: ~0.2 s) until *flag >= v; a single wave
__global__ void k_wait(const unsigned long long *flag, unsigned long long v, int *timed_out) {
    if (threadIdx.x != 0) return;
    for (long i = 0; i < 2000000; ++i) {
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= v) return;
        __builtin_amdgcn_s_sleep(8);
    }
    *timed_out = 1;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const int N = 2000, IT = 2000;
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    float *a, *b; CK(hipMalloc(&a, 4096)); CK(hipMalloc(&b, 4096)); CK(hipMemset(a, 0, 4096)); CK(hipMemset(b, 0, 4096));
    unsigned long long *flags; CK(hipMalloc(&flags, 256)); CK(hipMemset(flags, 0, 256));
    int *to; CK(hipMalloc(&to, 4)); CK(hipMemset(to, 0, 4));
    unsigned long long *sig = nullptr;
    bool have_sig = hipExtMallocWithFlags((void **)&sig, 256, hipMallocSignalMemory) == hipSuccess;
    if (!have_sig) (void)hipGetLastError(); else CK(hipMemset(sig, 0, 256));
    hipEvent_t e1[2], e2[2], e1d[2], e2d[2];
    for (int i = 0; i < 2; i++) {
        CK(hipEventCreateWithFlags(&e1[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e2[i], hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&e1d[i], hipEventDisableTiming | hipEventDisableSystemFence)); CK(hipEventCreateWithFlags(&e2d[i], hipEventDisableTiming | hipEventDisableSystemFence));
    }
    auto run = [&](const char *name, auto body) {
        for (int w = 0; w < 50; w++) body(w);
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
        const double t0 = now();
        for (int i = 0; i < N; i++) body(50 + i);
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
        const double us = (now() - t0) / N * 1e6;
        printf("%-78s %8.2f us per iteration\n", name, us);
        return us;
    };
    unsigned long long seq = 0, seq2 = 0;
    run("s1: K K                                   (two kernels back to back)", [&](int) { k_work<<<8, 64, 0, s1>>>(a, IT); k_work<<<8, 64, 0, s1>>>(a, IT); });
    run("s1: K record K                            (event nobody waits for)", [&](int i) { k_work<<<8, 64, 0, s1>>>(a, IT); CK(hipEventRecord(e1[i & 1], s1)); k_work<<<8, 64, 0, s1>>>(a, IT); });
    run("s1: K record K                            (... hipEventDisableSystemFence)", [&](int i) { k_work<<<8, 64, 0, s1>>>(a, IT); CK(hipEventRecord(e1d[i & 1], s1)); k_work<<<8, 64, 0, s1>>>(a, IT); });
    run("s1: K rec | s2: wait K rec | s1: wait K   (serial ping-pong, default events)", [&](int i) {
        k_work<<<8, 64, 0, s1>>>(a, IT); CK(hipEventRecord(e1[i & 1], s1)); CK(hipStreamWaitEvent(s2, e1[i & 1], 0));
        k_work<<<8, 64, 0, s2>>>(b, IT); CK(hipEventRecord(e2[i & 1], s2)); CK(hipStreamWaitEvent(s1, e2[i & 1], 0)); k_work<<<8, 64, 0, s1>>>(a, IT); });
    run("s1: K rec | s2: wait K rec | s1: wait K   (... hipEventDisableSystemFence)", [&](int i) {
        k_work<<<8, 64, 0, s1>>>(a, IT); CK(hipEventRecord(e1d[i & 1], s1)); CK(hipStreamWaitEvent(s2, e1d[i & 1], 0));
        k_work<<<8, 64, 0, s2>>>(b, IT); CK(hipEventRecord(e2d[i & 1], s2)); CK(hipStreamWaitEvent(s1, e2d[i & 1], 0)); k_work<<<8, 64, 0, s1>>>(a, IT); });
    run("s1: K K K on one stream                   (the same three kernels, no second stream)", [&](int) { k_work<<<8, 64, 0, s1>>>(a, IT); k_work<<<8, 64, 0, s1>>>(b, IT); k_work<<<8, 64, 0, s1>>>(a, IT); });
    if (have_sig) run("s1: K write | s2: waitvalue K write | s1: waitvalue K   (stream memory operations)", [&](int) {
        ++seq;
        k_work<<<8, 64, 0, s1>>>(a, IT); CK(hipStreamWriteValue64(s1, sig, seq, 0)); CK(hipStreamWaitValue64(s2, sig, seq, hipStreamWaitValueGte, ~0ull));
        k_work<<<8, 64, 0, s2>>>(b, IT); CK(hipStreamWriteValue64(s2, sig + 8, seq, 0)); CK(hipStreamWaitValue64(s1, sig + 8, seq, hipStreamWaitValueGte, ~0ull)); k_work<<<8, 64, 0, s1>>>(a, IT); });
    else printf("no signal memory on this runtime\n");
    run("s1: K set | s2: spin K set | s1: spin K   (flags in device memory, one-wave kernels)", [&](int) {
        ++seq2;
        k_work<<<8, 64, 0, s1>>>(a, IT); k_set<<<1, 64, 0, s1>>>(flags, seq2); k_wait<<<1, 64, 0, s2>>>(flags, seq2, to);
        k_work<<<8, 64, 0, s2>>>(b, IT); k_set<<<1, 64, 0, s2>>>(flags + 8, seq2); k_wait<<<1, 64, 0, s1>>>(flags + 8, seq2, to); k_work<<<8, 64, 0, s1>>>(a, IT); });
    // the steady-state shape of the library: two pipelines that exchange one dependency per batch in each direction, neither waiting in steady state
    int h_to = 0; CK(hipMemcpy(&h_to, to, 4, hipMemcpyDeviceToHost));
    printf("spin kernels timed out: %d\n", h_to);
    return 0;
}
