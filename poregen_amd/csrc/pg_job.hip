// pg_job.hip -- one gmove job over several GPUs of one node from ONE host process (include/pgmove.h: pg_job_*).
//
// The reference is a single process whose only parallel driver is work_db (src/thread.c:119-132): a batch split over
// worker threads, each taking a contiguous range. This is that shape with one MI355X per worker: a batch's reads are cut
// into contiguous PAF-ordered shards, rank g works shard g through its own pg_ctx on its own host thread, and the one
// dependency between shards -- how many accepted events of a k-mer precede a shard (src/gmove.cpp:925-927) -- is one
// ncclAllGather of uint64[n_slots] per batch over xGMI, in place in every rank's receive buffer:
//     row 0        accepted events of all EARLIER batches (the job's running total; the previous batch's job_total)
//     row 1 + g    accepted events of shard g of this batch (written by rank g's counting kernels themselves)
// pg_collect_gathered(buf, n + 1, g + 1) then sums the rows below its own on the device. The collective runs on a
// communication stream of its own: the statistics of every read (the streaming kernel, independent of the exchange) are
// queued on the compute stream behind its ISSUE and hide its latency. Host code only; every kernel lives in pg_kernels.hip.
// RCCL is resolved with dlopen at the first job that needs it, so libpgmove.so carries no DT_NEEDED on it (a process must
// hold one copy: PyTorch's under Python, /opt/rocm's otherwise).
#include "pg_job_rule.h"
#include "../../include/pgmove.h"
#include "pg_internal.h"
#include "pg_hostmem.h"
#include <rccl/rccl.h>

#include <algorithm>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

extern "C" void *pgi_stream(pg_ctx *c); // pg_api.hip: the stream the context's chain currently runs on
extern "C" const double *pgi_fin_dev(pg_ctx *c); // pg_api.hip: where pg_finish_deferred left the context's kept samples on its device (or null)
extern "C" uint64_t pgi_full_slots_settled(const pg_ctx *c); // pg_api.hip: k-mers complete after the last settled batch
extern "C" pg_status pgi_skip_stats(pg_ctx *c); // pg_api.hip: a deferred-statistics batch that will keep nothing needs none
extern "C" pg_status pgi_stats_gathered(pg_ctx *c, const uint64_t *all_counts, uint32_t world, uint32_t rank); // pg_api.hip: pg_stats, cancelled on the device when the rows below `rank` complete every k-mer

static thread_local std::string g_job_create_error;

namespace {

// ---- RCCL through dlopen ---------------------------------------------------------------------------------------------
struct Rccl {
    void *h = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
    bool ok() const { return h != nullptr; }
};
Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // a copy that is already in the process (PyTorch's) first, then the system one
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (int pass = 0; pass < 2 && !r.h; ++pass)
            for (const char *n : names) {
                r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
                if (r.h) break;
            }
        if (!r.h) { const char *e = dlerror(); r.why = e ? e : "librccl not found"; return; }
        auto sym = [&](const char *n) { void *p = dlsym(r.h, n); if (!p) { r.why = std::string("librccl lacks ") + n; } return p; };
        r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll"); r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather"); r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd"); r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        if (!r.why.empty()) r.h = nullptr;
    });
    return r;
}

// ---- one persistent host thread per rank --------------------------------------------------------------------------------
struct Worker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<void()> task;
    bool has_task = false, done = true, quit = false;
    void start() {
        th = std::thread([this] {
            for (;;) {
                std::function<void()> t;
                { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [this] { return has_task || quit; }); if (quit) return; t = std::move(task); has_task = false; }
                t();
                { std::lock_guard<std::mutex> lk(m); done = true; }
                cv.notify_all();
            }
        });
    }
    void post(std::function<void()> t) { { std::lock_guard<std::mutex> lk(m); task = std::move(t); has_task = true; done = false; } cv.notify_all(); }
    void wait() { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [this] { return done; }); }
    void stop() { { std::lock_guard<std::mutex> lk(m); quit = true; } cv.notify_all(); if (th.joinable()) th.join(); }
};

struct Shard { // one rank's part of the batch in flight: rebased offset arrays (a host batch's offsets start at 0)
    std::vector<uint64_t> sig_off, seq_off, op_off;
    pg_batch b{};
};

} // namespace

struct pg_job {
    uint32_t n = 0, n_slots = 0, sample_limit = 0;
    std::vector<int> devices;
    std::vector<pg_ctx *> ctx;
    std::vector<Worker> workers;
    std::vector<Shard> shard;
    bool use_rccl = false, have_batch = false;
    uint64_t full_slots_prev = 0;
    std::vector<ncclComm_t> comms;
    std::vector<uint64_t *> gbuf;          // per rank, on its device: uint64[n + 1][n_slots] (see the head of this file)
    std::vector<hipStream_t> comm_st;
    std::vector<hipEvent_t> ev_counted, ev_gathered;
    std::vector<uint64_t> host_rows;       // host exchange: uint64[n][n_slots]
    std::vector<uint64_t> host_row0;       // host exchange: the job's accepted events per slot in front of the current batch (row 0 of the device tables)
    std::vector<std::vector<uint64_t>> cuts; // per batch: n + 1 read boundaries inside the batch
    std::vector<uint64_t> batch_reads;     // per batch
    std::string err;
    // merged view (pg_job_finish)
    std::vector<uint64_t> r_counts, r_ev_off;
    BigVec64 r_samp_off;
    BigVec32 r_ev_len, r_ev_read;
    SampleVec r_samples; // huge pages, not zero-filled: every element is written by the merge
    std::vector<uint8_t> r_skipped;
    bool merged = false; pg_result merged_view{};
    bool samples_on_host = false, samples_on_dev = false, small_on_dev = false; // where the merged view's arrays are (device = md[] on the first device)
    // pg_job_model
    void *md[4] = {nullptr, nullptr, nullptr, nullptr}; size_t md_cap[4] = {0, 0, 0, 0};
};

static pg_status jfail(pg_job *j, pg_status code, const char *fmt, ...) {
    char buf[1200];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (j) j->err = buf; else g_job_create_error = buf;
    return code;
}

// fn(g) on rank g's own thread, all ranks at once; the first failing rank (lowest shard = first in PAF order) decides
static pg_status on_ranks(pg_job *j, const std::function<pg_status(uint32_t, std::string &)> &fn) {
    std::vector<pg_status> rc(j->n, PG_OK);
    std::vector<std::string> msg(j->n);
    for (uint32_t g = 0; g < j->n; ++g) j->workers[g].post([&, g] { rc[g] = fn(g, msg[g]); });
    for (uint32_t g = 0; g < j->n; ++g) j->workers[g].wait();
    for (uint32_t g = 0; g < j->n; ++g)
        if (rc[g] != PG_OK) return jfail(j, rc[g], "shard %u (device %d): %s", g, j->devices[g], msg[g].c_str());
    return PG_OK;
}

#define JHIP(j, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return jfail((j), PG_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } while (0)
#define JNCCL(j, expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) return jfail((j), PG_ERR_HIP, "%s failed: %s", #expr, rccl().GetErrorString ? rccl().GetErrorString(r_) : "?"); } while (0)

extern "C" {

const char *pg_job_last_error(const pg_job *j) { return j ? j->err.c_str() : g_job_create_error.c_str(); }
int32_t pg_job_uses_rccl(const pg_job *j) { return j && j->use_rccl; }

void pg_job_destroy(pg_job *j) {
    if (!j) return;
    for (uint32_t g = 0; g < j->ctx.size(); ++g) {
        if (j->ctx[g]) (void)pg_sync(j->ctx[g]);
    }
    // (a job whose creation failed on a device that does not exist is torn down here too: a refused hipSetDevice must not stay behind
    // as the calling thread's sticky "last error" -- the next HIP user of the thread, e.g. PyTorch, would report it as its own)
    for (uint32_t g = 0; g < j->n; ++g) {
        if (hipSetDevice(j->devices[g]) != hipSuccess) { (void)hipGetLastError(); continue; }
        if (g < j->comm_st.size() && j->comm_st[g]) { (void)hipStreamSynchronize(j->comm_st[g]); }
    }
    for (auto c : j->comms) if (c) (void)rccl().CommDestroy(c);
    for (uint32_t g = 0; g < j->n; ++g) {
        if (hipSetDevice(j->devices[g]) != hipSuccess) { (void)hipGetLastError(); continue; }
        if (g < j->comm_st.size() && j->comm_st[g]) (void)hipStreamDestroy(j->comm_st[g]);
        if (g < j->ev_counted.size() && j->ev_counted[g]) (void)hipEventDestroy(j->ev_counted[g]);
        if (g < j->ev_gathered.size() && j->ev_gathered[g]) (void)hipEventDestroy(j->ev_gathered[g]);
        if (g < j->gbuf.size() && j->gbuf[g]) (void)hipFree(j->gbuf[g]);
    }
    if (j->n) { (void)hipSetDevice(j->devices[0]); for (void *p : j->md) if (p) (void)hipFree(p); }
    for (auto c : j->ctx) if (c) pg_destroy(c);
    for (auto &w : j->workers) w.stop();
    delete j;
    (void)hipGetLastError();
}

pg_status pg_job_create(const pg_params *p, const int32_t *devices, uint32_t n, uint32_t exchange, pg_job **out) {
    if (!p || !devices || !out || n == 0 || n > 64) return jfail(nullptr, PG_ERR_INVALID_ARG, "pg_job_create: params / devices / n_devices (1..64)");
    *out = nullptr;
    if (exchange > PG_JOB_EXCHANGE_RCCL) return jfail(nullptr, PG_ERR_INVALID_ARG, "pg_job_create: unknown exchange mode %u", exchange);
    bool distinct = true;
    for (uint32_t a = 0; a < n; ++a) for (uint32_t b = a + 1; b < n; ++b) if (devices[a] == devices[b]) distinct = false;
    bool want_rccl = exchange == PG_JOB_EXCHANGE_RCCL || (exchange == PG_JOB_EXCHANGE_AUTO && distinct);
    if (want_rccl && !distinct) return jfail(nullptr, PG_ERR_INVALID_ARG, "pg_job_create: RCCL needs distinct devices (a device is listed twice)");
    if (want_rccl && !rccl().ok()) {
        if (exchange == PG_JOB_EXCHANGE_RCCL) return jfail(nullptr, PG_ERR_NO_DEVICE, "pg_job_create: RCCL is not available: %s", rccl().why.c_str());
        want_rccl = false;
    }
    pg_job *j = new pg_job();
    j->n = n; j->n_slots = p->n_slots; j->sample_limit = p->sample_limit; j->use_rccl = want_rccl;
    j->devices.assign(devices, devices + n);
    j->ctx.assign(n, nullptr); j->gbuf.assign(n, nullptr); j->comm_st.assign(n, nullptr);
    j->ev_counted.assign(n, nullptr); j->ev_gathered.assign(n, nullptr);
    j->workers = std::vector<Worker>(n); j->shard = std::vector<Shard>(n);
    for (auto &w : j->workers) w.start();
    auto bail = [&](pg_status s) { g_job_create_error = j->err; pg_job_destroy(j); return s; };
    // the contexts come up side by side (the HIP runtime takes 0.1-0.2 s per device)
    pg_status s = on_ranks(j, [&](uint32_t g, std::string &msg) -> pg_status {
        pg_params q = *p;
        q.device = j->devices[g];
        q.flags |= PG_FLAG_DEFER_STATS | PG_FLAG_ONE_STREAM;
        const pg_status st = pg_create(&q, &j->ctx[g]);
        if (st != PG_OK) { msg = pg_last_error(nullptr); return st; }
        hipError_t e = hipSetDevice(q.device);
        const size_t bytes = (size_t)(n + 1) * p->n_slots * sizeof(uint64_t);
        if (e == hipSuccess) e = hipMalloc((void **)&j->gbuf[g], bytes);
        if (e == hipSuccess) e = hipMemset(j->gbuf[g], 0, bytes);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&j->comm_st[g], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&j->ev_counted[g], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&j->ev_gathered[g], hipEventDisableTiming);
        if (e != hipSuccess) { msg = hipGetErrorString(e); return PG_ERR_HIP; }
        return PG_OK;
    });
    if (s != PG_OK) return bail(s);
    if (j->use_rccl) {
        j->comms.assign(n, nullptr);
        const ncclResult_t r = rccl().CommInitAll(j->comms.data(), (int)n, j->devices.data());
        if (r != ncclSuccess) {
            jfail(j, PG_ERR_HIP, "ncclCommInitAll over %u devices failed: %s", n, rccl().GetErrorString(r));
            j->comms.clear();
            if (exchange == PG_JOB_EXCHANGE_RCCL) return bail(PG_ERR_HIP);
            j->use_rccl = false; j->err.clear();
        }
    }
    if (!j->use_rccl) { j->host_rows.assign((size_t)n * p->n_slots, 0); j->host_row0.assign(p->n_slots, 0); }
    *out = j;
    return PG_OK;
}

// phase 1 of rank g: walk, filter, count its shard; the counts land in row 1 + g of the rank's own receive buffer
static pg_status job_count_shard(pg_job *j, uint32_t g, const pg_batch *q, std::string &msg) {
    const uint32_t ns = j->n_slots;
    pg_status st = pg_count(j->ctx[g], q, j->gbuf[g] + (size_t)(1 + g) * ns, PG_LOC_DEVICE);
    if (st != PG_OK) { msg = pg_last_error(j->ctx[g]); return st; }
    hipError_t e = hipSetDevice(j->devices[g]);
    if (e == hipSuccess) e = hipEventRecord(j->ev_counted[g], (hipStream_t)pgi_stream(j->ctx[g]));
    if (e == hipSuccess && !j->use_rccl) // host exchange: this rank's row comes down behind its counting kernels
        e = hipMemcpyAsync(j->host_rows.data() + (size_t)g * ns, j->gbuf[g] + (size_t)(1 + g) * ns, ns * sizeof(uint64_t), hipMemcpyDeviceToHost,
                           (hipStream_t)pgi_stream(j->ctx[g]));
    if (e != hipSuccess) { msg = hipGetErrorString(e); return PG_ERR_HIP; }
    return PG_OK;
}

// phases 2 and 3 of a batch whose shards have been counted (phase 1: pg_job_submit / pg_job_submit_shards)
static pg_status job_exchange_and_collect(pg_job *j, std::vector<uint64_t> &&cut, uint64_t nr, const std::vector<uint64_t> &shard_ops);

pg_status pg_job_submit(pg_job *j, const pg_batch *b) {
    if (!j || !b) return PG_ERR_INVALID_ARG;
    if (b->struct_size != sizeof(pg_batch)) return jfail(j, PG_ERR_INVALID_ARG, "pg_batch.struct_size mismatch");
    if (b->location != PG_LOC_HOST) return jfail(j, PG_ERR_UNSUPPORTED, "pg_job_submit takes host batches (shards that are resident on their devices: pg_job_submit_shards)");
    const uint32_t n = j->n, nr = b->n_reads;
    if (!b->sig_off || !b->seq_off || !b->op_off) return jfail(j, PG_ERR_INVALID_ARG, "batch offsets missing");
    j->merged = false;
    // contiguous shards of about equal numbers of samples (reads differ in length; any contiguous cut is correct)
    std::vector<uint64_t> cut(n + 1, 0);
    const uint64_t total = b->sig_off[nr] - b->sig_off[0];
    for (uint32_t g = 1; g < n; ++g) {
        const uint64_t want = b->sig_off[0] + total / n * g;
        uint64_t r = (uint64_t)(std::lower_bound(b->sig_off, b->sig_off + nr + 1, want) - b->sig_off);
        if (r > nr) r = nr;
        cut[g] = r < cut[g - 1] ? cut[g - 1] : r;
    }
    cut[n] = nr;
    std::vector<uint64_t> shard_ops(n, 0);
    for (uint32_t g = 0; g < n; ++g) shard_ops[g] = b->op_off[cut[g + 1]] - b->op_off[cut[g]];
    // phase 1, every rank on its own thread: settle + download its previous batch (pg_count does), stage its shard, walk,
    // filter, count -- the counts land in row 1 + g of the rank's own receive buffer
    pg_status s = on_ranks(j, [&](uint32_t g, std::string &msg) -> pg_status {
        Shard &sh = j->shard[g];
        const uint64_t lo = cut[g], hi = cut[g + 1], m = hi - lo;
        // the previous batch of this rank may still be staging out of the old offset arrays: wait for it first
        pg_status st = pg_sync(j->ctx[g]);
        if (st != PG_OK) { msg = pg_last_error(j->ctx[g]); return st; }
        sh.sig_off.resize(m + 1); sh.seq_off.resize(m + 1); sh.op_off.resize(m + 1);
        for (uint64_t i = 0; i <= m; ++i) {
            sh.sig_off[i] = b->sig_off[lo + i] - b->sig_off[lo]; sh.seq_off[i] = b->seq_off[lo + i] - b->seq_off[lo];
            sh.op_off[i] = b->op_off[lo + i] - b->op_off[lo];
        }
        pg_batch &q = sh.b;
        memset(&q, 0, sizeof q);
        q.struct_size = sizeof q; q.location = PG_LOC_HOST; q.n_reads = (uint32_t)m; q.flags = b->flags;
        q.sig = b->sig ? b->sig + b->sig_off[lo] : nullptr; q.sig_off = sh.sig_off.data();
        q.digitisation = b->digitisation ? b->digitisation + lo : nullptr; q.offset = b->offset ? b->offset + lo : nullptr; q.range = b->range ? b->range + lo : nullptr;
        q.query_start = b->query_start ? b->query_start + lo : nullptr; q.target_start = b->target_start ? b->target_start + lo : nullptr;
        q.target_end = b->target_end ? b->target_end + lo : nullptr;
        q.seq = b->seq ? b->seq + b->seq_off[lo] : nullptr; q.seq_off = sh.seq_off.data();
        q.op_n = b->op_n ? b->op_n + b->op_off[lo] : nullptr; q.op_t = b->op_t ? b->op_t + b->op_off[lo] : nullptr; q.op_off = sh.op_off.data();
        // the library stages the signal with 16-byte vectors in mind: a shard that starts at an odd multiple of 8 samples of
        // the caller's buffer is still fine for a HOST batch (it is copied to a fresh, aligned device buffer)
        return job_count_shard(j, g, &q, msg);
    });
    if (s != PG_OK) return s;
    return job_exchange_and_collect(j, std::move(cut), nr, shard_ops);
}

// The same step for shards that are ALREADY where they will be worked: shards[g] is the batch of rank g -- its reads follow rank g - 1's
// in PAF order -- as a pg_batch of its own: PG_LOC_DEVICE arrays resident on devices[g] (complete before the call), or PG_LOC_HOST.
// With device shards nothing crosses PCIe inside the step and nothing is cut or copied on the host: the C++ host drives a device-resident
// N-GPU step (north_star: "host code stays C++ ... partition the batch across the 8 GPUs").
pg_status pg_job_submit_shards(pg_job *j, const pg_batch *shards, uint32_t n_shards) {
    if (!j || !shards) return PG_ERR_INVALID_ARG;
    const uint32_t n = j->n;
    if (n_shards != n) return jfail(j, PG_ERR_INVALID_ARG, "pg_job_submit_shards: %u shards for a job of %u devices", n_shards, n);
    std::vector<uint64_t> cut(n + 1, 0), shard_ops(n, 0);
    for (uint32_t g = 0; g < n; ++g) {
        if (shards[g].struct_size != sizeof(pg_batch)) return jfail(j, PG_ERR_INVALID_ARG, "pg_batch.struct_size mismatch (shard %u)", g);
        if (shards[g].location != PG_LOC_DEVICE && shards[g].location != PG_LOC_HOST) return jfail(j, PG_ERR_INVALID_ARG, "shard %u: pg_batch.location", g);
        if (shards[g].location == PG_LOC_DEVICE && shards[g].n_reads) {
            // a device shard must live on the device that works it (advisor r05): one attribute query per shard and submit. A pointer the
            // runtime does not know (or managed / host memory) is refused here rather than faulting inside the first kernel that reads it.
            const void *probe[] = {shards[g].sig, shards[g].sig_off, shards[g].op_n, shards[g].seq};
            for (const void *ptr : probe) {
                hipPointerAttribute_t at; memset(&at, 0, sizeof at);
                const hipError_t e = ptr ? hipPointerGetAttributes(&at, ptr) : hipErrorInvalidValue;
                if (e != hipSuccess) { (void)hipGetLastError(); return jfail(j, PG_ERR_INVALID_ARG, "shard %u: a PG_LOC_DEVICE array is not device memory known to the runtime", g); }
                if (at.type != hipMemoryTypeDevice || at.device != j->devices[g])
                    return jfail(j, PG_ERR_INVALID_ARG, "shard %u: its arrays live on device %d, the job works it on device %d", g, at.device, j->devices[g]);
            }
        }
        cut[g + 1] = cut[g] + shards[g].n_reads;
        shard_ops[g] = shards[g].location == PG_LOC_DEVICE ? shards[g].n_ops : (shards[g].op_off ? shards[g].op_off[shards[g].n_reads] : 0); // (0 = unknown to the host)
    }
    j->merged = false;
    pg_status s = on_ranks(j, [&](uint32_t g, std::string &msg) -> pg_status {
        pg_status st = pg_sync(j->ctx[g]);
        if (st != PG_OK) { msg = pg_last_error(j->ctx[g]); return st; }
        j->shard[g].b = shards[g];
        return job_count_shard(j, g, &j->shard[g].b, msg);
    });
    if (s != PG_OK) return s;
    const uint64_t nr = cut[n];
    return job_exchange_and_collect(j, std::move(cut), nr, shard_ops);
}

static pg_status job_exchange_and_collect(pg_job *j, std::vector<uint64_t> &&cut_in, uint64_t nr, const std::vector<uint64_t> &shard_ops) {
    const uint32_t n = j->n, ns = j->n_slots;
    std::vector<uint64_t> cut = std::move(cut_in);
    pg_status s = PG_OK;
    // Nothing behind the completing read is touched by the reference (gmove.cpp:733-735). Was every k-mer complete before this batch?
    // (the last shard's cut of the previous batch saw every accepted event of the job; phase 1 has settled it.) Then no rank computes
    // statistics or gathers anything for this batch: its events only rank behind complete files.
    const bool done_before = j->have_batch && pg_all_slots_full_settled(j->ctx[n - 1]);
    j->full_slots_prev = j->have_batch ? pgi_full_slots_settled(j->ctx[n - 1]) : 0; // k-mers the job had completed before this batch (the last shard's cut saw them all)
    // phase 2: the exchange
    if (j->use_rccl) {
        for (uint32_t g = 0; g < n; ++g) { JHIP(j, hipSetDevice(j->devices[g])); JHIP(j, hipStreamWaitEvent(j->comm_st[g], j->ev_counted[g], 0)); }
        JNCCL(j, rccl().GroupStart());
        for (uint32_t g = 0; g < n; ++g) {
            uint64_t *rows = j->gbuf[g] + ns; // rows 1..n: sendbuff == recvbuff + rank * count, i.e. in place
            const ncclResult_t r = rccl().AllGather(rows + (size_t)g * ns, rows, ns, ncclUint64, j->comms[g], j->comm_st[g]);
            if (r != ncclSuccess) { (void)rccl().GroupEnd(); return jfail(j, PG_ERR_HIP, "ncclAllGather (rank %u) failed: %s", g, rccl().GetErrorString(r)); }
        }
        JNCCL(j, rccl().GroupEnd());
        for (uint32_t g = 0; g < n; ++g) { JHIP(j, hipSetDevice(j->devices[g])); JHIP(j, hipEventRecord(j->ev_gathered[g], j->comm_st[g])); }
    }
    else // through host memory: every rank's row has been queued for download behind its counting kernels (phase 1)
        for (uint32_t g = 0; g < n; ++g) { JHIP(j, hipSetDevice(j->devices[g])); JHIP(j, hipStreamSynchronize((hipStream_t)pgi_stream(j->ctx[g]))); }
    // phase 3, every rank on its own thread: the statistics behind the ISSUE of the collective, then the wait for it (on the
    // stream, not on the host), the cut + gather, and the new running total into row 0 for the next batch
    s = on_ranks(j, [&](uint32_t g, std::string &msg) -> pg_status {
        pg_ctx *c = j->ctx[g];
        hipStream_t st = (hipStream_t)pgi_stream(c);
        // rank-level early-out (gmove.cpp:733-735): complete before this batch, or complete below this rank INSIDE the batch (the rows of
        // the lower ranks fill every k-mer): this rank keeps nothing, so it needs no statistics. The host exchange has the table at hand
        // and decides here; with RCCL the table stays on the devices, and a rank that has rows below it queues its statistics BEHIND the
        // wait for the table, cancelled there if the table says so (pgi_stats_gathered). Rank 0 (nothing below it but the earlier
        // batches, which `done_before` covers) keeps its statistics in front of the wait, where they hide the collective.
        // PGMOVE_JOB_DEVICE_RULE=1: the host exchange takes the device's rule too (tests on one GPU).
        // ... but only where completion below this rank is PLAUSIBLE (round 5, advisor): behind the wait the statistics -- the largest
        // HBM-bound kernel -- no longer hide the collective and the slowest rank's counting chain, in every batch, for a rule that pays in
        // the one batch that completes the job. Plausible = the ops of the ranks below, spread evenly over the k-mers, could fill what the
        // earlier batches left open (ops_below / n_slots >= sample_limit * share of k-mers still open; an unknown op count counts as
        // plausible). configs[2] at the default limit: every rank behind the first; k = 9 (60 ops per k-mer and rank against a limit of
        // 1000): no rank, the statistics stay in front of the wait. UNMEASURED on more than one GPU (no such node in this pool): the rule lives
        // in pg_job_rule.h, is unit-tested there and can be overridden (PGMOVE_JOB_STATS_RULE=front|behind|auto).
        std::vector<uint64_t> reads_below(g ? g : 1, 0);
        for (uint32_t h = 0; h < g; ++h) reads_below[h] = cut[h + 1] - cut[h];
        // (PGMOVE_JOB_DEVICE_RULE=1: the rule of the device exchange on the host exchange too, every rank > 0 behind the wait: tests on one GPU)
        const char *rule_mode = getenv("PGMOVE_JOB_DEVICE_RULE") ? "behind" : getenv("PGMOVE_JOB_STATS_RULE");
        const bool behind = pg_job_stats_place(g, shard_ops.data(), reads_below.data(), ns, j->sample_limit, j->have_batch, j->full_slots_prev, rule_mode) == PG_JOB_STATS_BEHIND;
        const bool dev_rule = !done_before && behind && (j->use_rccl || getenv("PGMOVE_JOB_DEVICE_RULE") != nullptr);
        bool skip = done_before;
        if (!skip && !dev_rule && !j->use_rccl && j->sample_limit > 0) {
            skip = true;
            for (uint32_t sl = 0; sl < ns && skip; ++sl) {
                uint64_t base = j->host_row0[sl];
                for (uint32_t h = 0; h < g; ++h) base += j->host_rows[(size_t)h * ns + sl];
                if (base < j->sample_limit) skip = false;
            }
        }
        pg_status ps = PG_OK;
        if (!dev_rule) ps = skip ? pgi_skip_stats(c) : pg_stats(c);
        if (ps != PG_OK) { msg = pg_last_error(c); return ps; }
        hipError_t e = hipSetDevice(j->devices[g]);
        if (j->use_rccl) { if (e == hipSuccess) e = hipStreamWaitEvent(st, j->ev_gathered[g], 0); }
        else if (e == hipSuccess) // the table assembled on the host in phase 2
            e = hipMemcpyAsync(j->gbuf[g] + ns, j->host_rows.data(), (size_t)n * ns * sizeof(uint64_t), hipMemcpyHostToDevice, st);
        if (e != hipSuccess) { msg = hipGetErrorString(e); return PG_ERR_HIP; }
        if (dev_rule) { ps = pgi_stats_gathered(c, j->gbuf[g], n + 1, g + 1); if (ps != PG_OK) { msg = pg_last_error(c); return ps; } }
        ps = pg_collect_gathered(c, j->gbuf[g], n + 1, g + 1);
        if (ps != PG_OK) { msg = pg_last_error(c); return ps; }
        const uint64_t *d_total = nullptr;
        ps = pg_job_totals_device(c, &d_total, nullptr);
        if (ps != PG_OK) { msg = pg_last_error(c); return ps; }
        e = hipMemcpyAsync(j->gbuf[g], d_total, ns * sizeof(uint64_t), hipMemcpyDeviceToDevice, st); // row 0 for the next batch
        if (e != hipSuccess) { msg = hipGetErrorString(e); return PG_ERR_HIP; }
        return PG_OK;
    });
    if (s != PG_OK) return s;
    if (!j->use_rccl) for (uint32_t sl = 0; sl < ns; ++sl) for (uint32_t h = 0; h < n; ++h) j->host_row0[sl] += j->host_rows[(size_t)h * ns + sl];
    j->cuts.push_back(std::move(cut));
    j->batch_reads.push_back(nr);
    j->have_batch = true;
    return PG_OK;
}

// as pg_reset for the job: the next pg_job_submit starts a new job on the same devices, communicators and buffers
pg_status pg_job_reset(pg_job *j) {
    if (!j) return PG_ERR_INVALID_ARG;
    const uint32_t n = j->n, ns = j->n_slots;
    pg_status s = on_ranks(j, [&](uint32_t g, std::string &msg) -> pg_status {
        pg_status st = pg_sync(j->ctx[g]);
        if (st == PG_OK) st = pg_reset(j->ctx[g]);
        if (st != PG_OK) { msg = pg_last_error(j->ctx[g]); return st; }
        hipError_t e = hipSetDevice(j->devices[g]);
        if (e == hipSuccess && j->comm_st[g]) e = hipStreamSynchronize(j->comm_st[g]);
        if (e == hipSuccess) e = hipMemsetAsync(j->gbuf[g], 0, (size_t)(n + 1) * ns * sizeof(uint64_t), (hipStream_t)pgi_stream(j->ctx[g])); // row 0: nothing accepted so far
        if (e != hipSuccess) { msg = hipGetErrorString(e); return PG_ERR_HIP; }
        return PG_OK;
    });
    if (s != PG_OK) return s;
    std::fill(j->host_row0.begin(), j->host_row0.end(), 0); std::fill(j->host_rows.begin(), j->host_rows.end(), 0);
    j->cuts.clear(); j->batch_reads.clear(); j->have_batch = false; j->merged = false; j->full_slots_prev = 0;
    j->samples_on_host = j->samples_on_dev = j->small_on_dev = false;
    return PG_OK;
}

pg_status pg_job_sync(pg_job *j) {
    if (!j) return PG_ERR_INVALID_ARG;
    return on_ranks(j, [&](uint32_t g, std::string &msg) -> pg_status {
        const pg_status st = pg_sync(j->ctx[g]);
        if (st != PG_OK) msg = pg_last_error(j->ctx[g]);
        return st;
    });
}

int32_t pg_job_all_slots_full(pg_job *j) {
    if (!j || !j->have_batch) return 0;
    // the last shard's cut saw every accepted event of the job so far (its base is the sum of all rows below it)
    return pg_all_slots_full(j->ctx[j->n - 1]);
}

int32_t pg_job_poll(pg_job *j) {
    if (!j) return PG_ERR_INVALID_ARG;
    for (uint32_t g = 0; g < j->n; ++g) {
        const int32_t r = pg_poll(j->ctx[g]);
        if (r == 0) return 0;
        if (r < 0) return jfail(j, r, "shard %u (device %d): %s", g, j->devices[g], pg_last_error(j->ctx[g]));
    }
    return 1;
}

int32_t pg_job_all_slots_full_settled(const pg_job *j) {
    if (!j || !j->have_batch) return 0;
    return pg_all_slots_full_settled(j->ctx[j->n - 1]);
}

// md[i] (device 0) grown to hold `bytes`
static pg_status md_ensure(pg_job *j, int i, size_t bytes) {
    if (bytes + 16 <= j->md_cap[i]) return PG_OK;
    if (j->md[i]) JHIP(j, hipFree(j->md[i]));
    j->md[i] = nullptr; j->md_cap[i] = 0;
    JHIP(j, hipMalloc(&j->md[i], bytes + bytes / 8 + 64));
    j->md_cap[i] = bytes + bytes / 8 + 64;
    return PG_OK;
}

// The merged view of the job. Every rank finishes "deferred": its small arrays come to the host, its kept samples stay on its device.
// The small arrays are merged here (reference order inside a k-mer: batch, then shard = PAF line order); the samples are concatenated ON
// THE FIRST DEVICE -- north_star's "concatenate the buffers over xGMI": one peer copy per rank that sits on another device, one launch
// of (slot, batch, rank) segment copies -- and come to the host only if the caller asks for them (pg_job_finish), as one download.
// pg_job_finish_deferred leaves them there for pg_job_fetch_samples / pg_job_text / pg_job_model. A rank whose own batches had to be
// merged on the host (no room on its device) sends the whole job through the host merge, as before.
static pg_status job_finish(pg_job *j, pg_result *out, bool want_samples) {
    if (!j || !out) return PG_ERR_INVALID_ARG;
    const uint32_t n = j->n, ns = j->n_slots;
    if (j->merged) {
        if (want_samples && !j->samples_on_host && j->merged_view.n_samples) { // deferred before, wanted now: one download
            const uint64_t n_samples = j->merged_view.n_samples;
            j->r_samples.resize(n_samples);
            JHIP(j, hipSetDevice(j->devices[0]));
            const unsigned parts = n_samples * 8ull >= (64ull << 20) ? 8u : 1u;
            const uint64_t step = ((n_samples + parts - 1) / parts + 511) & ~511ull;
            std::vector<hipError_t> rc(parts, hipSuccess);
            std::vector<std::thread> pool;
            for (unsigned t = 0; t < parts; t++)
                pool.emplace_back([&, t]() {
                    const uint64_t a = std::min<uint64_t>(n_samples, t * step), b2 = std::min<uint64_t>(n_samples, a + step);
                    if (b2 <= a) return;
                    rc[t] = hipSetDevice(j->devices[0]);
                    if (rc[t] == hipSuccess) rc[t] = hipMemcpy(j->r_samples.data() + a, (const double *)j->md[3] + a, (b2 - a) * 8ull, hipMemcpyDeviceToHost);
                });
            for (auto &th : pool) th.join();
            for (hipError_t e2 : rc) JHIP(j, e2);
            j->samples_on_host = true;
            j->merged_view.samples = j->r_samples.data();
        }
        *out = j->merged_view;
        if (!want_samples && j->samples_on_dev) out->samples = nullptr; // pg_job_fetch_samples reads them on the device
        return PG_OK;
    }
    std::vector<pg_result> R(n);
    pg_status s = on_ranks(j, [&](uint32_t g, std::string &msg) -> pg_status {
        const pg_status st = pg_finish_deferred(j->ctx[g], &R[g]);
        if (st != PG_OK) msg = pg_last_error(j->ctx[g]);
        return st;
    });
    if (s != PG_OK) return s;
    std::vector<const double *> dev_src(n, nullptr);
    bool all_dev = !getenv("PGMOVE_JOB_HOST_MERGE");
    for (uint32_t g = 0; g < n; ++g) { dev_src[g] = pgi_fin_dev(j->ctx[g]); if (R[g].n_samples && !dev_src[g]) all_dev = false; }
    // a rank's own view is slot-major with its reads numbered over its shards of all batches, so the events of (slot, batch, rank)
    // are a run of that rank's slot stream
    const size_t nb = j->cuts.size();
    std::vector<std::vector<uint64_t>> rb(n, std::vector<uint64_t>(nb + 1, 0)); // rank-local read index where batch i starts
    std::vector<uint64_t> bstart(nb + 1, 0);
    for (size_t i = 0; i < nb; ++i) {
        bstart[i + 1] = bstart[i] + j->batch_reads[i];
        for (uint32_t g = 0; g < n; ++g) rb[g][i + 1] = rb[g][i] + (j->cuts[i][g + 1] - j->cuts[i][g]);
    }
    uint64_t n_events = 0, n_samples = 0;
    for (uint32_t g = 0; g < n; ++g) { n_events += R[g].n_events; n_samples += R[g].n_samples; }
    // through the host after all: every rank's samples come down first (those still on a device in slices, side by side)
    std::vector<SampleVec> host_src(all_dev ? 0 : n);
    std::vector<const double *> hsrc(n, nullptr);
    if (!all_dev) {
        s = on_ranks(j, [&](uint32_t g, std::string &msg) -> pg_status {
            if (R[g].samples || !R[g].n_samples) { hsrc[g] = R[g].samples; return PG_OK; }
            host_src[g].resize(R[g].n_samples);
            const pg_status st = pg_fetch_samples(j->ctx[g], 0, R[g].n_samples, host_src[g].data());
            if (st != PG_OK) msg = "pg_fetch_samples failed";
            hsrc[g] = host_src[g].data();
            return st;
        });
        if (s != PG_OK) return s;
    }
    j->r_counts.assign(ns, 0); j->r_ev_off.assign(ns + 1, 0); j->r_samp_off.resize(n_events + 1);
    j->r_ev_len.resize(n_events); j->r_ev_read.resize(n_events);
    if (!all_dev) j->r_samples.resize(n_samples); else { SampleVec().swap(j->r_samples); }
    j->r_skipped.assign(bstart[nb], 0);
    // where every slot's events and samples start in the merged arrays (cheap), then the copies -- hundreds of MB at large limits --
    // slot ranges of about equal sample counts side by side on a few threads
    std::vector<uint64_t> slot_e(ns + 1, 0), slot_s(ns + 1, 0);
    for (uint32_t sl = 0; sl < ns; ++sl) {
        uint64_t ne = 0, nsm = 0;
        for (uint32_t g = 0; g < n; ++g) { const uint64_t a = R[g].ev_off[sl], b2 = R[g].ev_off[sl + 1]; ne += b2 - a; if (b2 > a) nsm += R[g].samp_off[b2] - R[g].samp_off[a]; }
        slot_e[sl + 1] = slot_e[sl] + ne; slot_s[sl + 1] = slot_s[sl] + nsm;
    }
    const unsigned nt = n_samples >= (8u << 20) ? 8u : 1u;
    std::vector<std::vector<PgSeg>> tsegs(nt); // device merge: (source, destination, length) per (slot, batch, rank) run, per thread
    auto merge_slots = [&](uint32_t sl0, uint32_t sl1, unsigned tix) {
    uint64_t e = slot_e[sl0], sp = slot_s[sl0];
    std::vector<uint64_t> pos(n);
    for (uint32_t sl = sl0; sl < sl1; ++sl) {
        j->r_ev_off[sl] = e;
        for (uint32_t g = 0; g < n; ++g) pos[g] = R[g].ev_off[sl];
        for (size_t i = 0; i < nb; ++i)
            for (uint32_t g = 0; g < n; ++g) {
                const pg_result &r = R[g];
                const uint64_t end = r.ev_off[sl + 1];
                uint64_t &q = pos[g];
                const uint64_t q0 = q;
                while (q < end && r.ev_read[q] < rb[g][i + 1]) ++q;
                if (q == q0) continue;
                const uint64_t s0 = r.samp_off[q0], s1 = r.samp_off[q];
                if (!all_dev) memcpy(&j->r_samples[sp], hsrc[g] + s0, (s1 - s0) * sizeof(double));
                else if (s1 > s0) tsegs[tix].push_back(PgSeg{reinterpret_cast<const double *>((uintptr_t)((uint64_t)g << 48 | s0)), sp, s1 - s0}); // src = (rank, offset) until the sources' addresses on the first device are known
                for (uint64_t t = q0; t < q; ++t, ++e) {
                    j->r_ev_len[e] = r.ev_len[t];
                    j->r_ev_read[e] = (uint32_t)(bstart[i] + j->cuts[i][g] + (r.ev_read[t] - rb[g][i]));
                    j->r_samp_off[e] = sp + (r.samp_off[t] - s0);
                }
                sp += s1 - s0;
            }
        j->r_counts[sl] = e - j->r_ev_off[sl];
    }
    };
    {
        if (nt == 1) merge_slots(0, ns, 0);
        else {
            std::vector<std::thread> pool;
            uint32_t s0 = 0;
            for (unsigned t = 0; t < nt; t++) {
                uint32_t s1 = s0;
                const uint64_t want = slot_s[ns] / nt * (t + 1);
                while (s1 < ns && (t + 1 == nt || slot_s[s1 + 1] <= want)) s1++;
                if (s1 > s0) pool.emplace_back(merge_slots, s0, s1, t);
                s0 = s1;
            }
            for (auto &th : pool) th.join();
        }
    }
    j->r_ev_off[ns] = slot_e[ns]; j->r_samp_off[n_events] = slot_s[ns];
    for (size_t i = 0; i < nb; ++i)
        for (uint32_t g = 0; g < n; ++g) {
            const uint64_t m = j->cuts[i][g + 1] - j->cuts[i][g];
            if (m) memcpy(&j->r_skipped[bstart[i] + j->cuts[i][g]], R[g].read_skipped + rb[g][i], m);
        }
    j->samples_on_host = !all_dev; j->samples_on_dev = false;
    if (all_dev && n_samples) { // the concatenation on the first device
        const int dev0 = j->devices[0];
        JHIP(j, hipSetDevice(dev0));
        hipStream_t st0 = (hipStream_t)pgi_stream(j->ctx[0]);
        { const pg_status se = md_ensure(j, 3, n_samples * 8ull); if (se != PG_OK) return se; }
        std::vector<void *> staged(n, nullptr);
        auto drop_staged = [&]() { for (void *p : staged) if (p) (void)hipFree(p); };
        std::vector<const double *> base(n, nullptr);
        const bool force_peer = getenv("PGMOVE_JOB_FORCE_PEER") != nullptr; // tests on a one-GPU box: the staging + peer-copy branch for shards of the SAME device too
        for (uint32_t g = 0; g < n; ++g) {
            if (!R[g].n_samples) continue;
            if (j->devices[g] == dev0 && !force_peer) { base[g] = dev_src[g]; continue; }
            hipError_t e = hipMalloc(&staged[g], R[g].n_samples * 8ull); // this rank's stream over xGMI, whole
            if (e == hipSuccess) e = hipMemcpyPeerAsync(staged[g], dev0, dev_src[g], j->devices[g], R[g].n_samples * 8ull, st0);
            if (e != hipSuccess) { drop_staged(); return jfail(j, PG_ERR_HIP, "peer copy of shard %u's samples (device %d -> %d): %s", g, j->devices[g], dev0, hipGetErrorString(e)); }
            base[g] = static_cast<const double *>(staged[g]);
        }
        size_t nseg = 0;
        for (auto &v : tsegs) nseg += v.size();
        std::vector<PgSeg> segs; segs.reserve(nseg);
        for (auto &v : tsegs) for (const PgSeg &sg : v) {
            const uint64_t key = (uint64_t)(uintptr_t)sg.src;
            segs.push_back(PgSeg{base[key >> 48] + (key & ((1ull << 48) - 1)), sg.dst_off, sg.n});
        }
        void *dseg = nullptr;
        hipError_t e = hipMalloc(&dseg, segs.size() * sizeof(PgSeg) + 16);
        if (e == hipSuccess) e = hipMemcpyAsync(dseg, segs.data(), segs.size() * sizeof(PgSeg), hipMemcpyHostToDevice, st0);
        if (e == hipSuccess) e = pg_launch_merge_segments(st0, static_cast<const PgSeg *>(dseg), (uint32_t)segs.size(), static_cast<double *>(j->md[3]), n_samples);
        if (e == hipSuccess) e = hipStreamSynchronize(st0);
        if (dseg) (void)hipFree(dseg);
        drop_staged();
        if (e != hipSuccess) return jfail(j, PG_ERR_HIP, "device merge of the shards' samples: %s", hipGetErrorString(e));
        j->samples_on_dev = true;
    }
    pg_result &v = j->merged_view;
    v.n_slots = ns; v.reserved = 0; v.n_events = n_events; v.n_samples = n_samples; v.n_reads = bstart[nb];
    v.counts = j->r_counts.data(); v.ev_off = j->r_ev_off.data(); v.ev_len = j->r_ev_len.data(); v.ev_read = j->r_ev_read.data();
    v.samp_off = j->r_samp_off.data(); v.samples = j->samples_on_host ? j->r_samples.data() : nullptr; v.read_skipped = j->r_skipped.data();
    if (!n_samples) { j->samples_on_host = true; v.samples = j->r_samples.data(); }
    j->merged = true; j->small_on_dev = false;
    if (want_samples && !j->samples_on_host) return job_finish(j, out, true); // the one download
    *out = v;
    return PG_OK;
}

pg_status pg_job_finish(pg_job *j, pg_result *out) { return job_finish(j, out, true); }
pg_status pg_job_finish_deferred(pg_job *j, pg_result *out) { return job_finish(j, out, false); }

pg_status pg_job_fetch_samples(pg_job *j, uint64_t first, uint64_t n, double *dst) {
    if (!j || (!dst && n)) return PG_ERR_INVALID_ARG;
    if (!j->merged) return PG_ERR_STATE; // (no error text: several host threads may call this at once)
    if (first > j->merged_view.n_samples || n > j->merged_view.n_samples - first) return PG_ERR_INVALID_ARG;
    if (!n) return PG_OK;
    if (j->samples_on_host) { memcpy(dst, j->r_samples.data() + first, n * sizeof(double)); return PG_OK; }
    if (hipSetDevice(j->devices[0]) != hipSuccess || hipMemcpy(dst, (const double *)j->md[3] + first, n * 8ull, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return PG_ERR_HIP; }
    return PG_OK;
}

// the merged view's arrays on the first device (md[0] ev_off, md[1] samp_off, md[2] ev_len, md[3] samples): what pg_text_device / pg_model_device read
static pg_status job_device_view(pg_job *j, pg_result *R) {
    pg_status s = job_finish(j, R, false);
    if (s != PG_OK) return s;
    JHIP(j, hipSetDevice(j->devices[0]));
    const void *src[4] = {R->ev_off, R->samp_off, R->ev_len, j->samples_on_dev ? nullptr : j->r_samples.data()};
    const size_t bytes[4] = {(R->n_slots + 1) * 8ull, (R->n_events + 1) * 8ull, R->n_events * 4ull, R->n_samples * 8ull};
    for (int i = 0; i < 4; ++i) {
        if (i < 3 && j->small_on_dev) continue;
        if (i == 3 && (j->samples_on_dev || !bytes[3])) continue;
        s = md_ensure(j, i, bytes[i]);
        if (s != PG_OK) return s;
        if (bytes[i]) JHIP(j, hipMemcpy(j->md[i], src[i], bytes[i], hipMemcpyHostToDevice));
    }
    j->small_on_dev = true;
    if (bytes[3]) j->samples_on_dev = true; // (a host merge's samples are now there as well)
    return PG_OK;
}

pg_status pg_job_text(pg_job *j, pg_text_result *out) {
    if (!j || !out) return PG_ERR_INVALID_ARG;
    pg_result R;
    pg_status s = job_device_view(j, &R);
    if (s != PG_OK) return s;
    s = pg_text_device(j->ctx[0], R.n_slots, R.n_events, (const uint64_t *)j->md[0], (const uint64_t *)j->md[1], (const double *)j->md[3], out);
    if (s != PG_OK) return jfail(j, s, "%s", pg_last_error(j->ctx[0]));
    return PG_OK;
}
pg_status pg_job_fetch_text(pg_job *j, uint64_t first, uint64_t n, char *dst) { return j ? pg_fetch_text(j->ctx[0], first, n, dst) : PG_ERR_INVALID_ARG; }

pg_status pg_job_kernel_stats(pg_job *j, uint32_t shard, pg_kernel_stat *out, uint32_t cap, uint32_t *n_out) {
    if (!j || shard >= j->n) return PG_ERR_INVALID_ARG;
    const pg_status st = pg_kernel_stats(j->ctx[shard], out, cap, n_out);
    return st == PG_OK ? PG_OK : jfail(j, st, "shard %u: %s", shard, pg_last_error(j->ctx[shard]));
}

pg_status pg_job_model(pg_job *j, uint32_t flags, pg_model_result *out) {
    if (!j || !out) return PG_ERR_INVALID_ARG;
    pg_result R;
    pg_status s = job_device_view(j, &R);
    if (s != PG_OK) return s;
    s = pg_model_device(j->ctx[0], R.n_slots, (const uint64_t *)j->md[0], (const uint64_t *)j->md[1], (const uint32_t *)j->md[2], (const double *)j->md[3], flags, out);
    if (s != PG_OK) return jfail(j, s, "%s", pg_last_error(j->ctx[0]));
    return PG_OK;
}

} // extern "C"
