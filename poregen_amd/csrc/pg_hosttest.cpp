// pg_hosttest.cpp -- host-only build of the shared (host+device) arithmetic of libpgmove, so that the
// CPU test-suite can exercise exactly the code the kernels run (pg_select.h) without a GPU.
// Not part of the product path: nothing here is reachable from libpgmove's C ABI.
#include "pg_select.h"
#include <vector>
#include <cstdint>
#include <cstring>

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

// the symmetric fast path of the selection alone: 1 = applicable (outputs set), 0 = the kernel would take the general path
extern "C" int pgt_medmad_sym(const int16_t *raw, uint64_t n, double dig, double off, double range,
                              double pa_min, double pa_max, double *med, double *mad, double *mad_raw) {
    PgReadPlan p = pg_make_plan(dig, off, range, pa_min, pa_max);
    if (p.status != 0) return p.status;
    std::vector<uint32_t> pre((size_t)(p.span > 0 ? p.span : 1), 0u);
    for (uint64_t i = 0; i < n; i++) {
        int idx = (int)raw[i] - p.c_lo;
        if (idx >= 0 && idx < p.span) pre[(size_t)idx]++;
    }
    for (int b = 1; b < p.span; b++) pre[(size_t)b] += pre[(size_t)b - 1];
    PgMedMad mm;
    if (!pg_medmad_sym(pre.data(), p, n, off, range / dig, mm)) return 0;
    *med = mm.med; *mad = mm.mad; *mad_raw = mm.mad_raw;
    return 1;
}

// the dense gathers' division (pg_select.h: pg_div_by_recip) next to the plain one: 1 = the two agree bit for bit
extern "C" int pgt_div_by_recip(double a, double b, double *q_recip, double *q_div) {
    const double y = 1.0 / b;
    *q_recip = pg_div_by_recip(a, b, y); *q_div = a / b;
    uint64_t x, z; memcpy(&x, q_recip, 8); memcpy(&z, q_div, 8);
    return x == z;
}
extern "C" int pgt_div_domain_ok(double offset, double scale) { return pg_div_domain_ok(offset, scale) ? 1 : 0; }

// ---- host parsers / writer of the CLI (poregen_amd/csrc/host) ------------------------------------------
#include "host/pg_host.h"
#include <cstring>
#include <string>

extern "C" size_t pgt_format_f8(double v, char *buf) { return pgh::format_f8(v, buf); }

// returns number of ops, or -1 on a "Bad ss" condition; fills up to cap entries
extern "C" long pgt_tokenize_ss(const char *ss, uint32_t *op_n, uint8_t *op_t, size_t cap) {
    std::vector<uint32_t> n; std::vector<uint8_t> t; std::string err;
    if (!pgh::tokenize_ss(ss, strlen(ss), n, t, err)) return -1;
    for (size_t i = 0; i < n.size() && i < cap; i++) { op_n[i] = n[i]; op_t[i] = t[i]; }
    return (long)n.size();
}

// fetch [beg,end] of `name`; returns length or -2 if the name is absent (htslib contract), -3 on I/O error
extern "C" long pgt_fastx_fetch(const char *path, const char *name, long beg, long end, char *out, size_t cap) {
    pgh::FastxIndex fx; std::string err, s;
    if (!fx.load(path, err)) return -3;
    if (!fx.fetch(name, beg, end, s)) return -2;
    if (s.size() < cap) { memcpy(out, s.data(), s.size()); out[s.size()] = 0; }
    return (long)s.size();
}

// decode one read of a SLOW5/BLOW5 file; returns len_raw_signal or -1; copies up to cap samples
extern "C" long pgt_slow5_get(const char *path, const char *read_id, double *dig_off_range, int16_t *raw, size_t cap) {
    pgh::Slow5File f; std::string err;
    if (!f.open(path, err)) return -1;
    pgh::Slow5Rec r;
    if (!f.get(read_id, r, err)) return -1;
    dig_off_range[0] = r.digitisation; dig_off_range[1] = r.offset; dig_off_range[2] = r.range;
    for (size_t i = 0; i < r.raw.size() && i < cap; i++) raw[i] = r.raw[i];
    return (long)r.raw.size();
}

extern "C" long pgt_slow5_count(const char *path) {
    pgh::Slow5File f; std::string err;
    if (!f.open(path, err)) return -1;
    return (long)f.n_reads();
}

extern "C" int pgt_parse_paf(char *line, int32_t *cols6, char *rid, char *tid, char *ss, size_t cap) {
    pgh::PafRec p;
    int rc = pgh::parse_paf_line(line, strlen(line), p);
    if (rc != 0) return rc;
    cols6[0] = p.qlen; cols6[1] = p.query_start; cols6[2] = p.query_end; cols6[3] = p.tlen; cols6[4] = p.target_start; cols6[5] = p.target_end;
    snprintf(rid, cap, "%s", p.rid.c_str()); snprintf(tid, cap, "%s", p.tid.c_str());
    if (p.ss_len < cap) { memcpy(ss, p.ss, p.ss_len); ss[p.ss_len] = 0; }
    return 0;
}

// first record of a SAM/BAM file: returns number of moves (mv entries after the stride) or -1; seq/qname copied
extern "C" long pgt_sam_first(const char *path, char *qname, char *seq, size_t cap, long long *stride_ns_ts, uint8_t *is_one, size_t mcap) {
    pgh::SamBamReader r; std::string err;
    if (!r.open(path, err)) return -1;
    pgh::MoveRec m;
    if (r.next(m, err) != 1) return -1;
    snprintf(qname, cap, "%s", m.qname.c_str()); snprintf(seq, cap, "%s", m.seq.c_str());
    stride_ns_ts[0] = m.stride; stride_ns_ts[1] = m.has_ns ? (long long)m.ns : -1; stride_ns_ts[2] = m.has_ts ? (long long)m.ts : -1;
    const long n_moves = (long)m.is_one.size();
    for (size_t i = 0; i < m.is_one.size() && i < mcap; i++) is_one[i] = m.is_one[i];
    pgh::MoveRec m2;
    if (r.next(m2, err) != 0) return -2; // the fixtures hold exactly one record
    return n_moves;
}

// pg_model.h: the fixed-point view of a "%.8f" print-out, and the host-side finishing arithmetic on hand-made moments
#include "pg_model.h"
#include <algorithm>
#include <cstdio>
extern "C" long long pgt_fixed8(double x, int *bad) { bool b = false; const long long v = pg_fixed8(x, b); *bad = b; return v; }
// values given as 1e-8 units: median text and sstdev text the library would print (same arithmetic as pg_model_format)
extern "C" void pgt_model_texts(const long long *units, size_t n, char *med, char *sd, size_t cap) {
    std::vector<long long> v(units, units + n);
    PgSlotModel m{};
    m.n = n; m.origin = n ? v[0] : 0;
    for (size_t i = 0; i < n; i++) {
        const long long d = v[i] - m.origin; const unsigned long long ad = (unsigned long long)(d < 0 ? -d : d);
        const unsigned long long h = ad >> PG_MODEL_LIMB_BITS, l = ad & ((1u << PG_MODEL_LIMB_BITS) - 1);
        m.s1 += d; m.s2_hh += h * h; m.s2_hl += h * l; m.s2_ll += l * l;
    }
    std::sort(v.begin(), v.end());
    if (n) { m.mid_lo = v[(n - 1) / 2]; m.mid_hi = v[n / 2]; }
    med[0] = sd[0] = 0;
    if (!n) return;
    snprintf(med, cap, "%.14Lg", pg_model_median(m));
    if (n < 2) snprintf(sd, cap, "nan");
    else {
        const unsigned __int128 s2 = ((unsigned __int128)m.s2_hh << 40) + ((unsigned __int128)m.s2_hl << 21) + m.s2_ll;
        const __int128 s1 = m.s1;
        pg_model_sstdev_text(n, (unsigned __int128)n * s2 - (unsigned __int128)(s1 * s1), sd, cap);
    }
}
// the sstdev text of n values whose n * sum d^2 - (sum d)^2 is num (given as two 64-bit halves), and -- for comparison -- the plain
// "%.14Lg" of the long double square root
extern "C" void pgt_sstdev_text(unsigned long long n, unsigned long long num_hi, unsigned long long num_lo, char *exact, char *plain, size_t cap) {
    const unsigned __int128 num = ((unsigned __int128)num_hi << 64) | num_lo;
    pg_model_sstdev_text(n, num, exact, cap);
    snprintf(plain, cap, "%.14Lg", sqrtl((long double)num / ((long double)n * (long double)(n - 1))) / 1e8L);
}

// ---- pg_hostmem.h: does a big SampleVec really ask for transparent huge pages? -------------------------------------------------
#include "pg_hostmem.h"
#include <fstream>
#include <sstream>
// 1 = the mapping that holds a SampleVec of n doubles carries the "hg" VmFlag (madvise(MADV_HUGEPAGE) took effect), 0 = it does not,
// -1 = /proc/self/smaps could not be read. anon_huge_kb (may be null): the mapping's AnonHugePages after a first touch.
extern "C" int pgt_samplevec_hugepage(size_t n, long *anon_huge_kb) {
    SampleVec v;
    v.resize(n);
    for (size_t i = 0; i < n; i += 512) v[i] = 1.0; // first touch
    const uintptr_t p = reinterpret_cast<uintptr_t>(v.data());
    std::ifstream f("/proc/self/smaps");
    if (!f) return -1;
    std::string line; bool inside = false; int hg = 0; long ahp = 0;
    while (std::getline(f, line)) {
        unsigned long a = 0, b = 0;
        if (sscanf(line.c_str(), "%lx-%lx ", &a, &b) == 2) { inside = p >= a && p < b; continue; } /* a mapping's header line */
        if (!inside) continue;
        if (line.rfind("AnonHugePages:", 0) == 0) ahp = atol(line.c_str() + 14);
        if (line.rfind("VmFlags:", 0) == 0) { std::istringstream is(line.substr(8)); std::string t; while (is >> t) if (t == "hg") hg = 1; }
    }
    if (anon_huge_kb) *anon_huge_kb = ahp;
    return hg;
}

// ---- corrupt-input probes (tests/test_host_corrupt.py): every read of a SLOW5/BLOW5 file / every record of a SAM/BAM file is decoded;
// returns the number decoded, or -1 with the reader's message in errbuf -- never a crash
static void put_err(const std::string &e, char *errbuf, size_t cap) { if (errbuf && cap) { const size_t n = e.size() < cap - 1 ? e.size() : cap - 1; memcpy(errbuf, e.data(), n); errbuf[n] = 0; } }
extern "C" long pgt_slow5_scan(const char *path, char *errbuf, size_t cap) {
    pgh::Slow5File f; std::string err;
    if (!f.open(path, err)) { put_err(err, errbuf, cap); return -1; }
    long n = 0;
    for (const std::string &id : f.ids_in_file_order()) {
        pgh::Slow5Rec r;
        if (!f.get(id, r, err)) { put_err(err, errbuf, cap); return -1; }
        ++n;
    }
    return n;
}
extern "C" long pgt_sambam_scan(const char *path, char *errbuf, size_t cap) {
    pgh::SamBamReader rd; std::string err;
    if (!rd.open(path, err)) { put_err(err, errbuf, cap); return -1; }
    long n = 0;
    for (;;) {
        pgh::MoveRec m;
        const int rc = rd.next(m, err);
        if (rc < 0) { put_err(err, errbuf, cap); return -1; }
        if (rc == 0) break;
        ++n;
    }
    return n;
}

// the placement of a rank's statistics relative to the count exchange (pg_job_rule.h): 1 = behind the wait, 0 = in front of it
#include "pg_job_rule.h"
extern "C" int pgt_job_stats_rule(uint32_t rank, const uint64_t *shard_ops, const uint64_t *shard_reads, uint32_t n_slots, uint64_t sample_limit,
                                  int have_batch, uint64_t full_slots_prev, const char *mode) {
    return (int)pg_job_stats_place(rank, shard_ops, shard_reads, n_slots, sample_limit, have_batch != 0, full_slots_prev, mode && *mode ? mode : nullptr);
}
