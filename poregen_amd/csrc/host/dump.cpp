// dump.cpp -- k-mer list and the dump directory writer (exact "%.8f" text, -d delimiters, freq.txt).
#include "pg_host.h"
#include "../pg_model.h"
#include <chrono>
#include <cmath>

#include <atomic>
#include <charconv>
#include <cstdio>
#include <cstring>
#include <dirent.h>
#include <fstream>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>

namespace pgh {

static void gen_rec(const char *set, std::string &prefix, int k, std::vector<std::string> &out) {
    if ((int)prefix.size() == k) { out.push_back(prefix); return; }
    for (int i = 0; i < 4; i++) { prefix.push_back(set[i]); gen_rec(set, prefix, k, out); prefix.pop_back(); }
}
void generate_kmers(int k, bool rna, std::vector<std::string> &out) {
    // generate_kmers (src/poregen.cpp:248-267): depth-first in alphabet order => lexicographic
    std::string prefix;
    out.reserve((size_t)1 << (2 * k));
    gen_rec(rna ? "ACGU" : "ACGT", prefix, k, out);
}

int read_kmer_file(const std::string &path, int k, std::vector<std::string> &out, std::string &err) {
    // src/gmove.cpp:396-418: getline; the last character is dropped; the line must have k+1 characters
    FILE *f = fopen(path.c_str(), "r");
    if (!f) { err = "Error in opening file " + path; return 1; }
    char *line = nullptr; size_t cap = 0; ssize_t got;
    int rc = 0;
    while ((got = getline(&line, &cap, f)) != -1) {
        if (got != (ssize_t)k + 1) {
            err = "The length of kmers in " + path + " have a different value (" + std::to_string(got - 1) + ") than the kmer size " + std::to_string(k) + ".";
            rc = 2; break;
        }
        out.emplace_back(line, (size_t)got - 1);
    }
    free(line); fclose(f);
    return rc;
}

int create_dir(const char *dir_name) { // src/gmove.cpp:126-140
    struct stat st;
    if (stat(dir_name, &st) == -1) { if (mkdir(dir_name, 0700) == -1) return -2; }
    else {
        DIR *d = opendir(dir_name); size_t n = 0;
        if (d) { while (readdir(d)) n++; closedir(d); }
        if (n > 2) return -1;
    }
    return 0;
}

size_t format_f8(double v, char *buf) {
    // printf("%.8f") prints the exactly-rounded (round-half-even on the binary value) decimal expansion with 8 fractional
    // digits. pg_fixed8 (../pg_model.h) is that rounding as an integer number of 1e-8 units for |v| < 4e7: digits by integer
    // division, ~4x faster than the general routine, which stays for everything else (std::to_chars(fixed, 8) is specified
    // to produce printf's digits).
    bool general = false;
    const int64_t units = pg_fixed8(v, general);
    if (general) {
        auto r = std::to_chars(buf, buf + 380, v, std::chars_format::fixed, 8);
        return (size_t)(r.ptr - buf);
    }
    char *p = buf;
    if (std::signbit(v)) *p++ = '-'; // also "-0.00000000" for a negative value that rounds to zero, like printf
    const uint64_t mag = (uint64_t)(units < 0 ? -units : units);
    uint64_t ip = mag / 100000000u;
    uint32_t fp = (uint32_t)(mag % 100000000u);
    char tmp[20];
    int n = 0;
    do { tmp[n++] = (char)('0' + ip % 10); ip /= 10; } while (ip);
    while (n) *p++ = tmp[--n];
    *p++ = '.';
    for (int i = 7; i >= 0; --i) { p[i] = (char)('0' + fp % 10); fp /= 10; }
    return (size_t)(p + 8 - buf);
}

bool touch_dump_files(const std::string &out_dir, const std::vector<std::string> &slot_kmers, std::string &err) {
    for (size_t i = 0; i < slot_kmers.size(); i++) { // src/gmove.cpp:460-473: one fopen(...,"w") per k-mer of the slice
        const std::string path = out_dir + "/dump/" + slot_kmers[i];
        FILE *f = fopen(path.c_str(), "w");
        if (!f) { err = "Error in opening " + std::to_string(i + 1) + "th kmer-file " + path; return false; }
        fclose(f);
    }
    return true;
}

// samples / first: the samples of (at least) this slot, the index of samples[0] in the job's sample stream
static void slot_text(const DumpInput &in, uint32_t s, bool delimit, uint32_t sample_limit, std::string &out, const double *samples, uint64_t first) {
    char buf[400];
    const uint64_t a = in.ev_off[s], b = in.ev_off[s + 1];
    out.reserve((size_t)(in.samp_off[b] - in.samp_off[a]) * 13 + (delimit ? (size_t)in.n_reads : 0) + 64); // ~12 bytes per sample
    auto put_event = [&](uint64_t e) {
        const uint64_t so = in.samp_off[e], n = in.ev_len[e];
        for (uint64_t i = 0; i < n; i++) { // "%.8f," ... "%.8f;" (src/gmove.cpp:941-944)
            size_t w = format_f8(samples[so - first + i], buf);
            buf[w] = i + 1 == n ? ';' : ',';
            out.append(buf, w + 1);
        }
    };
    if (!delimit) { for (uint64_t e = a; e < b; e++) put_event(e); return; }
    // -d (src/gmove.cpp:960-962, 196-203): after every read that was not skipped, ':' goes to every file that
    // is still open; a file closes the moment its count reaches sample_limit (src/gmove.cpp:946-949)
    uint64_t closed_at = UINT64_MAX;
    if (sample_limit > 0 && b - a == sample_limit) closed_at = in.ev_read[b - 1];
    uint64_t e = a;
    for (uint64_t r = 0; r < in.n_reads; r++) {
        while (e < b && in.ev_read[e] == r) put_event(e++);
        if (!in.read_skipped[r] && r < closed_at) out.push_back(':');
        if (r >= closed_at) break;
    }
}

bool write_dump_dir(const std::string &out_dir, const std::vector<std::string> &slot_kmers, const DumpInput &in, bool delimit,
                    uint32_t sample_limit, unsigned n_threads, std::string &err) {
    std::atomic<uint32_t> next(0);
    std::atomic<bool> ok(true);
    std::string first_err;
    // the writers take RANGES of slots. With the samples still on the device (in.samples == null) a range is fetched as one piece of
    // about 8 MB -- one k-mer at a large sample_limit, thousands of them at k = 9 -- so that a copy over PCIe is never a few KB, and
    // the next range travels while this one is formatted and written (the reference prints as it goes: src/gmove.cpp:938-944)
    std::vector<uint32_t> range_end;
    {
        const uint64_t piece = 1u << 20; // samples
        uint32_t s = 0;
        while (s < in.n_slots) {
            const uint64_t a = in.samp_off[in.ev_off[s]];
            uint32_t e = s + 1;
            while (e < in.n_slots && e - s < 4096 && in.samp_off[in.ev_off[e + 1]] - a <= piece) ++e;
            range_end.push_back(e);
            s = e;
        }
    }
    auto work = [&]() {
        std::string text;
        HugeBuf piece;
        for (;;) {
            const uint32_t ri = next.fetch_add(1);
            if (ri >= range_end.size() || !ok.load()) break;
            const uint32_t s0 = ri ? range_end[ri - 1] : 0u, s1 = range_end[ri];
            const uint64_t first = in.samp_off[in.ev_off[s0]], n = in.samp_off[in.ev_off[s1]] - first;
            const double *samples = in.samples ? in.samples + first : nullptr;
            if (!in.samples && n) {
                if (!piece.grow(n * sizeof(double), 0) || !in.fetch || !in.fetch(first, n, static_cast<double *>(piece.p))) { ok = false; break; }
                samples = static_cast<const double *>(piece.p);
            }
            for (uint32_t s = s0; s < s1 && ok.load(); ++s) {
                text.clear();
                slot_text(in, s, delimit, sample_limit, text, samples, first);
                if (text.empty()) continue; // the file already exists, empty (touch_dump_files)
                const std::string path = out_dir + "/dump/" + slot_kmers[s];
                FILE *f = fopen(path.c_str(), "w");
                if (!f || fwrite(text.data(), 1, text.size(), f) != text.size()) { ok = false; if (f) fclose(f); break; }
                fclose(f);
            }
        }
    };
    if (n_threads < 1) n_threads = 1;
    std::vector<std::thread> th;
    for (unsigned t = 1; t < n_threads; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    if (!ok) { err = "error writing dump files under " + out_dir; return false; }
    // freq.txt (src/gmove.cpp:525-534)
    const std::string fp = out_dir + "/freq.txt";
    FILE *f = fopen(fp.c_str(), "w");
    if (!f) { err = "Error in opening " + fp; return false; }
    for (uint32_t s = 0; s < in.n_slots; s++) fprintf(f, "%s\t%llu\n", slot_kmers[s].c_str(), (unsigned long long)in.counts[s]);
    fclose(f);
    return true;
}

void HugeBuf::release() { if (p) munmap(p, bytes); p = nullptr; bytes = 0; }
bool HugeBuf::grow(size_t want, size_t keep) {
    if (want <= bytes) return true;
    const size_t huge = (size_t)2 << 20, nb = (want + huge - 1) & ~(huge - 1);
    void *q = mmap(nullptr, nb, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (q == MAP_FAILED) return false;
#ifdef MADV_HUGEPAGE
    if (nb >= 2 * huge) (void)madvise(q, nb, MADV_HUGEPAGE);
#endif
    if (keep) memcpy(q, p, keep);
    release();
    p = q; bytes = nb;
    return true;
}

bool write_dump_dir_text(const std::string &out_dir, const std::vector<std::string> &slot_kmers, const TextInput &in, unsigned n_threads, std::string &err) {
    std::vector<uint32_t> range_end; // ranges of slots of about 8 MB of text: one copy over PCIe each
    for (uint32_t s = 0; s < in.n_slots;) {
        uint32_t e = s + 1;
        while (e < in.n_slots && e - s < 4096 && in.slot_off[e + 1] - in.slot_off[s] <= (8u << 20)) ++e;
        range_end.push_back(e);
        s = e;
    }
    std::atomic<uint32_t> next(0);
    std::atomic<bool> ok(true);
    const bool probe = getenv("POREGEN_DUMP_PROBE") != nullptr; // thread-seconds in the fetches and in the file writes, to stderr
    std::atomic<uint64_t> ns_fetch(0), ns_write(0);
    auto now = []() { return std::chrono::steady_clock::now(); };
    auto work = [&]() {
        HugeBuf hb;
        for (;;) {
            const uint32_t ri = next.fetch_add(1);
            if (ri >= range_end.size() || !ok.load()) break;
            const uint32_t s0 = ri ? range_end[ri - 1] : 0u, s1 = range_end[ri];
            const uint64_t first = in.slot_off[s0], n = in.slot_off[s1] - first;
            if (!n) continue; // the files already exist, empty (touch_dump_files)
            if (!hb.grow(n, 0)) { ok = false; break; }
            char *const piece = static_cast<char *>(hb.p);
            const auto t0 = now();
            if (!in.fetch(first, n, piece)) { ok = false; break; }
            const auto t1 = now();
            if (probe) ns_fetch += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count();
            struct Stamp { std::atomic<uint64_t> &acc; std::chrono::steady_clock::time_point t; bool on;
                           ~Stamp() { if (on) acc += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t).count(); } } stamp{ns_write, t1, probe};
            for (uint32_t s = s0; s < s1; ++s) {
                const uint64_t a = in.slot_off[s] - first, b = in.slot_off[s + 1] - first;
                if (b == a) continue;
                const std::string path = out_dir + "/dump/" + slot_kmers[s];
                FILE *f = fopen(path.c_str(), "w");
                if (!f || fwrite(piece + a, 1, b - a, f) != b - a) { ok = false; if (f) fclose(f); break; }
                fclose(f);
            }
        }
    };
    if (n_threads < 1) n_threads = 1;
    std::vector<std::thread> th;
    for (unsigned t = 1; t < n_threads; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    if (probe) fprintf(stderr, "[dump probe] %zu ranges on %u threads: %.3f thread-s in pg_fetch_text, %.3f thread-s in fopen/fwrite/fclose\n", range_end.size(), n_threads, ns_fetch.load() * 1e-9, ns_write.load() * 1e-9);
    if (!ok) { err = "error writing dump files under " + out_dir; return false; }
    const std::string fp = out_dir + "/freq.txt"; // src/gmove.cpp:525-534
    FILE *f = fopen(fp.c_str(), "w");
    if (!f) { err = "Error in opening " + fp; return false; }
    for (uint32_t s = 0; s < in.n_slots; s++) fprintf(f, "%s\t%llu\n", slot_kmers[s].c_str(), (unsigned long long)in.counts[s]);
    fclose(f);
    return true;
}

} // namespace pgh
