// io.cpp -- memory-mapped files, ASCII SLOW5 / BLOW5 reader, FASTA/FASTQ index, PAF + ss tokeniser.
#include "pg_host.h"

// zstd record compression (`make zstd=1` in the reference: /root/reference/Makefile:12-13,67 links -lzstd into slow5lib). Here libzstd is
// looked up at run time, once, so that neither bin/poregen nor the test shim carries a link-time dependency: a BLOW5 file with zstd
// records on a machine without libzstd.so.1 is refused with an error that says so, never misread.
#include <dlfcn.h>
namespace {
struct Zstd {
    size_t (*decompress)(void *, size_t, const void *, size_t) = nullptr;
    unsigned long long (*frame_size)(const void *, size_t) = nullptr;
    unsigned (*is_error)(size_t) = nullptr;
    void *(*create_dstream)() = nullptr;
    size_t (*free_dstream)(void *) = nullptr;
    size_t (*init_dstream)(void *) = nullptr;
    struct Buf { const void *src; size_t size, pos; };
    struct OBuf { void *dst; size_t size, pos; };
    size_t (*decompress_stream)(void *, OBuf *, Buf *) = nullptr;
    bool ok = false;
    Zstd() {
        void *h = nullptr;
        for (const char *n : {"libzstd.so.1", "libzstd.so"}) if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!h) return;
        decompress = (decltype(decompress))dlsym(h, "ZSTD_decompress");
        frame_size = (decltype(frame_size))dlsym(h, "ZSTD_getFrameContentSize");
        is_error = (decltype(is_error))dlsym(h, "ZSTD_isError");
        create_dstream = (decltype(create_dstream))dlsym(h, "ZSTD_createDStream");
        free_dstream = (decltype(free_dstream))dlsym(h, "ZSTD_freeDStream");
        init_dstream = (decltype(init_dstream))dlsym(h, "ZSTD_initDStream");
        decompress_stream = (decltype(decompress_stream))dlsym(h, "ZSTD_decompressStream");
        ok = decompress && frame_size && is_error && create_dstream && free_dstream && init_dstream && decompress_stream;
    }
};
const Zstd &zstd() { static const Zstd z; return z; }
// a whole zstd-compressed record; false (err) for anything but one complete frame of a sane size
bool zstd_record(const unsigned char *src, size_t n, std::vector<unsigned char> &out, std::string &err) {
    const Zstd &z = zstd();
    const unsigned long long fs = z.frame_size(src, n);
    // (unknown / error content sizes are 2^64 - 1 and 2^64 - 2; slow5lib writes one-shot frames, which carry their size. A record is a read:
    // ids + 44 bytes + samples; a frame that claims more than 2^32 bytes or more than 2^17 times its compressed size is not one)
    if (fs == ~0ull) { // a frame without its content size (a writer that streamed the record): through the streaming decoder, buffer grown as needed
        void *ds = z.create_dstream();
        if (!ds || z.is_error(z.init_dstream(ds))) { if (ds) z.free_dstream(ds); err = "zstd init failed"; return false; }
        out.resize(n * 4 + 4096);
        Zstd::Buf in{src, n, 0}; Zstd::OBuf ob{out.data(), out.size(), 0};
        bool done = false, bad = false;
        while (!done && !bad) {
            const size_t before_in = in.pos, before_out = ob.pos;
            const size_t rc = z.decompress_stream(ds, &ob, &in);
            if (z.is_error(rc)) bad = true;
            else if (rc == 0) done = true;
            else if (ob.pos == ob.size) { if (out.size() >= (1ull << 32)) bad = true; else { out.resize(out.size() * 2); ob.dst = out.data(); ob.size = out.size(); } }
            else if (in.pos == before_in && ob.pos == before_out) bad = true; // no progress with room left: the frame is cut short
        }
        z.free_dstream(ds);
        if (bad) { err = "zstd error in BLOW5 record"; return false; }
        out.resize(ob.pos);
        return true;
    }
    if (fs >= (1ull << 32) || fs / (1u << 17) > (unsigned long long)n + 1) { err = "zstd error in BLOW5 record (implausible frame size)"; return false; }
    out.resize((size_t)fs ? (size_t)fs : 1);
    const size_t got = z.decompress(out.data(), out.size(), src, n);
    if (z.is_error(got) || got != (size_t)fs) { err = "zstd error in BLOW5 record"; return false; }
    out.resize(got);
    return true;
}
} // namespace

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

namespace pgh {

bool MappedFile::open(const std::string &path) {
    close();
    fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0) { close(); return false; }
    size = (size_t)st.st_size;
    if (size == 0) { data = ""; return true; }
    void *p = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (p == MAP_FAILED) { close(); return false; }
    data = (const char *)p;
    return true;
}
void MappedFile::close() {
    if (data && size) munmap((void *)data, size);
    if (fd >= 0) ::close(fd);
    data = nullptr; size = 0; fd = -1;
}

// ======================================================================================================
// SLOW5
// ======================================================================================================

static const char *next_tab(const char *p, const char *e) { const char *q = (const char *)memchr(p, '\t', (size_t)(e - p)); return q ? q : e; }

bool Slow5File::open(const std::string &path, std::string &err) {
    if (!f_.open(path)) { err = "cannot open " + path; return false; }
    binary_ = f_.size >= 6 && memcmp(f_.data, "BLOW5\1", 6) == 0;
    return binary_ ? index_blow5(err) : index_ascii(err);
}

bool Slow5File::index_ascii(std::string &err) {
    const char *p = f_.data, *e = f_.data + f_.size;
    while (p < e) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(e - p));
        const char *le = nl ? nl : e;
        if (le > p) {
            if (*p == '#') {
                if (le - p > 8 && memcmp(p, "#read_id", 8) == 0) { // column names: locate the primary fields
                    int col = 0; const char *c = p + 1;
                    while (c < le) {
                        const char *t = next_tab(c, le);
                        std::string name(c, t);
                        if (name == "digitisation") col_dig_ = col; else if (name == "offset") col_off_ = col;
                        else if (name == "range") col_range_ = col; else if (name == "len_raw_signal") col_len_ = col;
                        else if (name == "raw_signal") col_sig_ = col;
                        col++; c = t < le ? t + 1 : le;
                    }
                }
            } else if (*p != '@') {
                const char *t = next_tab(p, le);
                std::string id(p, t);
                if (!index_.emplace(id, Loc{(uint64_t)(p - f_.data), (uint64_t)(le - p)}).second) { err = "duplicate read id " + id; return false; }
                order_.push_back(id);
            }
        }
        p = nl ? nl + 1 : e;
    }
    return true;
}

// BLOW5 layout (slow5 specification 0.2.0 / 1.0.0): 64-byte file header = "BLOW5\1", version (3 bytes),
// record compression (0 none, 1 zlib, 2 zstd), num_read_groups (u32), signal compression (0 none, 1 svb-zd,
// 2 ex-zd), zero padding; u32 header length + ASCII header; records = u64 size + body; EOF marker "5WOLB".
bool Slow5File::index_blow5(std::string &err) {
    if (f_.size < 68) { err = "truncated BLOW5 header"; return false; }
    const unsigned char *d = (const unsigned char *)f_.data;
    rec_press_ = d[9];
    sig_press_ = d[14];
    if (rec_press_ > 2) { err = "BLOW5 record compression other than none/zlib/zstd is not supported"; return false; }
    if (rec_press_ == 2 && !zstd().ok) { err = "BLOW5 records are zstd-compressed and libzstd.so.1 was not found on this machine"; return false; }
    if (sig_press_ > 1) { err = "BLOW5 signal compression other than none/svb-zd is not supported"; return false; } // (ex-zd: slow5lib >= 1.1 only; the reference's pinned slow5lib writes svb-zd)
    uint32_t hlen; memcpy(&hlen, d + 64, 4);
    uint64_t pos = 68 + (uint64_t)hlen;
    if (pos > f_.size) { err = "truncated BLOW5 header"; return false; }
    bool eof_marker = false;
    z_stream zs; bool zs_ready = false; // one inflate state for all record heads (inflateReset is far cheaper than inflateInit)
    struct ZEnd { z_stream *z; bool *on; ~ZEnd() { if (*on) inflateEnd(z); } } zend{&zs, &zs_ready};
    for (;;) {
        if (f_.size - pos >= 5 && memcmp(d + pos, "5WOLB", 5) == 0) { eof_marker = true; break; }
        if (f_.size - pos < 8) break;
        uint64_t sz; memcpy(&sz, d + pos, 8);
        pos += 8;
        if (sz > f_.size - pos) { err = "truncated BLOW5 record"; return false; } // (not pos + sz: a size field of 2^64 - 1 must not wrap)
        if (sz < 2) { err = "corrupt BLOW5 record"; return false; }
        Loc l{pos, sz};
        // the read id sits at the start of the (possibly zlib-compressed) body
        std::string id;
        if (rec_press_ == 0) {
            uint16_t il; memcpy(&il, d + pos, 2);
            if ((uint64_t)il + 2 > sz) { err = "corrupt BLOW5 record"; return false; }
            id.assign((const char *)d + pos + 2, il);
        } else if (rec_press_ == 2) {
            // zstd: the first bytes of the record through the streaming decoder (one context for the whole file), a 256-byte window first
            const Zstd &z = zstd();
            static thread_local struct DS { void *p = nullptr; ~DS() { if (p) zstd().free_dstream(p); } } ds;
            if (!ds.p && !(ds.p = z.create_dstream())) { err = "zstd init failed"; return false; }
            unsigned char head[2 + 65536];
            uint16_t il = 0; size_t got = 0;
            for (size_t want : {(size_t)256, sizeof head}) {
                if (z.is_error(z.init_dstream(ds.p))) { err = "zstd init failed"; return false; }
                Zstd::Buf in{d + pos, (size_t)sz, 0}; Zstd::OBuf ob{head, want, 0};
                while (ob.pos < ob.size && in.pos < in.size) {
                    const size_t before_in = in.pos, before_out = ob.pos;
                    const size_t rc = z.decompress_stream(ds.p, &ob, &in);
                    if (z.is_error(rc)) { err = "zstd error in BLOW5 record"; return false; }
                    if (rc == 0 || (in.pos == before_in && ob.pos == before_out)) break; // frame complete / no progress
                }
                got = ob.pos;
                if (got < 2) { err = "zstd error in BLOW5 record"; return false; }
                memcpy(&il, head, 2);
                if ((size_t)il + 2 <= got) break;
            }
            if ((size_t)il + 2 > got) { err = "corrupt BLOW5 record"; return false; }
            id.assign((const char *)head + 2, il);
        } else {
            if (sz > 0xFFFFFFFFull) { err = "BLOW5 record of 4 GiB or more (zlib)"; return false; } // zlib counts input in 32 bits: never hand it a truncated size
            // only the first bytes of the record are inflated: u16 id length + id (a first try of 256 bytes covers every
            // real read id; longer ones get a second, full-size try) -- inflating whole records made indexing a 50 000-read
            // file take a second
            unsigned char head[2 + 65536];
            uint16_t il = 0; size_t got = 0;
            for (size_t want : {(size_t)256, sizeof head}) {
                if (!zs_ready) { memset(&zs, 0, sizeof zs); if (inflateInit(&zs) != Z_OK) { err = "zlib init failed"; return false; } zs_ready = true; }
                else if (inflateReset(&zs) != Z_OK) { err = "zlib reset failed"; return false; }
                zs.next_in = (Bytef *)(d + pos); zs.avail_in = (uInt)sz; zs.next_out = head; zs.avail_out = (uInt)want;
                const int rc = inflate(&zs, Z_SYNC_FLUSH);
                got = want - zs.avail_out;
                if ((rc != Z_OK && rc != Z_STREAM_END && rc != Z_BUF_ERROR) || got < 2) { err = "zlib error in BLOW5 record"; return false; }
                memcpy(&il, head, 2);
                if ((size_t)il + 2 <= got) break;
            }
            if ((size_t)il + 2 > got) { err = "corrupt BLOW5 record"; return false; }
            id.assign((const char *)head + 2, il);
        }
        if (!id.empty() && id.back() == '\0') id.pop_back();
        if (!index_.emplace(id, l).second) { err = "duplicate read id " + id; return false; }
        order_.push_back(id);
        pos += sz;
    }
    if (!eof_marker) { err = "truncated BLOW5 file (no end-of-file marker)"; return false; } // slow5 specification: every BLOW5 file ends in "5WOLB"
    return true;
}

// streamvbyte (Lemire) decode of `n` uint32 values: ceil(n/4) control bytes, then 1-4 data bytes per value
static bool svb_decode(const unsigned char *in, size_t in_len, uint32_t n, std::vector<uint32_t> &out) {
    const size_t nctrl = ((size_t)n + 3) / 4;
    if (in_len < nctrl || (size_t)n > in_len - nctrl) return false; // every value has a control field and at least one data byte: checked BEFORE anything is sized by n
    const unsigned char *ctrl = in, *dp = in + nctrl, *end = in + in_len;
    out.resize(n);
    for (uint32_t i = 0; i < n; i++) {
        const unsigned code = (ctrl[i >> 2] >> ((i & 3) * 2)) & 3u;
        if (dp + code + 1 > end) return false;
        uint32_t v = 0;
        for (unsigned b = 0; b <= code; b++) v |= (uint32_t)dp[b] << (8 * b);
        dp += code + 1;
        out[i] = v;
    }
    return true;
}

bool Slow5File::decode_blow5(const Loc &l, Slow5Rec &out, std::string &err) const {
    const unsigned char *body = (const unsigned char *)f_.data + l.off;
    size_t blen = l.len;
    std::vector<unsigned char> inflated;
    if (rec_press_ == 2) {
        if (!zstd_record(body, blen, inflated, err)) return false;
        body = inflated.data(); blen = inflated.size();
    } else if (rec_press_ == 1) {
        if (blen > 0xFFFFFFFFull) { err = "BLOW5 record of 4 GiB or more (zlib)"; return false; } // (uLong / the products below: bounded before anything is sized by it)
        size_t cap = blen * 4 + 1024;
        for (;;) {
            inflated.resize(cap);
            uLongf dl = (uLongf)cap;
            int rc = uncompress(inflated.data(), &dl, body, (uLong)blen);
            if (rc == Z_OK) { inflated.resize(dl); break; }
            if (rc != Z_BUF_ERROR) { err = "zlib error in BLOW5 record"; return false; }
            // Z_BUF_ERROR is "output too small" AND "the stream ends before its end marker": deflate expands by at most 1032 : 1, so a
            // buffer beyond that says the record is cut short -- not a reason to double the buffer until memory runs out
            if (cap > blen * 1100 + (1u << 20)) { err = "zlib error in BLOW5 record (incomplete stream)"; return false; }
            cap *= 2;
        }
        body = inflated.data(); blen = inflated.size();
    }
    size_t p = 0;
    auto need = [&](size_t n) { return p + n <= blen; };
    uint16_t il;
    if (!need(2)) { err = "corrupt BLOW5 record"; return false; }
    memcpy(&il, body, 2); p = 2 + il;
    if (!need(4 + 32 + 8)) { err = "corrupt BLOW5 record"; return false; }
    p += 4; // read_group
    double sampling;
    memcpy(&out.digitisation, body + p, 8); memcpy(&out.offset, body + p + 8, 8); memcpy(&out.range, body + p + 16, 8);
    memcpy(&sampling, body + p + 24, 8); p += 32;
    uint64_t len; memcpy(&len, body + p, 8); p += 8;
    if (sig_press_ == 0) {
        if (len > (blen - p) / 2) { err = "corrupt BLOW5 record (signal)"; return false; } // (before the vector is sized by a number the file supplied)
        out.raw.resize(len);
        memcpy(out.raw.data(), body + p, len * 2);
        return true;
    }
    // svb-zd: in the record len_raw_signal holds the BYTE length of the compressed signal; the compressed
    // block is u32 count + streamvbyte of zig-zag deltas of the int16 samples widened to int32
    const uint64_t clen = len;
    if (clen > blen - p || clen < 4) { err = "corrupt BLOW5 record (svb-zd)"; return false; }
    uint32_t count; memcpy(&count, body + p, 4);
    std::vector<uint32_t> zz;
    if (!svb_decode(body + p + 4, clen - 4, count, zz)) { err = "corrupt streamvbyte block"; return false; }
    out.raw.resize(count);
    uint32_t prev = 0; // accumulated modulo 2^32 (crafted deltas must not be a signed overflow: UB, and an abort under -fsanitize=undefined)
    for (uint32_t i = 0; i < count; i++) {
        const uint32_t delta = (zz[i] >> 1) ^ (0u - (zz[i] & 1u)); // zig-zag
        prev += delta;
        out.raw[i] = (int16_t)(uint16_t)prev;
    }
    return true;
}

bool Slow5File::raw_view(const std::string &read_id, RawView &v, std::string &err) const {
    auto it = index_.find(read_id);
    if (it == index_.end()) { err = "read " + read_id + " not found"; return false; }
    if (!has_raw_views()) { err = "BLOW5 records are compressed"; return false; }
    const unsigned char *body = (const unsigned char *)f_.data + it->second.off;
    const size_t blen = it->second.len;
    uint16_t il;
    if (blen < 2) { err = "corrupt BLOW5 record"; return false; }
    memcpy(&il, body, 2);
    size_t p = 2 + (size_t)il;
    if (p + 4 + 32 + 8 > blen) { err = "corrupt BLOW5 record"; return false; }
    p += 4; // read_group
    memcpy(&v.digitisation, body + p, 8); memcpy(&v.offset, body + p + 8, 8); memcpy(&v.range, body + p + 16, 8); p += 32;
    memcpy(&v.n, body + p, 8); p += 8;
    if (v.n > (blen - p) / 2) { err = "corrupt BLOW5 record (signal)"; return false; }
    v.samples = body + p; // (2-byte alignment is not guaranteed: copy with memcpy)
    return true;
}

bool Slow5File::get(const std::string &read_id, Slow5Rec &out, std::string &err) const {
    auto it = index_.find(read_id);
    if (it == index_.end()) { err = "read " + read_id + " not found"; return false; }
    if (binary_) return decode_blow5(it->second, out, err);
    const char *p = f_.data + it->second.off, *e = p + it->second.len;
    int col = 0;
    uint64_t len = 0; bool have_len = false;
    const char *sig_b = nullptr, *sig_e = nullptr;
    while (p <= e) {
        const char *t = next_tab(p, e);
        if (col == col_dig_) out.digitisation = strtod(std::string(p, t).c_str(), nullptr);
        else if (col == col_off_) out.offset = strtod(std::string(p, t).c_str(), nullptr);
        else if (col == col_range_) out.range = strtod(std::string(p, t).c_str(), nullptr);
        else if (col == col_len_) { len = strtoull(std::string(p, t).c_str(), nullptr, 10); have_len = true; }
        else if (col == col_sig_) { sig_b = p; sig_e = t; }
        col++;
        if (t >= e) break;
        p = t + 1;
    }
    if (!have_len || !sig_b) { err = "malformed SLOW5 record for " + read_id; return false; }
    out.raw.resize(len);
    const char *c = sig_b;
    for (uint64_t i = 0; i < len; i++) {
        if (c >= sig_e) { err = "raw_signal shorter than len_raw_signal for " + read_id; return false; }
        bool neg = false; if (*c == '-') { neg = true; c++; }
        int v = 0; while (c < sig_e && *c >= '0' && *c <= '9') { v = v * 10 + (*c - '0'); c++; }
        out.raw[i] = (int16_t)(neg ? -v : v);
        if (c < sig_e && *c == ',') c++;
    }
    return true;
}

// ======================================================================================================
// FASTA / FASTQ index
// ======================================================================================================

bool FastxIndex::load(const std::string &path, std::string &err) {
    if (!f_.open(path)) { err = "cannot open " + path; return false; }
    const char *p = f_.data, *e = f_.data + f_.size;
    // Scan the file (a .fai next to it, if any, is not needed: the scan yields the same offsets). Handles
    // multi-line FASTA and 4-line / multi-line FASTQ; the name is the first word after '>' / '@'.
    while (p < e) {
        if (*p != '>' && *p != '@') { const char *nl = (const char *)memchr(p, '\n', (size_t)(e - p)); p = nl ? nl + 1 : e; continue; }
        const bool fq = *p == '@';
        const char *nl = (const char *)memchr(p, '\n', (size_t)(e - p));
        const char *le = nl ? nl : e;
        const char *ne = p + 1; while (ne < le && !isspace((unsigned char)*ne)) ne++;
        std::string name(p + 1, ne);
        p = nl ? nl + 1 : e;
        Ent ent{(uint64_t)(p - f_.data), 0, 0, 0};
        bool first = true;
        while (p < e && *p != '>' && !(fq && *p == '+')) {
            const char *n2 = (const char *)memchr(p, '\n', (size_t)(e - p));
            const char *l2 = n2 ? n2 : e;
            size_t bases = (size_t)(l2 - p); if (bases && p[bases - 1] == '\r') bases--;
            if (first) { ent.line_bases = (uint32_t)bases; ent.line_width = (uint32_t)((n2 ? n2 + 1 : e) - p); first = false; }
            ent.len += (int64_t)bases;
            p = n2 ? n2 + 1 : e;
            if (!fq && p < e && *p == '>') break;
        }
        if (fq && p < e && *p == '+') { // skip the separator line and as many quality characters as bases
            const char *n2 = (const char *)memchr(p, '\n', (size_t)(e - p)); p = n2 ? n2 + 1 : e;
            int64_t left = ent.len;
            while (p < e && left > 0) {
                const char *n3 = (const char *)memchr(p, '\n', (size_t)(e - p));
                const char *l3 = n3 ? n3 : e;
                size_t q = (size_t)(l3 - p); if (q && p[q - 1] == '\r') q--;
                left -= (int64_t)q; p = n3 ? n3 + 1 : e;
            }
        }
        idx_.emplace(name, ent); // like htslib, the first record of a name wins
    }
    return true;
}

bool FastxIndex::fetch(const std::string &name, int64_t beg, int64_t end, std::string &out) const {
    out.clear();
    auto it = idx_.find(name);
    if (it == idx_.end()) return false;
    const Ent &en = it->second;
    // faidx_adjust_position (htslib 1.17 faidx.c)
    if (end < beg) beg = end;
    if (beg < 0) beg = 0; else if (en.len <= beg) beg = en.len;
    if (end < 0) end = 0; else if (en.len <= end) end = en.len - 1;
    const int64_t n = end + 1 - beg;
    if (n <= 0 || en.line_bases == 0) return true;
    out.reserve((size_t)n);
    const uint64_t off = en.seq_off + (uint64_t)(beg / en.line_bases) * en.line_width + (uint64_t)(beg % en.line_bases);
    const char *p = f_.data + off, *e = f_.data + f_.size;
    while ((int64_t)out.size() < n && p < e) { if (isgraph((unsigned char)*p)) out.push_back(*p); p++; }
    return true;
}

// ======================================================================================================
// PAF
// ======================================================================================================

int parse_paf_line(char *line, size_t len, PafRec &out) {
    // columns are split on \t \r \n like strtok in the reference (empty fields collapse)
    char *col[12]; int nc = 0;
    char *p = line, *e = line + len;
    out.ss = nullptr; out.ss_len = 0;
    auto is_sep = [](char c) { return c == '\t' || c == '\r' || c == '\n'; };
    while (p < e) {
        while (p < e && is_sep(*p)) p++;
        if (p >= e) break;
        char *t = p; while (t < e && !is_sep(*t)) t++;
        if (nc < 12) col[nc++] = p;
        else if (t - p >= 5 && memcmp(p, "ss:Z:", 5) == 0) { out.ss = p + 5; out.ss_len = (size_t)(t - p - 5); } // the last ss tag wins
        if (t < e) *t = '\0';
        p = t + 1;
    }
    if (nc < 12) return 1;
    out.rid = col[0]; out.qlen = atoi(col[1]); out.query_start = atoi(col[2]); out.query_end = atoi(col[3]);
    out.tid = col[5]; out.tlen = atoi(col[6]); out.target_start = atoi(col[7]); out.target_end = atoi(col[8]);
    if (!out.ss) return 2;
    return 0;
}

bool tokenize_ss(const char *ss, size_t len, std::vector<uint32_t> &op_n, std::vector<uint8_t> &op_t, std::string &err) {
    uint64_t num = 0; int digits = 0;
    for (size_t i = 0; i < len; i++) {
        const char c = ss[i];
        if (c == ',' || c == 'I' || c == 'D') {
            if (digits <= 0) { err = "Bad ss: Preceding digit missing"; return false; }
            if (num > 0x7fffffffull) { err = "Bad ss: Cannot have negative numbers"; return false; } // atoi wraps negative there
            op_n.push_back((uint32_t)num);
            op_t.push_back(c == ',' ? 0 : (c == 'I' ? 1 : 2));
            num = 0; digits = 0;
        } else {
            if (c < '0' || c > '9') { err = "Bad ss: A non-digit found when expected a digit"; return false; }
            if (digits >= 10) { err = "Bad ss: number with more than 10 digits"; return false; } // buff[11] in the reference
            num = num * 10 + (uint64_t)(c - '0'); digits++;
        }
    }
    return true;
}

// ======================================================================================================
// SAM / BAM
// ======================================================================================================

static char seq_letter(char c) { // gmove.cpp:1128-1134: alphabet "NACNGNNNT" over htslib's 4-bit codes
    if (c >= 'a' && c <= 'z') c = (char)(c - 32);
    return (c == 'A' || c == 'C' || c == 'G' || c == 'T') ? c : 'N';
}

bool SamBamReader::open(const std::string &path, std::string &err) {
    if (!f_.open(path)) { err = "cannot open " + path; return false; }
    bam_ = f_.size >= 4 && (unsigned char)f_.data[0] == 0x1f && (unsigned char)f_.data[1] == 0x8b;
    pos_ = 0; buf_.clear(); bpos_ = 0;
    if (!bam_) return true;
    // BAM header: magic, l_text, text, n_ref, references
    if (!fill(12, err)) { if (err.empty()) err = "truncated BAM header"; return false; }
    if (memcmp(buf_.data(), "BAM\1", 4) != 0) { err = "not a BAM file"; return false; }
    int32_t l_text; memcpy(&l_text, buf_.data() + 4, 4);
    if (l_text < 0 || !fill(12 + (size_t)l_text, err)) { if (err.empty()) err = "truncated BAM header"; return false; }
    bpos_ = 8 + (size_t)l_text;
    int32_t n_ref; memcpy(&n_ref, buf_.data() + bpos_, 4); bpos_ += 4;
    if (n_ref < 0) { err = "corrupt BAM header"; return false; }
    for (int32_t i = 0; i < n_ref; i++) {
        if (!fill(bpos_ + 4, err)) { if (err.empty()) err = "truncated BAM header"; return false; }
        int32_t l_name; memcpy(&l_name, buf_.data() + bpos_, 4);
        if (l_name < 0 || !fill(bpos_ + 8 + (size_t)l_name, err)) { if (err.empty()) err = "truncated BAM header"; return false; }
        bpos_ += 4 + (size_t)l_name + 4;
    }
    return true;
}

// make at least `need` inflated bytes available in buf_ (from offset 0); false at end of data
bool SamBamReader::fill(size_t need, std::string &err) {
    while (buf_.size() < need) {
        if (pos_ + 18 > f_.size) return false;
        const unsigned char *b = (const unsigned char *)f_.data + pos_;
        if (b[0] != 0x1f || b[1] != 0x8b || !(b[3] & 4)) { err = "corrupt BGZF block"; return false; }
        uint16_t xlen; memcpy(&xlen, b + 10, 2);
        if (12u + (size_t)xlen > f_.size - pos_) { err = "corrupt BGZF block"; return false; } // the extra field itself must lie inside the file
        uint32_t bsize = 0; bool found = false;
        for (size_t o = 12; o + 4 <= 12u + xlen;) { // extra subfields: SI1 SI2 SLEN data
            uint16_t slen; memcpy(&slen, b + o + 2, 2);
            if (b[o] == 'B' && b[o + 1] == 'C' && slen == 2 && o + 6 <= 12u + xlen) { uint16_t v; memcpy(&v, b + o + 4, 2); bsize = (uint32_t)v + 1; found = true; }
            o += 4u + slen;
        }
        const size_t hdr = 12u + xlen;
        if (!found || bsize > f_.size - pos_ || bsize < hdr + 8) { err = "corrupt BGZF block"; return false; }
        const size_t clen = bsize - hdr - 8;
        uint32_t isize; memcpy(&isize, b + bsize - 4, 4);
        if (isize > 65536u) { err = "corrupt BGZF block"; return false; } // (SAM specification 4.1: a block inflates to at most 64 KiB)
        const size_t old = buf_.size();
        buf_.resize(old + isize);
        if (isize) {
            z_stream zs; memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, -15) != Z_OK) { err = "zlib init failed"; return false; }
            zs.next_in = (Bytef *)(b + hdr); zs.avail_in = (uInt)clen; zs.next_out = buf_.data() + old; zs.avail_out = isize;
            const int rc = inflate(&zs, Z_FINISH);
            inflateEnd(&zs);
            if (rc != Z_STREAM_END) { err = "zlib error in BGZF block"; return false; }
        }
        pos_ += bsize;
    }
    return true;
}

int SamBamReader::next(MoveRec &out, std::string &err) {
    out = MoveRec();
    if (!bam_) {
        const char *e = f_.data + f_.size;
        while (pos_ < f_.size) {
            const char *p = f_.data + pos_;
            const char *nl = (const char *)memchr(p, '\n', (size_t)(e - p));
            const char *le = nl ? nl : e;
            pos_ = (size_t)((nl ? nl + 1 : e) - f_.data);
            if (le == p || *p == '@') continue; // header / blank line
            int col = 0;
            while (p <= le) {
                const char *t = (const char *)memchr(p, '\t', (size_t)(le - p)); if (!t) t = le;
                if (col == 0) out.qname.assign(p, t);
                else if (col == 9) { out.seq.assign(p, t); if (out.seq == "*") out.seq.clear(); for (auto &ch : out.seq) ch = seq_letter(ch); }
                else if (col >= 11 && t - p >= 5) {
                    if (memcmp(p, "ns:i:", 5) == 0) { out.ns = strtoull(std::string(p + 5, t).c_str(), nullptr, 10); out.has_ns = true; }
                    else if (memcmp(p, "ts:i:", 5) == 0) { out.ts = strtoull(std::string(p + 5, t).c_str(), nullptr, 10); out.has_ts = true; }
                    else if (memcmp(p, "mv:", 3) == 0) {
                        out.has_mv = true;
                        out.mv_is_Bc = t - p >= 7 && memcmp(p, "mv:B:c", 6) == 0;
                        if (out.mv_is_Bc) {
                            const char *q = p + 6; bool first = true;
                            while (q < t) {
                                if (*q == ',') q++;
                                char *endp; const long v = strtol(q, &endp, 10);
                                if (endp == q) break;
                                if (first) { out.stride = (int)v; first = false; } else out.is_one.push_back(v == 1);
                                out.mv_len++;
                                q = endp;
                            }
                        }
                    }
                }
                col++;
                if (t >= le) break;
                p = t + 1;
            }
            if (col < 11) { err = "malformed SAM record"; return -1; }
            return 1;
        }
        return 0;
    }
    // BAM record
    // drop consumed bytes now and then
    if (bpos_ > (1u << 20)) { buf_.erase(buf_.begin(), buf_.begin() + (long)bpos_); bpos_ = 0; }
    if (!fill(bpos_ + 4, err)) return err.empty() ? 0 : -1;
    int32_t block_size; memcpy(&block_size, buf_.data() + bpos_, 4);
    if (block_size < 32 || !fill(bpos_ + 4 + (size_t)block_size, err)) { if (err.empty()) err = "truncated BAM record"; return -1; }
    const unsigned char *r = buf_.data() + bpos_ + 4, *rend = r + block_size;
    bpos_ += 4 + (size_t)block_size;
    const uint8_t l_read_name = r[8];
    uint16_t n_cigar; memcpy(&n_cigar, r + 12, 2);
    int32_t l_seq; memcpy(&l_seq, r + 16, 4);
    // every length below comes from the file: compared as sizes against what is left of the record, never added to a pointer first
    const unsigned char *p = r + 32;
    auto left = [&]() { return (size_t)(rend - p); };
    if ((size_t)l_read_name > left()) { err = "corrupt BAM record"; return -1; }
    out.qname.assign((const char *)p, l_read_name ? l_read_name - 1 : 0); p += l_read_name;
    if (4u * (size_t)n_cigar > left()) { err = "corrupt BAM record"; return -1; }
    p += 4u * (size_t)n_cigar;
    if (l_seq < 0 || ((size_t)l_seq + 1) / 2 + (size_t)l_seq > left()) { err = "corrupt BAM record"; return -1; }
    static const char code[] = "=ACMGRSVTWYHKDBN";
    out.seq.resize((size_t)l_seq);
    for (int32_t i = 0; i < l_seq; i++) out.seq[(size_t)i] = seq_letter(code[(p[i >> 1] >> ((~i & 1) << 2)) & 15]);
    p += ((size_t)l_seq + 1) / 2 + (size_t)l_seq;
    while (left() >= 3) { // tags
        const char t0 = (char)p[0], t1 = (char)p[1], ty = (char)p[2];
        p += 3;
        int64_t iv = 0; bool is_int = false; size_t adv = 0;
        const size_t fixed = (ty == 'A' || ty == 'c' || ty == 'C') ? 1 : (ty == 's' || ty == 'S') ? 2 : (ty == 'i' || ty == 'I' || ty == 'f') ? 4 : 0;
        if (fixed > left()) { err = "corrupt BAM tag"; return -1; } // the value is read below: it must lie inside the record
        switch (ty) {
            case 'A': adv = 1; break;
            case 'c': iv = (int8_t)p[0]; is_int = true; adv = 1; break;
            case 'C': iv = p[0]; is_int = true; adv = 1; break;
            case 's': { int16_t v; memcpy(&v, p, 2); iv = v; is_int = true; adv = 2; break; }
            case 'S': { uint16_t v; memcpy(&v, p, 2); iv = v; is_int = true; adv = 2; break; }
            case 'i': { int32_t v; memcpy(&v, p, 4); iv = v; is_int = true; adv = 4; break; }
            case 'I': { uint32_t v; memcpy(&v, p, 4); iv = v; is_int = true; adv = 4; break; }
            case 'f': adv = 4; break;
            case 'Z': case 'H': { const unsigned char *z = (const unsigned char *)memchr(p, 0, left()); if (!z) { err = "corrupt BAM tag"; return -1; } adv = (size_t)(z - p) + 1; break; }
            case 'B': {
                if (left() < 5) { err = "corrupt BAM tag"; return -1; }
                const char sub = (char)p[0]; int32_t cnt; memcpy(&cnt, p + 1, 4);
                const size_t esz = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
                if (cnt < 0 || (size_t)cnt > (left() - 5) / esz) { err = "corrupt BAM tag"; return -1; }
                if (t0 == 'm' && t1 == 'v') {
                    out.has_mv = true; out.mv_is_Bc = sub == 'c'; out.mv_len = (uint32_t)cnt;
                    if (out.mv_is_Bc && cnt > 0) {
                        out.stride = (int8_t)p[5];
                        out.is_one.resize((size_t)cnt - 1);
                        for (int32_t i = 1; i < cnt; i++) out.is_one[(size_t)i - 1] = (int8_t)p[5 + i] == 1;
                    }
                }
                adv = 5 + esz * (size_t)cnt; break;
            }
            default: err = "unknown BAM tag type"; return -1;
        }
        if (adv > left()) { err = "corrupt BAM tag"; return -1; }
        if (is_int && t0 == 'n' && t1 == 's') { out.ns = (uint64_t)iv; out.has_ns = true; }
        if (is_int && t0 == 't' && t1 == 's') { out.ts = (uint64_t)iv; out.has_ts = true; }
        p += adv;
    }
    return 1;
}

} // namespace pgh
