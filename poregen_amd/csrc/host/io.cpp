// io.cpp -- memory-mapped files, ASCII SLOW5 / BLOW5 reader, FASTA/FASTQ index, PAF + ss tokeniser.
#include "pg_host.h"

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
    if (rec_press_ > 1) { err = "BLOW5 record compression other than none/zlib is not supported"; return false; }
    if (sig_press_ > 1) { err = "BLOW5 signal compression other than none/svb-zd is not supported"; return false; }
    uint32_t hlen; memcpy(&hlen, d + 64, 4);
    uint64_t pos = 68 + (uint64_t)hlen;
    while (pos + 8 <= f_.size) {
        if (f_.size - pos >= 5 && memcmp(d + pos, "5WOLB", 5) == 0) break;
        uint64_t sz; memcpy(&sz, d + pos, 8);
        pos += 8;
        if (pos + sz > f_.size) { err = "truncated BLOW5 record"; return false; }
        Loc l{pos, sz};
        // the read id sits at the start of the (possibly zlib-compressed) body
        std::string id;
        if (rec_press_ == 0) {
            uint16_t il; memcpy(&il, d + pos, 2);
            id.assign((const char *)d + pos + 2, il);
        } else {
            unsigned char head[2 + 65536];
            z_stream zs; memset(&zs, 0, sizeof zs);
            if (inflateInit(&zs) != Z_OK) { err = "zlib init failed"; return false; }
            zs.next_in = (Bytef *)(d + pos); zs.avail_in = (uInt)sz; zs.next_out = head; zs.avail_out = sizeof head;
            int rc = inflate(&zs, Z_SYNC_FLUSH);
            size_t got = sizeof head - zs.avail_out;
            inflateEnd(&zs);
            if ((rc != Z_OK && rc != Z_STREAM_END) || got < 2) { err = "zlib error in BLOW5 record"; return false; }
            uint16_t il; memcpy(&il, head, 2);
            if ((size_t)il + 2 > got) { err = "corrupt BLOW5 record"; return false; }
            id.assign((const char *)head + 2, il);
        }
        if (!id.empty() && id.back() == '\0') id.pop_back();
        if (!index_.emplace(id, l).second) { err = "duplicate read id " + id; return false; }
        order_.push_back(id);
        pos += sz;
    }
    return true;
}

// streamvbyte (Lemire) decode of `n` uint32 values: ceil(n/4) control bytes, then 1-4 data bytes per value
static bool svb_decode(const unsigned char *in, size_t in_len, uint32_t n, std::vector<uint32_t> &out) {
    const size_t nctrl = ((size_t)n + 3) / 4;
    if (in_len < nctrl) return false;
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
    if (rec_press_ == 1) {
        size_t cap = blen * 4 + 1024;
        for (;;) {
            inflated.resize(cap);
            uLongf dl = (uLongf)cap;
            int rc = uncompress(inflated.data(), &dl, body, (uLong)blen);
            if (rc == Z_OK) { inflated.resize(dl); break; }
            if (rc != Z_BUF_ERROR) { err = "zlib error in BLOW5 record"; return false; }
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
    out.raw.resize(len);
    if (sig_press_ == 0) {
        if (!need(len * 2)) { err = "corrupt BLOW5 record (signal)"; return false; }
        memcpy(out.raw.data(), body + p, len * 2);
        return true;
    }
    // svb-zd: in the record len_raw_signal holds the BYTE length of the compressed signal; the compressed
    // block is u32 count + streamvbyte of zig-zag deltas of the int16 samples widened to int32
    const uint64_t clen = len;
    if (!need(clen) || clen < 4) { err = "corrupt BLOW5 record (svb-zd)"; return false; }
    uint32_t count; memcpy(&count, body + p, 4);
    std::vector<uint32_t> zz;
    if (!svb_decode(body + p + 4, clen - 4, count, zz)) { err = "corrupt streamvbyte block"; return false; }
    out.raw.resize(count);
    int32_t prev = 0;
    for (uint32_t i = 0; i < count; i++) {
        const int32_t delta = (int32_t)(zz[i] >> 1) ^ -(int32_t)(zz[i] & 1);
        prev += delta;
        out.raw[i] = (int16_t)prev;
    }
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

} // namespace pgh
