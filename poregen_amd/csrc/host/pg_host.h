// pg_host.h -- host side of the `poregen gmove` drop-in: file parsers, k-mer list, dump writer.
// Everything here is I/O and bookkeeping; the computation is behind include/pgmove.h.
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <string>
#include <unordered_map>
#include <vector>

namespace pgh {

// ---- memory-mapped read-only file ---------------------------------------------------------------------
struct MappedFile {
    const char *data = nullptr;
    size_t size = 0;
    int fd = -1;
    bool open(const std::string &path);
    void close();
    ~MappedFile() { close(); }
};

// ---- SLOW5 / BLOW5 (replaces slow5_open / slow5_idx_load / slow5_get, src/gmove.cpp:493-503,745) --------
struct Slow5Rec {
    double digitisation = 0, offset = 0, range = 0;
    std::vector<int16_t> raw;
};
class Slow5File {
public:
    bool open(const std::string &path, std::string &err); // opens and indexes (read id -> record location)
    bool get(const std::string &read_id, Slow5Rec &out, std::string &err) const; // false if absent / malformed
    // uncompressed BLOW5 (records and signals stored as they are): the samples of a read where they lie in the mapped file, so that a
    // caller can copy them ONCE, straight to where they are needed. false if absent / malformed (err) -- only when has_raw_views()
    struct RawView { const void *samples = nullptr; uint64_t n = 0; double digitisation = 0, offset = 0, range = 0; };
    bool has_raw_views() const { return binary_ && rec_press_ == 0 && sig_press_ == 0; }
    bool raw_view(const std::string &read_id, RawView &v, std::string &err) const;
    size_t n_reads() const { return index_.size(); }
    bool is_binary() const { return binary_; }
    const std::vector<std::string> &ids_in_file_order() const { return order_; }
private:
    struct Loc { uint64_t off, len; };
    MappedFile f_;
    bool binary_ = false;
    uint8_t rec_press_ = 0, sig_press_ = 0;
    int col_dig_ = 2, col_off_ = 3, col_range_ = 4, col_len_ = 6, col_sig_ = 7;
    std::unordered_map<std::string, Loc> index_;
    std::vector<std::string> order_;
    bool index_ascii(std::string &err);
    bool index_blow5(std::string &err);
    bool decode_blow5(const Loc &l, Slow5Rec &out, std::string &err) const;
};

// ---- FASTA/FASTQ with faidx semantics (replaces fai_load / faidx_fetch_seq, src/gmove.cpp:724,805) -----
class FastxIndex {
public:
    bool load(const std::string &path, std::string &err);
    // htslib 1.17 faidx_fetch_seq(fai, name, beg, end, &len): [beg,end] 0-based inclusive, clamped into the
    // sequence; returns false (len = -2 in htslib) when the name is absent
    bool fetch(const std::string &name, int64_t beg, int64_t end, std::string &out) const;
private:
    struct Ent { uint64_t seq_off; int64_t len; uint32_t line_bases, line_width; };
    MappedFile f_;
    std::unordered_map<std::string, Ent> idx_;
};

// ---- PAF with ss:Z: (replaces parse_paf_rec, src/gmove.cpp:977-1052) -----------------------------------
struct PafRec {
    std::string rid, tid;
    int32_t qlen = 0, query_start = 0, query_end = 0, tlen = 0, target_start = 0, target_end = 0;
    const char *ss = nullptr; // points into the line buffer
    size_t ss_len = 0;
};
// returns 0 ok, 1 = fewer than 12 columns (the reference asserts), 2 = no ss:Z: tag (the reference exits 1)
int parse_paf_line(char *line, size_t len, PafRec &out);
// tokenises "<n>," "<n>I" "<n>D" (src/gmove.cpp:831-871); returns false on the reference's "Bad ss" exits
bool tokenize_ss(const char *ss, size_t len, std::vector<uint32_t> &op_n, std::vector<uint8_t> &op_t, std::string &err);

// ---- SAM / BAM records with the move table tags (replaces sam_open/sam_read1/bam_aux_get, src/gmove.cpp:1067-1134) --
struct MoveRec {
    std::string qname, seq;        // SEQ with every letter outside ACGT mapped to 'N' (gmove.cpp:1128-1134)
    int stride = 0;                // mv[0]
    uint32_t mv_len = 0;           // bam_auxB_len of mv (stride element included)
    std::vector<uint8_t> is_one;   // mv[1..]: 1 where the value is 1
    uint64_t ns = 0, ts = 0;       // signal length / trim offset tags
    bool has_ns = false, has_ts = false, has_mv = false, mv_is_Bc = false;
};
class SamBamReader {
public:
    bool open(const std::string &path, std::string &err); // SAM text or BGZF-compressed BAM, detected by content
    int next(MoveRec &out, std::string &err);              // 1 = record, 0 = end of file, -1 = error
private:
    MappedFile f_;
    bool bam_ = false;
    size_t pos_ = 0;                 // SAM: offset into the file; BAM: offset of the next BGZF block
    std::vector<unsigned char> buf_; // BAM: inflated bytes not yet consumed
    size_t bpos_ = 0;
    bool fill(size_t need, std::string &err);
};

// ---- k-mer list (src/poregen.cpp:248-267, src/gmove.cpp:394-426) ----------------------------------------
void generate_kmers(int k, bool rna, std::vector<std::string> &out);
// returns 0 ok, 1 cannot open, 2 a line does not have exactly k characters + '\n'
int read_kmer_file(const std::string &path, int k, std::vector<std::string> &out, std::string &err);

// ---- dump directory (src/gmove.cpp:107-140, 460-473, 525-534, 938-950, 196-203) -------------------------
int create_dir(const char *dir_name); // 0 ok / created, -1 exists and not empty, -2 cannot create
// "%.8f" exactly as glibc printf prints it; returns the number of characters written (buf >= 400 bytes)
size_t format_f8(double v, char *buf);
struct DumpInput {
    uint32_t n_slots;
    const uint64_t *counts, *ev_off, *samp_off;
    const uint32_t *ev_len, *ev_read;
    const double *samples;   // the job's k-mer-major sample stream on the host, or null: then `fetch` brings ranges of it (pg_fetch_samples)
    const uint8_t *read_skipped;
    uint64_t n_reads;
    std::function<bool(uint64_t first, uint64_t n, double *dst)> fetch; // callable from several threads at once
};
// Plain storage for the big host buffers (a batch's signal, the pieces the dump writers fetch): a mapping of its own with transparent huge
// pages asked for (the boxes run THP in "madvise" mode). 200 page faults instead of 100 000 per 400 MB at the first touch, the runtime pins
// 200 pages instead of 100 000 when it copies from / into it (staging a 176 MB batch: 35 -> 14 ms; tools/probe/e2e_staging.sh), and as many
// pages less to hand back when the process ends. Not initialised, not copyable.
struct HugeBuf {
    void *p = nullptr; size_t bytes = 0;
    HugeBuf() = default;
    HugeBuf(const HugeBuf &) = delete; HugeBuf &operator=(const HugeBuf &) = delete;
    ~HugeBuf() { release(); }
    void release();
    bool grow(size_t want_bytes, size_t keep_bytes); // false: out of memory; the first keep_bytes survive
};

// writes dump/<kmer> for every slot and freq.txt; delimit = -d; returns false on I/O error
bool write_dump_dir(const std::string &out_dir, const std::vector<std::string> &slot_kmers, const DumpInput &in, bool delimit,
                    uint32_t sample_limit, unsigned n_threads, std::string &err);
// the same from text the device has produced (pg_text): slot s is bytes [slot_off[s], slot_off[s + 1]) of one stream, fetched in pieces
struct TextInput {
    uint32_t n_slots;
    const uint64_t *slot_off, *counts;
    std::function<bool(uint64_t first, uint64_t n, char *dst)> fetch; // callable from several threads at once
};
bool write_dump_dir_text(const std::string &out_dir, const std::vector<std::string> &slot_kmers, const TextInput &in, unsigned n_threads, std::string &err);
bool touch_dump_files(const std::string &out_dir, const std::vector<std::string> &slot_kmers, std::string &err);

} // namespace pgh
