// reform_cli.cpp -- `poregen reform`: rewrite a basecaller move table (SAM/BAM tags mv:B:c, ns, ts) as a
// per-k-mer TSV or as the PAF + ss:Z: records `poregen gmove --paf` consumes.
//
// Replaces reform() of the reference (src/reform.cpp:79-371). Host-only: the work is one pass over a byte
// array per read, far below anything worth a kernel launch. The output is derived from the list of move
// positions of each read instead of the reference's three interleaved scans; the bytes written are the same
// (pinned by the reference's own expected files, tests/golden/reform/).
#include "pg_host.h"

#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <getopt.h>
#include <string>
#include <vector>

namespace {

constexpr uint32_t kStride = 5; // EXPECTED_STRIDE, src/reform.cpp:22

const struct option kLongOptions[] = {
    {"kmer_length", required_argument, nullptr, 'k'},
    {"sig_move_offset", required_argument, nullptr, 'm'},
    {"", no_argument, nullptr, 'c'},
    {"threads", required_argument, nullptr, 't'},
    {"batchsize", required_argument, nullptr, 'K'},
    {"max-bytes", required_argument, nullptr, 'B'},
    {"verbose", required_argument, nullptr, 'v'},
    {"help", no_argument, nullptr, 'h'},
    {"version", no_argument, nullptr, 'V'},
    {"output", required_argument, nullptr, 'o'},
    {"debug-break", required_argument, nullptr, 0},
    {nullptr, 0, nullptr, 0}};

void usage(FILE *fp, uint32_t k, uint32_t m) {
    fprintf(fp, "Usage: poregen reform basecalled.SAM/BAM\n\nbasic options:\n");
    fprintf(fp, "   -k, --kmer_length          kmer length [%u]\n", k);
    fprintf(fp, "   -m, --sig_move_offset      signal move offset [%u]\n", m);
    fprintf(fp, "   -c                         write move table in paf format\n");
    fprintf(fp, "   -h                         help\n");
    fprintf(fp, "   -o FILE                    output to file [stdout]\n");
    fprintf(fp, "   --verbose INT              verbosity level\n");
    fprintf(fp, "   --version                  print version\n");
}

int fail(const char *msg) {
    fprintf(stderr, "[reform::ERROR]\033[1;31m %s\033[0m\n", msg);
    return -1;
}

// One record. Returns 0, or -1 where the reference returns -1 (src/reform.cpp:212-240,344-357) or would read
// past the end of the mv array (src/reform.cpp:248-254,287-294: fewer than sig_move_offset+1 moves).
int reform_record(FILE *out, const pgh::MoveRec &r, uint32_t k, uint32_t m, bool paf) {
    if (!r.has_ns) return fail("tag 'ns' is not found. Please check your SAM/BAM file: ");
    if (!r.has_ts) return fail("tag 'ts' is not found. Please check your SAM/BAM file: ");
    if (!r.has_mv) return fail("NULL returned for tag mv: ");
    if (!r.mv_is_Bc) return fail("tag 'mv' specification is incorrect");
    if (r.mv_len == 0) return fail("mv array length is 0: ");
    if (r.stride != (int)kStride) return fail("expected stride of 5 is missing.");

    const uint32_t len_mv = r.mv_len;
    const int64_t ns = (int64_t)r.ns, ts = (int64_t)r.ts;
    // 1-based positions in mv[] of the moves
    std::vector<uint32_t> pos;
    for (uint32_t i = 1; i < len_mv; i++)
        if (r.is_one[i - 1]) pos.push_back(i);
    if (pos.size() < (size_t)m + 1) return fail("the move table holds fewer moves than sig_move_offset + 1");

    // number of k-mers, with the reference's unsigned wrap-around when the read is shorter than k-1
    uint32_t n_kmers = (uint32_t)r.seq.size() - k + 1;
    const uint32_t first = pos[m];            // the move that opens k-mer 0
    const size_t n_after = pos.size() - (m + 1); // moves that close a k-mer
    const bool body = first + 1 <= len_mv - 1;  // at least one mv element lies after `first`
    const char *id = r.qname.c_str();

    if (!paf) { // src/reform.cpp:245-282
        uint64_t start = (uint64_t)(ts + ((int64_t)first - 1) * kStride);
        uint32_t kmer_idx = 0;
        for (size_t j = 0; j < n_after && n_kmers > 0; j++, n_kmers--) {
            const uint64_t end = (uint64_t)(ts + ((int64_t)pos[m + 1 + j] - 1) * kStride);
            fprintf(out, "%s\t%" PRIu32 "\t%" PRIu64 "\t%" PRIu64 "\n", id, kmer_idx++, start, end);
            start = end;
        }
        if (body && n_kmers > 0) // the last k-mer runs to the end of the signal
            fprintf(out, "%s\t%" PRIu32 "\t%" PRIu64 "\t%" PRIu64 "\n", id, kmer_idx, start, (uint64_t)ns);
        return 0;
    }

    // PAF, src/reform.cpp:284-358. Column 4: the raw-signal end = the move that closes the last k-mer
    // (counted over ALL moves, the skipped ones included), or ns when the table runs out first.
    const uint32_t want = n_kmers + m + 1;
    uint64_t raw_end;
    if (want > pos.size()) raw_end = (uint64_t)ns;
    else raw_end = (uint64_t)(ts + ((int64_t)(want ? pos[want - 1] : 2u) - 1) * kStride);
    fprintf(out, "%s\t%" PRIu64 "\t%" PRIu64 "\t%" PRIu64 "\t+\t%s\t%" PRIu32 "\t0\t%" PRIu32 "\t%" PRIu32 "\t%" PRIu32 "\t255\tss:Z:", id, (uint64_t)ns,
            (uint64_t)(ts + ((int64_t)first - 1) * kStride), raw_end, id, n_kmers, n_kmers, n_kmers, n_kmers);
    uint32_t prev = first;
    for (size_t j = 0; j < n_after && n_kmers > 0; j++, n_kmers--) {
        fprintf(out, "%" PRIu32 ",", (pos[m + 1 + j] - prev) * kStride);
        prev = pos[m + 1 + j];
    }
    if (body && n_kmers > 0) {
        const uint32_t last = len_mv - 1;
        const int64_t tail = ns - ((int64_t)((last - 1) * kStride) + ts);
        if (tail < 0) return fail("Error in calcuation. (ns - ((i-1)*EXPECTED_STRIDE + ts)) > 0 is not valid");
        n_kmers--;
        fprintf(out, "%" PRIu32 ",", (uint32_t)((last - prev) * kStride + tail));
    }
    if (n_kmers != 0) {
        fprintf(stderr, "[reform::ERROR]\033[1;31m Error in the implementation. Please report the command with minimal reproducible data. Read_id: %s\033[0m\n", id);
        return -1;
    }
    fputc('\n', out);
    return 0;
}

} // namespace

int reform_main(int argc, char **argv) {
    uint32_t k = 9, m = 0; // init_opt, src/poregen.cpp:209-237
    bool paf = false, help_to_stdout = false;
    const char *out_path = nullptr;
    int c, longindex = 0;
    optind = 1;
    while ((c = getopt_long(argc, argv, "k:m:ct:B:K:v:o:hV", kLongOptions, &longindex)) >= 0) {
        if (c == 'k') k = (uint32_t)atoi(optarg);
        else if (c == 'm') m = (uint32_t)atoi(optarg);
        else if (c == 'c') paf = true;
        else if (c == 'K') { if (atoi(optarg) < 1) { fail("Batch size should larger than 0."); exit(EXIT_FAILURE); } }
        else if (c == 't') { if (atoi(optarg) < 1) { fail("Number of threads should larger than 0."); exit(EXIT_FAILURE); } }
        else if (c == 'o') out_path = optarg;
        else if (c == 'V') { fprintf(stdout, "subtool0 0.1.0\n"); exit(EXIT_SUCCESS); }
        else if (c == 'h') help_to_stdout = true;
    }
    if (k < 1) return fail("kmer length must be a positive integer");
    if (k <= m) return fail("signal move offset value must less than the kmer length");
    if (argc - optind != 1 || help_to_stdout) {
        fail("not enough arguments");
        usage(help_to_stdout ? stdout : stderr, k, m);
        if (help_to_stdout) exit(EXIT_SUCCESS);
        return EXIT_FAILURE;
    }
    const char *in_path = argv[optind];
    fprintf(stderr, "bam_file : %s\nkmer length : %" PRIu32 "\nsignal move offset : %" PRIu32 "\noutput format : %s\n", in_path, k, m, paf ? "paf" : "tsv");

    FILE *out = stdout;
    if (out_path) {
        out = fopen(out_path, "w");
        if (!out) { fprintf(stderr, "[reform::ERROR]\033[1;31m Failed to open %s\033[0m\n", out_path); exit(EXIT_FAILURE); }
    }
    pgh::SamBamReader rd;
    std::string err;
    if (!rd.open(in_path, err)) { fprintf(stderr, "[reform::ERROR]\033[1;31m %s: %s\033[0m\n", in_path, err.c_str()); exit(EXIT_FAILURE); }
    pgh::MoveRec rec;
    int got, ret = 0;
    while ((got = rd.next(rec, err)) > 0)
        if ((ret = reform_record(out, rec, k, m, paf)) != 0) break;
    if (got < 0) { fail(err.c_str()); ret = -1; }
    if (out_path) fclose(out);
    else fflush(out);
    return ret;
}
