// gmove_cli.cpp -- `poregen gmove`: the reference's command line and output layout (src/gmove.cpp:213-537)
// over libpgmove (include/pgmove.h). Host work here: option handling, k-mer list and slice, SLOW5/PAF/FASTQ
// parsing into batches, writing the dump directory. The per-read computation runs on the GPU.
#include "../../../include/pgmove.h"
#include "pg_host.h"
#include <future>
#include <algorithm>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <getopt.h>
#include <string>
#include <thread>
#include <chrono>
#include <atomic>
#include <functional>
#include <vector>

#define POREGEN_VERSION "0.1.0"

namespace {

struct Opt { // opt_t subset, defaults of init_opt (src/poregen.cpp:209-237, src/poregen.h:30-43)
    uint32_t kmer_size = 9, sig_move_offset = 0, kmer_start_offset = 0, signal_print_margin = 0, sample_limit = 100;
    uint32_t file_limit = 500, index_start = 1, index_end = 500, max_dur = 70, min_dur = 5, kmer_pick_margin = 2;
    int delimit_files = 0, flag_rna = 0;
    double pa_max = 180.0, pa_min = 40.0;
};

// same order as the reference's table: handlers below test longindex like the reference does (gmove.cpp:49-72)
struct option long_options[] = {
    {"kmer_size", required_argument, 0, 'k'}, {"sig_move_offset", required_argument, 0, 'm'},
    {"kmer_start_offset", required_argument, 0, 's'}, {"scaling", required_argument, 0, 0},
    {"margin", required_argument, 0, 0}, {"sample_limit", required_argument, 0, 0},
    {"file_limit", required_argument, 0, 0}, {"kmer_file", required_argument, 0, 0},
    {"index_start", required_argument, 0, 0}, {"index_end", required_argument, 0, 0},
    {"fastq", required_argument, 0, 0}, {"", no_argument, 0, 'd'},
    {"max_dur", required_argument, 0, 0}, {"min_dur", required_argument, 0, 0},
    {"pa_min", required_argument, 0, 0}, {"pa_max", required_argument, 0, 0},
    {"kmer_pick_margin", required_argument, 0, 0}, {"rna", no_argument, 0, 0},
    {"verbose", required_argument, 0, 'v'}, {"help", no_argument, 0, 'h'},
    {"version", no_argument, 0, 'V'}, {"debug-break", required_argument, 0, 0},
    // extensions of this implementation (not in the reference)
    {"batch_reads", required_argument, 0, 0}, {"device", required_argument, 0, 0}, {"lazy_stats", no_argument, 0, 0},
    {"raw_model", required_argument, 0, 0}, {"stdv_limit", required_argument, 0, 0}, {"dwell_model", required_argument, 0, 0},
    {"devices", required_argument, 0, 0}, {"exchange", required_argument, 0, 0},
    {0, 0, 0, 0}};

void print_help(FILE *fp, const Opt &o) { // src/gmove.cpp:80-104
    fprintf(fp, "Usage: poregen gmove reads.blow5 event_alignment_file output_dir\n");
    fprintf(fp, "\nbasic options:\n");
    fprintf(fp, "   -k INT                     kmer_size [%d]\n", o.kmer_size);
    fprintf(fp, "   -m INT                     move start offset [%d]\n", o.sig_move_offset);
    fprintf(fp, "   -s INT                     kmer start offset [%d]\n", o.kmer_start_offset);
    fprintf(fp, "   --scaling INT              scaling [%d] (0-no scaling, 1-medmad)\n", 1);
    fprintf(fp, "   --margin INT               signal print margin on both sides of the sub signal[%u] \n", o.signal_print_margin);
    fprintf(fp, "   --sample_limit INT         maximum number of instances to output for a kmer [%u] \n", o.sample_limit);
    fprintf(fp, "   --file_limit INT           maximum number of kmer files to output [%u] \n", o.file_limit);
    fprintf(fp, "   --kmer_file FILE           kmer file (optional) \n");
    fprintf(fp, "   --index_start INT          1-based closed interval index of start kmer [%u] \n", o.file_limit);
    fprintf(fp, "   --index_end INT            1-based closed interval index of end kmer [%u] \n", o.file_limit);
    fprintf(fp, "   --fastq FILE               fastq file (optional - should be provided with .paf) \n");
    fprintf(fp, "   -d                         delimit output files per read\n");
    fprintf(fp, "   --max_dur                  maximum move duration allowed for samples [%d]\n", o.max_dur);
    fprintf(fp, "   --min_dur                  maximum move duration allowed for samples [%d]\n", o.min_dur);
    fprintf(fp, "   --pa_min                   minimum pA level a sampling signal should have [%.3f]\n", o.pa_min);
    fprintf(fp, "   --pa_max                  maximum pA level a sampling signal should have [%.3f]\n", o.pa_max);
    fprintf(fp, "   --kmer_pick_margin         distance in bases from an indel when picking a kmer as sample [%d]\n", o.kmer_pick_margin);
    fprintf(fp, "   --rna                      dataset is rna\n");
    fprintf(fp, "   -h                         help\n");
    fprintf(fp, "   --verbose INT              verbosity level [%d]\n", 3);
    fprintf(fp, "   --version                  print version\n");
    fprintf(fp, "\nMI355X implementation options:\n");
    fprintf(fp, "   --batch_reads INT          reads per batch [20000 per device: a batch is cut into one shard per device]\n");
    fprintf(fp, "   --device INT               HIP device [0]\n");
    fprintf(fp, "   --devices LIST             several GPUs of this node, e.g. 0,1,2,3,4,5,6,7: every batch is cut into contiguous shards,\n");
    fprintf(fp, "                              one per listed device, with one RCCL all-gather of per-k-mer counts per batch (pg_job_*);\n");
    fprintf(fp, "                              a device may be listed twice (the exchange then goes through host memory)\n");
    fprintf(fp, "   --exchange auto|host|rccl  how --devices exchanges the counts [auto]\n");
    fprintf(fp, "   --lazy_stats               median/MAD only for reads that contribute a kept event\n");
    fprintf(fp, "   --raw_model FILE           also write KMER<TAB>median<TAB>stddev of the kept samples per k-mer, computed on the GPU\n");
    fprintf(fp, "                              (what scripts/poregen.sh calculate_mean_stddev_all derives from the dump files)\n");
    fprintf(fp, "   --stdv_limit NUM           cap of the stddev column of --raw_model [3.1]\n");
    fprintf(fp, "   --dwell_model FILE         also write KMER<TAB>median dwell (scripts/poregen.sh calculate_dwell_times_medians)\n");
}

// The signal of a batch is hundreds of MB: a std::vector would zero every byte on resize (one thread, and the page faults
// with it) before the pool threads overwrite it. Plain storage on huge pages (pgh::HugeBuf); the first touch happens in the copying threads.
struct SampleBuf {
    pgh::HugeBuf hb; size_t n = 0;
    int16_t *data() { return static_cast<int16_t *>(hb.p); } const int16_t *data() const { return static_cast<const int16_t *>(hb.p); }
    size_t size() const { return n; }
    void clear() { n = 0; }
    void reserve(size_t want) {
        const size_t cap = hb.bytes / sizeof(int16_t);
        if (want <= cap) return;
        size_t c = cap ? cap : 4096; while (c < want) c += c / 2 + 4096;
        if (!hb.grow(c * sizeof(int16_t), n * sizeof(int16_t))) { fprintf(stderr, "[gmove] out of memory for %zu samples\n", c); exit(EXIT_FAILURE); }
    }
    void resize(size_t want) { reserve(want); n = want; } // new elements are NOT initialised
    void append(const int16_t *a, const int16_t *b) { const size_t k = (size_t)(b - a); reserve(n + k); if (k) memcpy(data() + n, a, k * sizeof(int16_t)); n += k; }
};

struct HostBatch {
    SampleBuf sig;
    std::vector<uint64_t> sig_off{0}, seq_off{0}, op_off{0};
    std::vector<double> dig, off, range;
    std::vector<int32_t> qs, ts, te;
    std::vector<uint8_t> seq, op_t;
    std::vector<uint32_t> op_n;
    uint32_t n() const { return (uint32_t)dig.size(); }
    void clear() {
        sig.clear(); sig_off.assign(1, 0); seq_off.assign(1, 0); op_off.assign(1, 0); dig.clear(); off.clear(); range.clear();
        qs.clear(); ts.clear(); te.clear(); seq.clear(); op_t.clear(); op_n.clear();
    }
};

int die(const char *fmt, const std::string &a = "") { fprintf(stderr, fmt, a.c_str()); fputc('\n', stderr); return EXIT_FAILURE; }

// the device side of the run: one context (--device) or one job over several GPUs (--devices), same calls either way
struct Backend {
    pg_ctx *ctx = nullptr; pg_job *job = nullptr;
    bool ok() const { return ctx || job; }
    pg_status submit(const pg_batch *b) { return job ? pg_job_submit(job, b) : pg_submit(ctx, b); }
    pg_status sync() { return job ? pg_job_sync(job) : pg_sync(ctx); }
    bool all_full() { return job ? pg_job_all_slots_full(job) != 0 : pg_all_slots_full(ctx) != 0; }            // waits for the device
    int32_t poll() { return job ? pg_job_poll(job) : pg_poll(ctx); } // 1: the device has finished the last batch (errors < 0), 0: not yet; no wait
    bool all_full_settled() const { return job ? pg_job_all_slots_full_settled(job) != 0 : pg_all_slots_full_settled(ctx) != 0; } // as of the batch already waited for
    // the kept samples stay on the device (a job: concatenated on its first device) and the dump writers fetch them range by range,
    // or the files' text itself is produced there (pg_text / pg_job_text)
    pg_status finish(pg_result *r) { return job ? pg_job_finish_deferred(job, r) : pg_finish_deferred(ctx, r); }
    bool fetch(uint64_t first, uint64_t n, double *dst) { return (job ? pg_job_fetch_samples(job, first, n, dst) : pg_fetch_samples(ctx, first, n, dst)) == PG_OK; }
    pg_status text(pg_text_result *t) { return job ? pg_job_text(job, t) : pg_text(ctx, t); }
    bool fetch_text(uint64_t first, uint64_t n, char *dst) { return (job ? pg_job_fetch_text(job, first, n, dst) : pg_fetch_text(ctx, first, n, dst)) == PG_OK; }
    pg_status model(pg_model_result *m) { return job ? pg_job_model(job, 0, m) : pg_model(ctx, 0, m); }
    const char *error() const { return job ? pg_job_last_error(job) : pg_last_error(ctx); }
    void destroy() { if (job) pg_job_destroy(job); if (ctx) pg_destroy(ctx); job = nullptr; ctx = nullptr; }
};

} // namespace

int gmove_main(int argc, char **argv) {
    const std::chrono::steady_clock::time_point t_main0 = std::chrono::steady_clock::now();
    Opt opt;
    int longindex = 0, c, signal_scale = 0;
    const char *input_kmer_file = nullptr, *input_fastq_file = nullptr;
    FILE *fp_help = stderr;
    uint32_t batch_reads = 20000; bool batch_reads_set = false; int device = 0; bool lazy = false;
    std::vector<int32_t> devices; uint32_t exchange = PG_JOB_EXCHANGE_AUTO;
    const char *raw_model_path = nullptr, *dwell_model_path = nullptr, *stdv_limit = "3.1";
    optind = 1;
    while ((c = getopt_long(argc, argv, "k:m:s:d", long_options, &longindex)) >= 0) { // src/gmove.cpp:240-327
        if (c == 'k') { if (atoi(optarg) < 1) { fprintf(stderr, "Kmer length should be larger than 0. You entered %d\n", atoi(optarg)); return EXIT_FAILURE; } opt.kmer_size = atoi(optarg); }
        else if (c == 'm') { if (atoi(optarg) < 0) { fprintf(stderr, "signal move offset value must not be less than zero. You entered %d\n", atoi(optarg)); return EXIT_FAILURE; } opt.sig_move_offset = atoi(optarg); }
        else if (c == 's') { if (atoi(optarg) < 1) { fprintf(stderr, "Kmer offset should be larger than 0. You entered %d\n", atoi(optarg)); return EXIT_FAILURE; } opt.kmer_start_offset = atoi(optarg); }
        else if (c == 'd') opt.delimit_files = 1;
        else if (c == 'v') {}
        else if (c == 'V') { fprintf(stdout, "gmove %s\n", POREGEN_VERSION); return EXIT_SUCCESS; }
        else if (c == 'h') fp_help = stdout;
        else if (c == 0 && longindex == 3) signal_scale = atoi(optarg);
        else if (c == 0 && longindex == 4) { if (atoi(optarg) < 0) return die("Signal print margin should be non negative."); opt.signal_print_margin = atoi(optarg); }
        else if (c == 0 && longindex == 5) { if (atoi(optarg) < 0) return die("Maximum number of instances to output for a kmer should be non negative."); opt.sample_limit = atoi(optarg); }
        else if (c == 0 && longindex == 6) { if (atoi(optarg) < 0) return die("Maximum number of kmer files to output should be non negative."); opt.file_limit = atoi(optarg); opt.index_end = opt.index_start + opt.file_limit - 1; }
        else if (c == 0 && longindex == 7) input_kmer_file = optarg;
        else if (c == 0 && longindex == 8) { if (atoi(optarg) < 1) return die("kmer index start should be a positive number."); opt.index_start = atoi(optarg); opt.file_limit = opt.index_end - opt.index_start + 1; }
        else if (c == 0 && longindex == 9) { if (atoi(optarg) < 1) return die("kmer index end should be a positive number."); opt.index_end = atoi(optarg); opt.file_limit = opt.index_end - opt.index_start + 1; }
        else if (c == 0 && longindex == 10) input_fastq_file = optarg;
        else if (c == 0 && longindex == 12) opt.max_dur = atoi(optarg);
        else if (c == 0 && longindex == 13) opt.min_dur = atoi(optarg);
        else if (c == 0 && longindex == 14) opt.pa_min = atof(optarg);
        else if (c == 0 && longindex == 15) opt.pa_max = atof(optarg);
        else if (c == 0 && longindex == 16) opt.kmer_pick_margin = atoi(optarg);
        else if (c == 0 && longindex == 17) opt.flag_rna = 1;
        else if (c == 0 && longindex == 22) { batch_reads = (uint32_t)std::max(1, atoi(optarg)); batch_reads_set = true; }
        else if (c == 0 && longindex == 23) device = atoi(optarg);
        else if (c == 0 && longindex == 24) lazy = true;
        else if (c == 0 && longindex == 25) raw_model_path = optarg;
        else if (c == 0 && longindex == 26) stdv_limit = optarg;
        else if (c == 0 && longindex == 27) dwell_model_path = optarg;
        else if (c == 0 && longindex == 28) {
            for (const char *q = optarg; *q;) {
                char *e = nullptr; const long v = strtol(q, &e, 10);
                if (e == q || v < 0 || (*e && *e != ',')) return die("--devices takes a comma-separated list of device ordinals. You entered %s", optarg);
                devices.push_back((int32_t)v); q = *e ? e + 1 : e;
            }
            if (devices.empty()) return die("--devices takes a comma-separated list of device ordinals. You entered %s", optarg);
        }
        else if (c == 0 && longindex == 29) {
            if (!strcmp(optarg, "auto")) exchange = PG_JOB_EXCHANGE_AUTO; else if (!strcmp(optarg, "host")) exchange = PG_JOB_EXCHANGE_HOST;
            else if (!strcmp(optarg, "rccl")) exchange = PG_JOB_EXCHANGE_RCCL; else return die("--exchange must be auto, host or rccl. You entered %s", optarg);
        }
    }
    if (argc - optind != 3 || fp_help == stdout) { // src/gmove.cpp:330-336
        print_help(fp_help, opt);
        return fp_help == stdout ? EXIT_SUCCESS : EXIT_FAILURE;
    }
    const char *slow5file = argv[optind], *move_table = argv[optind + 1], *output_dir = argv[optind + 2];
    // --devices cuts every batch into one contiguous shard per device: the default batch grows with the device count, so that a shard
    // stays at the 20 000 reads (160 MB of signal) the one-device pipeline is tuned for instead of shrinking to a latency-bound sliver
    const size_t n_shards = devices.empty() ? 1 : devices.size();
    if (!batch_reads_set) batch_reads = (uint32_t)std::min<uint64_t>(20000ull * n_shards, 0x7fffffffull);
    const size_t batch_samples_cap = ((size_t)1 << 29) * n_shards; // and so does the byte budget of a batch (1 GB of samples per device)
    // The reference stops reading once every k-mer of the WHOLE list is complete (gmove.cpp:733-735): at the default sample_limit that is a few
    // thousand reads into the file. A first batch of 20 000 reads then does ~6 x the reference's work before anyone can look,
    // so a whole-list job at the default batch size ramps up: 2 048 reads per device, then twice that, ... up to --batch_reads -- the job ends
    // within a factor of two of the completing read, a job that never completes pays four small batches. An explicit --batch_reads, or a
    // slice (which reads every line anyway), keeps its one size. (POREGEN_BATCH_RAMP=0: off, =N: first batch of N reads per device.)
    uint32_t ramp_reads = batch_reads; // reads of the NEXT batch; doubled by next_batch_reads() up to batch_reads
    bool ramp_armed = !batch_reads_set; // ... and only for a job on the whole list (known below: whole_list)
    if (const char *rv = getenv("POREGEN_BATCH_RAMP")) { const long v = atol(rv); if (v <= 0) ramp_armed = false; else { ramp_armed = true; ramp_reads = (uint32_t)std::min<uint64_t>((uint64_t)v * n_shards, batch_reads); } }
    else if (ramp_armed) ramp_reads = (uint32_t)std::min<uint64_t>(2048ull * n_shards, batch_reads);
    if (getenv("POREGEN_BATCH_PROBE")) fprintf(stderr, "[batch probe] batch_reads %u (%zu device%s x %u)\n", batch_reads, n_shards, n_shards == 1 ? "" : "s", (unsigned)(batch_reads / n_shards));
    // the HIP runtime takes 0.1-0.2 s to come up: it starts NOW, on a thread of its own, next to the directory set-up, the k-mer list,
    // the file indices and the parsing of the first batch; the context is created behind it
    const int first_device = devices.empty() ? device : devices[0];
    std::future<pg_status> rt_ready = std::async(std::launch::async, [first_device]() { return pg_runtime_init(first_device); });
    if (opt.kmer_size <= opt.sig_move_offset) fprintf(stderr, "[gmove::WARNING] signal move offset value should be less than the kmer length\n");

    if (raw_model_path && opt.delimit_files) return die("--raw_model cannot be combined with -d: the ':' delimiters are not numbers (datamash stops on them)");
    { char *end = nullptr; (void)strtold(stdv_limit, &end); if (end == stdv_limit || *end) return die("--stdv_limit must be a number. You entered %s", stdv_limit); }

    int rcd = pgh::create_dir(output_dir); // src/gmove.cpp:374-392
    if (rcd == -1) { fprintf(stderr, "Output directory %s is not empty. Please remove it or specify another directory.\n", output_dir); return EXIT_FAILURE; }
    if (rcd == -2) { fprintf(stderr, "Could not create the output dir %s.\n", output_dir); return EXIT_FAILURE; }
    const std::string kmer_dump = std::string(output_dir) + "/dump";
    rcd = pgh::create_dir(kmer_dump.c_str());
    if (rcd == -1) { fprintf(stderr, "Output directory %s is not empty. Please remove it or specify another directory.", kmer_dump.c_str()); return EXIT_FAILURE; }
    if (rcd == -2) { fprintf(stderr, "Could not create the output dir %s.", kmer_dump.c_str()); return EXIT_FAILURE; }

    std::vector<std::string> kmers; // src/gmove.cpp:394-426
    std::string err;
    if (input_kmer_file) {
        int r = pgh::read_kmer_file(input_kmer_file, (int)opt.kmer_size, kmers, err);
        if (r != 0) { fprintf(stderr, "%s\n", err.c_str()); return EXIT_FAILURE; }
    } else {
        if (opt.kmer_size > 13) return die("kmer sizes above 13 are not supported by this implementation");
        pgh::generate_kmers((int)opt.kmer_size, opt.flag_rna != 0, kmers);
    }

    uint32_t num_kmers = (uint32_t)kmers.size(); // src/gmove.cpp:428-440
    fprintf(stderr, "num_kmers: %d\n", num_kmers);
    if (opt.file_limit < num_kmers) {
        num_kmers = opt.file_limit;
        fprintf(stderr, "only dumping %d kmers in kmer interval [%d-%d]\n", num_kmers, opt.index_start, opt.index_end);
    } else if (opt.file_limit > num_kmers - opt.index_start + 1) {
        if (opt.index_end > (uint32_t)kmers.size()) { opt.file_limit = (uint32_t)kmers.size() - opt.index_start + 1; opt.index_end = opt.index_start + opt.file_limit - 1; }
        else opt.file_limit = opt.index_end - opt.index_start + 1;
    }
    // the reference indexes kmers[i] for i in [index_start-1, index_end) without a bounds check
    if (opt.index_start < 1 || opt.index_end > kmers.size() || opt.index_end + 1 < opt.index_start) {
        fprintf(stderr, "k-mer interval [%u-%u] is outside the k-mer list (%zu k-mers)\n", opt.index_start, opt.index_end, kmers.size());
        return EXIT_FAILURE;
    }
    fprintf(stderr, "slow5_file_path: %s\n", slow5file);
    fprintf(stderr, "guppy_sam_output_file: %s\n", move_table);
    fprintf(stderr, "kmer_output_dir: %s\n", output_dir);
    fprintf(stderr, "kmer_size: %d\n", opt.kmer_size);
    fprintf(stderr, "sig_move_offset: %d\n", opt.sig_move_offset);
    fprintf(stderr, "signal_print_margin: %d\n", opt.signal_print_margin);
    fprintf(stderr, "kmer index closed interval : [%d-%d]\n", opt.index_start, opt.index_end);
    fprintf(stderr, "no.of output files: %d\n", opt.file_limit);
    fprintf(stderr, "sample limit : %d\n", opt.sample_limit);
    fprintf(stderr, "sample max duration : %d\n", opt.max_dur);
    fprintf(stderr, "sample min duration : %d\n", opt.min_dur);
    fprintf(stderr, "sample kmer_pick_margin : %d\n", opt.kmer_pick_margin);
    fprintf(stderr, "dataset: %s\n", opt.flag_rna ? "RNA" : "DNA");

    std::vector<std::string> slot_kmers(kmers.begin() + (opt.index_start - 1), kmers.begin() + opt.index_end);
    if (!pgh::touch_dump_files(output_dir, slot_kmers, err)) { fprintf(stderr, "%s\n", err.c_str()); return EXIT_FAILURE; } // gmove.cpp:460-473

    int scaling; // src/gmove.cpp:479-491
    if (signal_scale == 0) { scaling = 0; fprintf(stderr, "scaling: %s\n", "no scale"); }
    else if (signal_scale == 1) { scaling = 1; fprintf(stderr, "scaling: %s\n", "medmad scale"); }
    else { print_help(fp_help, opt); return EXIT_FAILURE; }

    const std::string mt(move_table); // src/gmove.cpp:505-521
    const std::string ext = mt.size() >= 4 ? mt.substr(mt.size() - 4) : "";
    const bool is_paf = ext == ".paf", is_bam = ext == ".bam" || ext == ".sam";

    // ---- device context: created on its own thread while this one builds the file indices (the HIP runtime takes 0.1-0.2 s
    // to come up, about as long as indexing a compressed BLOW5) -------------------------------------------------------------
    if (opt.kmer_size > 13) return die("kmer sizes above 13 are not supported by this implementation");
    std::vector<int32_t> table_t((size_t)1 << (2 * opt.kmer_size)), table_u(table_t.size());
    {
        std::vector<const char *> ptrs; for (auto &s : slot_kmers) ptrs.push_back(s.c_str());
        if (pg_build_slot_tables(opt.kmer_size, ptrs.data(), (uint32_t)ptrs.size(), table_t.data(), table_u.data()) != PG_OK) {
            fprintf(stderr, "%s\n", pg_last_error(nullptr)); return EXIT_FAILURE;
        }
    }
    pg_params prm; pg_default_params(&prm);
    prm.kmer_size = opt.kmer_size; prm.sig_move_offset = opt.sig_move_offset; prm.signal_print_margin = opt.signal_print_margin;
    prm.sample_limit = opt.sample_limit; prm.max_dur = opt.max_dur; prm.min_dur = opt.min_dur; prm.kmer_pick_margin = (int32_t)opt.kmer_pick_margin;
    if (!is_paf) { // the move-table front-end has no indel logic and resolves -m / -s on the host (gmove.cpp:620-639)
        prm.kmer_pick_margin = 0; prm.sig_move_offset = 0;
    }
    prm.scaling = scaling; prm.allow_rna = opt.flag_rna; prm.pa_min = opt.pa_min; prm.pa_max = opt.pa_max;
    const bool whole_list = slot_kmers.size() == kmers.size(); // only then can the reference's loop end early (gmove.cpp:733-735)
    if (!whole_list || !ramp_armed) ramp_reads = batch_reads; // the ramp of the batch sizes (above) is for jobs that can end early
    auto next_batch_reads = [&]() -> uint32_t { const uint32_t v = ramp_reads; ramp_reads = (uint32_t)std::min<uint64_t>(2ull * ramp_reads, batch_reads); return v; };
    uint32_t this_batch_reads = next_batch_reads();
    prm.n_slots = (uint32_t)slot_kmers.size(); prm.flags = PG_FLAG_ONE_STREAM /* a job of a few batches: no second hardware queue (15-20 ms) */ | (whole_list ? PG_FLAG_STOP_WHEN_FULL : 0) | (lazy ? PG_FLAG_LAZY_STATS : 0) | (is_paf ? 0 : PG_FLAG_SHORT_READS_OK) | (is_bam ? PG_FLAG_SKIP_OUT_OF_RANGE : 0);
    prm.device = device;
    prm.table_t = table_t.data(); prm.table_u = table_u.data();
    Backend dev;
    std::string ctx_err; double t_ctx = 0;
    const std::chrono::steady_clock::time_point t_setup0 = std::chrono::steady_clock::now();
    std::future<pg_status> ctx_ready = std::async(std::launch::async, [&]() {
        const std::chrono::steady_clock::time_point a = std::chrono::steady_clock::now();
        pg_status st = rt_ready.get(); // (an error is reported again, with its text, by pg_create below)
        if (devices.empty()) { st = pg_create(&prm, &dev.ctx); if (st != PG_OK) ctx_err = pg_last_error(nullptr); } // the text lives in the creating thread
        else { st = pg_job_create(&prm, devices.data(), (uint32_t)devices.size(), exchange, &dev.job); if (st != PG_OK) ctx_err = pg_job_last_error(nullptr); }
        t_ctx = std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
        return st;
    });
    auto give_up = [&](int code) { if (ctx_ready.valid() && ctx_ready.get() == PG_OK) dev.destroy(); return code; }; // early error exits

    pgh::Slow5File s5;
    if (!s5.open(slow5file, err)) { fprintf(stderr, "Error in opening file %s\n[gmove] %s\n", slow5file, err.c_str()); return give_up(EXIT_FAILURE); } // gmove.cpp:493-503 (+ the reader's reason: slow5lib prints its own there)

    pgh::FastxIndex fai;
    pgh::SamBamReader sam;
    if (is_bam && !sam.open(move_table, err)) { fprintf(stderr, "[gmove] %s\n", err.c_str()); return give_up(EXIT_FAILURE); } // F_CHK(bam_fp), gmove.cpp:1067-1068
    if (is_paf) {
        if (!input_fastq_file) { fprintf(stderr, ".paf input requires an additional .fastq file\n"); return give_up(EXIT_FAILURE); } // gmove.cpp:510-513
        if (!fai.load(input_fastq_file, err)) { fprintf(stderr, "Error in loading fastq index for %s\n", input_fastq_file); return give_up(EXIT_FAILURE); }
    }
    FILE *paf_fp = fopen(move_table, "r");
    if (!paf_fp) { fprintf(stderr, "Error in opening file %s\n", move_table); return give_up(EXIT_FAILURE); }
    const std::chrono::steady_clock::time_point t_setup1 = std::chrono::steady_clock::now();
    double t_ctx_wait = 0; // the context is awaited at its first use (the first batch), behind the parsing of that batch
    auto need_ctx = [&]() -> bool {
        if (!ctx_ready.valid()) return dev.ok();
        const std::chrono::steady_clock::time_point a = std::chrono::steady_clock::now();
        const pg_status st = ctx_ready.get();
        t_ctx_wait = std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
        if (st != PG_OK) { fprintf(stderr, "[gmove] %s\n", ctx_err.c_str()); return false; }
        return true;
    };

    // ---- the read loop (src/gmove.cpp:732-969), batched ---------------------------------------------------
    HostBatch hbs[2]; // two batches: one is parsed while the other is on its way through the device
    int cur = 0;
    pgh::Slow5Rec rec;
    std::string seq;
    char *line = nullptr; size_t cap = 0; ssize_t got;
    int status = EXIT_SUCCESS;
    uint64_t count_reads = 0, total_samples = 0;
    bool stop = false;
    std::atomic<bool> batch_all_matches(true); // cleared by whoever puts an I or a D op into the batch under construction
    using clk = std::chrono::steady_clock;
    auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    double t_device = 0, t_finish = 0, t_dump = 0, t_lines = 0, t_decode = 0, t_concat = 0;
    const clk::time_point t_loop0 = clk::now();
    bool flush_failed = false; // the move-table / SAM / BAM loop: a submit failed (as opposed to a bad record)
    auto flush = [&]() -> bool {
        if (hbs[cur].n() == 0) return true;
        const clk::time_point tf0 = clk::now();
        pg_batch b; memset(&b, 0, sizeof b);
        b.struct_size = sizeof b; b.location = PG_LOC_HOST; b.n_reads = hbs[cur].n();
        b.sig = hbs[cur].sig.data(); b.sig_off = hbs[cur].sig_off.data(); b.digitisation = hbs[cur].dig.data(); b.offset = hbs[cur].off.data(); b.range = hbs[cur].range.data();
        b.query_start = hbs[cur].qs.data(); b.target_start = hbs[cur].ts.data(); b.target_end = hbs[cur].te.data(); b.seq = hbs[cur].seq.data(); b.seq_off = hbs[cur].seq_off.data();
        b.op_n = hbs[cur].op_n.data(); b.op_t = hbs[cur].op_t.data(); b.op_off = hbs[cur].op_off.data();
        // PG_BATCH_ALL_MATCHES vouches that no read of the batch needs the generic walk: no I / D op (every ss string `reform` writes) AND
        // k <= ops <= fetched bases for every read -- a shorter or overlong read goes to that walk too (a basecall shorter than k on the
        // move-table / SAM / BAM front-ends simply has no events there, PG_FLAG_SHORT_READS_OK). The device verifies the claim.
        bool vouch = batch_all_matches;
        for (size_t r = 0; vouch && r < hbs[cur].n(); r++) {
            const uint64_t nops = hbs[cur].op_off[r + 1] - hbs[cur].op_off[r], slen = hbs[cur].seq_off[r + 1] - hbs[cur].seq_off[r];
            if (slen >= (uint64_t)opt.kmer_size && (nops < (uint64_t)opt.kmer_size || nops > slen)) vouch = false; // (slen < k: a skipped read, no walk)
        }
        if (vouch) b.flags |= PG_BATCH_ALL_MATCHES;
        batch_all_matches = true;
        if (!need_ctx()) return false;
        // Queued, not awaited: the next batch is parsed meanwhile. pg_submit itself waits for the batch BEFORE this one (and hands its
        // per-read errors back), so the buffers of that batch -- hbs[cur ^ 1], about to be refilled -- are no longer read by anyone.
        pg_status s = dev.submit(&b);
        if (s != PG_OK) { fprintf(stderr, "[gmove] %s\n", dev.error()); return false; }
        t_device += secs(tf0, clk::now());
        cur ^= 1;
        hbs[cur].clear();
        this_batch_reads = next_batch_reads();
        // every k-mer of the WHOLE list complete: the reference stops reading (gmove.cpp:733-735). With a slice it reads on (and would
        // still fail on a malformed later line), so we do too. Asked without waiting: this is the state behind the batch before the one
        // just queued; a batch too many changes nothing in the output (its events rank behind the complete files').
        if (whole_list && dev.all_full_settled()) stop = true;
        return true;
    };
    // Move-table style records (table file and SAM/BAM, gmove.cpp:557-700 / 1080-1261): resolve -m (first window starts
    // at the (m+1)-th move) and -s (first k-mer starts at base s) on the host, turn every closed move segment into a match
    // op of (gap x stride) samples, drop the trimmed prefix of the signal. The device then runs the same collector with
    // kmer_pick_margin 0 and no indels.
    std::vector<uint8_t> is_one;
    auto add_move_record = [&](const char *read_id, int fastq_len, const std::string &fseq, int stride, uint64_t signal_len, long long trim) -> bool {
        if (!s5.get(read_id, rec, err)) { fprintf(stderr, "Error in when fetching the read (%s)\n", err.c_str()); status = EXIT_FAILURE; return false; }  // gmove.cpp:582-586
        if (rec.raw.size() != signal_len || trim < 0 || (uint64_t)trim >= rec.raw.size()) {                                    // asserts, gmove.cpp:589-590
            fprintf(stderr, "move record of %s disagrees with the SLOW5 record (signal_len / trim_offset)\n", read_id); status = EXIT_FAILURE; return false;
        }
        hbs[cur].sig.append(rec.raw.data() + trim, rec.raw.data() + rec.raw.size()); // gmove.cpp:591-598: only the trimmed signal is used
        total_samples += rec.raw.size();
        seq.clear();
        uint32_t qstart = 0;
        const size_t move_len = is_one.size();
        if (fastq_len >= 10) { // gmove.cpp:616-619: shorter reads are skipped (no events, no ':')
            size_t idx = 0, start_idx = 0; uint32_t ones = 0;
            while (ones < opt.sig_move_offset + 1 && idx < move_len) { if (is_one[idx]) { ones++; start_idx = idx; } idx++; } // gmove.cpp:623-629
            if (ones < opt.sig_move_offset + 1) { fprintf(stderr, "move array of %s has fewer than %u moves\n", read_id, opt.sig_move_offset + 1); status = EXIT_FAILURE; return false; }
            if (opt.kmer_start_offset > fseq.size()) { fprintf(stderr, "kmer start offset beyond the sequence of %s\n", read_id); status = EXIT_FAILURE; return false; }
            seq = fseq.substr(opt.kmer_start_offset);
            qstart = (uint32_t)(start_idx * (size_t)stride);
            size_t n_seg = 0, prev = start_idx;
            for (size_t i = start_idx + 1; i < move_len; i++) // the last move is never closed (gmove.cpp:632, 1195)
                if (is_one[i]) {
                    if (n_seg < seq.size()) { hbs[cur].op_n.push_back((uint32_t)((i - prev) * (size_t)stride)); hbs[cur].op_t.push_back(0); }
                    n_seg++; prev = i;
                }
            // event j pairs segment j with the k-mer at base j even when fewer than k segments follow it: pad with
            // zero-length matches (never used as windows) so that the collector sees k matched bases for it
            const size_t real = n_seg < seq.size() ? n_seg : seq.size();
            size_t pad = seq.size() > n_seg ? seq.size() - n_seg : 0; if (pad > opt.kmer_size - 1) pad = opt.kmer_size - 1;
            for (size_t i = 0; i < pad; i++) { hbs[cur].op_n.push_back(0); hbs[cur].op_t.push_back(0); }
            seq.resize(real + pad);
            if (seq.size() < opt.kmer_size) seq.append(opt.kmer_size - seq.size(), 'N'); // not "skipped": the read still gets its ':' with -d
        }
        hbs[cur].qs.push_back((int32_t)qstart); hbs[cur].ts.push_back(0); hbs[cur].te.push_back((int32_t)seq.size());
        return true;
    };
    // ---- PAF front-end: a batch of lines is read, then every thread of a pool parses / decodes a contiguous run of them
    // into its own (reused) buffers -- BLOW5 inflate + svb-zd, ss tokenising and sequence fetch are independent per
    // read -- and the runs are concatenated into the SoA batch. The first failing line in file order decides the
    // outcome, as in the reference's one-line-at-a-time loop.
    if (is_paf) {
        unsigned nt = std::thread::hardware_concurrency(); if (nt > 16) nt = 16; if (nt < 1) nt = 1;
        struct Run { HostBatch b; std::string line_buf, seq, e2, msg; pgh::Slow5Rec rec; size_t lo = 0, hi = 0, n_ok = 0; bool bad = false;
                     std::vector<pgh::Slow5File::RawView> views; std::vector<pgh::PafRec> pafs; };
        std::vector<Run> runs(nt);
        // Uncompressed BLOW5: a read's samples are copied ONCE, from the mapped file to their place in the batch. Pass 1 (per run) parses
        // the lines and looks the records up -- their lengths give every run its place; pass 2 copies the samples there and does the rest
        // (ss tokens, sequence). Otherwise a run decodes into its own buffers, which are copied into the batch (two more passes over the
        // signal and as many more resident pages). The lines kept are a prefix of the file order either way: everything behind the first
        // failing line is dropped, so the samples of the kept reads are a prefix of what was placed.
        const bool can_place = s5.has_raw_views();
        std::vector<std::string> lines;
        auto on_threads = [&](const std::function<void(unsigned)> &fn) {
            if (nt == 1) { fn(0); return; }
            std::vector<std::thread> pool;
            for (unsigned t = 0; t < nt; t++) pool.emplace_back(fn, t);
            for (auto &th : pool) th.join();
        };
        bool eof = false;
        while (!stop && !eof && status == EXIT_SUCCESS) {
            size_t n_lines = 0;
            const clk::time_point tp0 = clk::now();
            while (n_lines < this_batch_reads) {
                if ((got = getline(&line, &cap, paf_fp)) == -1) { eof = true; break; }
                if (n_lines == lines.size()) lines.emplace_back();
                lines[n_lines++].assign(line, (size_t)got);
            }
            if (n_lines == 0) break;
            const clk::time_point tp1 = clk::now();
            t_lines += secs(tp0, tp1);
            // the batch in flight may have completed every k-mer by now: then the reference would not have read these lines
            auto job_complete = [&]() -> bool {
                if (!whole_list || !dev.ok()) return false;
                const int32_t pr = dev.poll();
                if (pr < 0) { fprintf(stderr, "[gmove] %s\n", dev.error()); status = EXIT_FAILURE; return true; }
                return pr == 1 && dev.all_full_settled();
            };
            if (job_complete()) { if (status == EXIT_SUCCESS) stop = true; break; }
            // the rest of a line once its record is at hand: ss tokens, sequence, the per-read scalars
            auto finish_line = [&](Run &r, const pgh::PafRec &paf, double dig, double off, double range) -> bool {
                    if (!pgh::tokenize_ss(paf.ss, paf.ss_len, r.b.op_n, r.b.op_t, r.e2)) { r.bad = true; r.msg = r.e2; return false; }
                    // faidx_fetch_seq(m_fai, tid, st_k, end_k-1, &len) with st_k/end_k = min/max of the target columns (gmove.cpp:792-805)
                    const int64_t a = paf.target_start, b2 = paf.target_end;
                    const int64_t st_k = (uint64_t)a > (uint64_t)b2 ? b2 : a, end_k = (uint64_t)a > (uint64_t)b2 ? a : b2;
                    fai.fetch(paf.tid, (int)st_k, (int)(end_k - 1), r.seq); // absent name: empty sequence -> the read is skipped on the device
                    r.b.seq.insert(r.b.seq.end(), r.seq.begin(), r.seq.end()); r.b.seq_off.push_back(r.b.seq.size());
                    r.b.op_off.push_back(r.b.op_n.size());
                    r.b.dig.push_back(dig); r.b.off.push_back(off); r.b.range.push_back(range);
                    r.b.qs.push_back(paf.query_start); r.b.ts.push_back(paf.target_start); r.b.te.push_back(paf.target_end);
                    return true;
            };
            bool placing = can_place;
            auto pass1 = [&](unsigned t) {
                Run &r = runs[t];
                r.b.clear(); r.bad = false; r.msg.clear(); r.views.clear(); r.pafs.clear();
                r.lo = n_lines * t / nt; r.hi = n_lines * (t + 1) / nt; r.n_ok = 0;
                for (size_t i = r.lo; i < r.hi; i++) {
                    pgh::PafRec paf;
                    const int pr = pgh::parse_paf_line(&lines[i][0], lines[i].size(), paf);
                    if (pr == 1) { r.bad = true; r.msg = "malformed PAF record (fewer than 12 columns)"; return; }
                    if (pr == 2) { r.bad = true; r.msg = "ss:Z: tag not found in paf record for " + paf.rid; return; }    // gmove.cpp:1046-1049
                    if (placing) {
                        pgh::Slow5File::RawView v;
                        if (!s5.raw_view(paf.rid, v, r.e2)) { r.bad = true; r.msg = "Error in when fetching the read (" + r.e2 + ")"; return; } // gmove.cpp:745-749
                        r.b.sig_off.push_back(r.b.sig_off.back() + v.n);
                        r.views.push_back(v); r.pafs.push_back(std::move(paf));
                        r.n_ok++;
                        continue;
                    }
                    if (!s5.get(paf.rid, r.rec, r.e2)) { r.bad = true; r.msg = "Error in when fetching the read (" + r.e2 + ")"; return; } // gmove.cpp:745-749
                    if (!finish_line(r, paf, r.rec.digitisation, r.rec.offset, r.rec.range)) return;
                    if (i == r.lo) r.b.sig.reserve((r.hi - r.lo) * (r.rec.raw.size() + r.rec.raw.size() / 8)); // reads of a run are of similar length
                    r.b.sig.append(r.rec.raw.data(), r.rec.raw.data() + r.rec.raw.size());
                    r.b.sig_off.push_back(r.b.sig.size());
                    r.n_ok++;
                }
            };
            on_threads(pass1);
            // runs in file order up to the first failing line; one device batch unless that would exceed 2^29 samples
            unsigned last_run = nt; // first run that stopped early
            uint64_t tot = 0;
            auto kept_runs = [&]() {
                last_run = nt;
                for (unsigned t = 0; t < nt; t++) if (runs[t].bad) { last_run = t; break; }
                const unsigned k = last_run < nt ? last_run + 1 : nt;
                tot = 0;
                for (unsigned t = 0; t < k; t++) tot += runs[t].b.sig_off.back();
                return k;
            };
            unsigned n_runs = kept_runs();
            if (placing && tot > batch_samples_cap) { placing = false; on_threads(pass1); n_runs = kept_runs(); } // several device batches: the runs keep their own samples
            if (placing) { // pass 2: every run's samples to their place, then the rest of its lines
                hbs[cur].sig.resize(tot);
                std::vector<uint64_t> base(nt + 1, 0);
                for (unsigned t = 0; t < n_runs; t++) base[t + 1] = base[t] + runs[t].b.sig_off.back();
                on_threads([&](unsigned t) {
                    if (t >= n_runs) return;
                    Run &r = runs[t];
                    int16_t *dst = hbs[cur].sig.data() + base[t];
                    for (size_t k = 0; k < r.n_ok; k++) {
                        const pgh::Slow5File::RawView &v = r.views[k];
                        if (v.n) memcpy(dst + r.b.sig_off[k], v.samples, v.n * sizeof(int16_t));
                        if (!finish_line(r, r.pafs[k], v.digitisation, v.offset, v.range)) { // "Bad ss": the run ends in front of this line
                            r.n_ok = k; r.b.sig_off.resize(k + 1);
                            return;
                        }
                    }
                });
                n_runs = kept_runs();
            }
            t_decode += secs(tp1, clk::now());
            if (job_complete()) { if (status == EXIT_SUCCESS) stop = true; break; }
            unsigned t0 = 0;
            while (t0 < n_runs && !stop && status == EXIT_SUCCESS) {
                const unsigned t1 = tot > batch_samples_cap ? t0 + 1 : n_runs; // too big for one batch: run by run
                uint64_t ns = 0, nq = 0, no = 0, nb = 0;
                for (unsigned t = t0; t < t1; t++) { ns += runs[t].b.sig_off.back(); nq += runs[t].b.seq_off.back(); no += runs[t].b.op_off.back(); nb += runs[t].b.n(); }
                const clk::time_point tc0 = clk::now();
                hbs[cur].sig.resize(ns); hbs[cur].seq.resize(nq); hbs[cur].op_n.resize(no); hbs[cur].op_t.resize(no);
                hbs[cur].sig_off.resize(nb + 1); hbs[cur].seq_off.resize(nb + 1); hbs[cur].op_off.resize(nb + 1);
                hbs[cur].dig.resize(nb); hbs[cur].off.resize(nb); hbs[cur].range.resize(nb); hbs[cur].qs.resize(nb); hbs[cur].ts.resize(nb); hbs[cur].te.resize(nb);
                std::vector<uint64_t> bs(nt + 1, 0), bq(nt + 1, 0), bo(nt + 1, 0), bn(nt + 1, 0);
                for (unsigned t = t0; t < t1; t++) { bs[t + 1] = bs[t] + runs[t].b.sig_off.back(); bq[t + 1] = bq[t] + runs[t].b.seq_off.back(); bo[t + 1] = bo[t] + runs[t].b.op_off.back(); bn[t + 1] = bn[t] + runs[t].b.n(); }
                on_threads([&](unsigned t) {
                    if (t < t0 || t >= t1) return;
                    const HostBatch &b = runs[t].b;
                    const size_t n = b.n();
                    if (!placing && b.sig_off.back()) memcpy(hbs[cur].sig.data() + bs[t], b.sig.data(), b.sig_off.back() * sizeof(int16_t)); // (placed: already there)
                    if (b.seq_off.back()) memcpy(hbs[cur].seq.data() + bq[t], b.seq.data(), b.seq_off.back());
                    if (b.op_off.back()) { memcpy(hbs[cur].op_n.data() + bo[t], b.op_n.data(), b.op_off.back() * sizeof(uint32_t)); memcpy(hbs[cur].op_t.data() + bo[t], b.op_t.data(), b.op_off.back()); } // a failed line may have left ops behind op_off.back()
                    if (std::any_of(b.op_t.begin(), b.op_t.begin() + (ptrdiff_t)b.op_off.back(), [](uint8_t x) { return x != 0; })) batch_all_matches = false;
                    for (size_t k = 0; k < n; k++) {
                        const size_t g = bn[t] + k;
                        hbs[cur].sig_off[g] = bs[t] + b.sig_off[k]; hbs[cur].seq_off[g] = bq[t] + b.seq_off[k]; hbs[cur].op_off[g] = bo[t] + b.op_off[k];
                        hbs[cur].dig[g] = b.dig[k]; hbs[cur].off[g] = b.off[k]; hbs[cur].range[g] = b.range[k]; hbs[cur].qs[g] = b.qs[k]; hbs[cur].ts[g] = b.ts[k]; hbs[cur].te[g] = b.te[k];
                    }
                });
                hbs[cur].sig_off[nb] = ns; hbs[cur].seq_off[nb] = nq; hbs[cur].op_off[nb] = no;
                t_concat += secs(tc0, clk::now());
                total_samples += ns;
                for (uint64_t k = 0; k < nb; k++) if (++count_reads % 10000 == 0) fprintf(stderr, "*"); // PROGRESS_BATCH_SIZE
                if (nb && !flush()) { status = EXIT_FAILURE; break; }
                t0 = t1;
            }
            if (status == EXIT_SUCCESS && !stop && last_run < nt) {
                // a line the reference may never have read: it stops once every k-mer of the whole list is complete. Wait for the batches
                // in flight and look before failing.
                bool complete = false;
                if (whole_list && dev.ok()) {
                    if (dev.sync() != PG_OK) { fprintf(stderr, "[gmove] %s\n", dev.error()); status = EXIT_FAILURE; }
                    else complete = dev.all_full();
                }
                if (complete) stop = true;
                else if (status == EXIT_SUCCESS) { fprintf(stderr, "%s\n", runs[last_run].msg.c_str()); status = EXIT_FAILURE; }
            }
        }
    } else
    for (;;) {
        if (stop) break;
        pgh::MoveRec mrec;
        if (is_bam) {
            const int nr = sam.next(mrec, err);
            if (nr == 0) break;
            if (nr < 0) { fprintf(stderr, "[gmove] %s\n", err.c_str()); status = EXIT_FAILURE; break; }
            if (!mrec.has_ns) { fprintf(stderr, "tag 'ns' is not found. Please check your SAM/BAM file: \n"); status = EXIT_FAILURE; break; }   // gmove.cpp:1086-1089
            if (!mrec.has_ts) { fprintf(stderr, "tag 'ts' is not found. Please check your SAM/BAM file: \n"); status = EXIT_FAILURE; break; }   // gmove.cpp:1094-1097
            if (!mrec.has_mv) { fprintf(stderr, "NULL returned for tag mv: \n"); status = EXIT_FAILURE; break; }                               // gmove.cpp:1102-1105
            if (!mrec.mv_is_Bc) { fprintf(stderr, "tag 'mv' specification is incorrect\n"); status = EXIT_FAILURE; break; }                    // gmove.cpp:1120-1123
            is_one.swap(mrec.is_one);
            if (!add_move_record(mrec.qname.c_str(), (int)mrec.seq.size(), mrec.seq, mrec.stride, mrec.ns, (long long)mrec.ts)) break;
        } else if ((got = getline(&line, &cap, paf_fp)) == -1) break;
        else {
            // move table (gmove.cpp:557-700): read_id, fastq_len, fastq_seq, stride, moves, signal_len, trim_offset
            char *col[7]; int nc = 0;
            for (char *p = line, *e = line + got; p < e && nc < 7;) {
                char *t = (char *)memchr(p, '\t', (size_t)(e - p)); if (!t) t = e;
                col[nc++] = p; *t = 0; p = t + 1;
            }
            if (nc < 7) { fprintf(stderr, "malformed move-table record (fewer than 7 columns)\n"); status = EXIT_FAILURE; break; }
            const char *moves = col[4]; const size_t move_len = strlen(moves);
            is_one.resize(move_len);
            for (size_t i = 0; i < move_len; i++) is_one[i] = moves[i] == '1';
            if (!add_move_record(col[0], atoi(col[1]), col[2], atoi(col[3]), strtoull(col[5], nullptr, 10), atoll(col[6]))) break;
        }
        hbs[cur].sig_off.push_back(hbs[cur].sig.size());
        hbs[cur].dig.push_back(rec.digitisation); hbs[cur].off.push_back(rec.offset); hbs[cur].range.push_back(rec.range);
        hbs[cur].seq.insert(hbs[cur].seq.end(), seq.begin(), seq.end()); hbs[cur].seq_off.push_back(hbs[cur].seq.size());
        hbs[cur].op_off.push_back(hbs[cur].op_n.size());
        if (++count_reads % 10000 == 0) fprintf(stderr, "*"); // PROGRESS_BATCH_SIZE
        if (hbs[cur].n() >= this_batch_reads || hbs[cur].sig.size() >= batch_samples_cap) { if (!flush()) { status = EXIT_FAILURE; flush_failed = true; break; } }
    }
    if (status == EXIT_FAILURE && !is_paf && !flush_failed && whole_list) {
        // A RECORD error (a failed submit / device error is never rescued): the reference may never have read that record -- it stops once
        // every k-mer of the whole list is complete (gmove.cpp:733-735), and the reads in front of the bad record, queued or still in
        // the unflushed batch, may do that. As the PAF loop above: drop what the bad record left half-appended, submit the valid reads in
        // front of it, wait for the batches in flight, and look before failing.
        HostBatch &h = hbs[cur];
        h.sig.resize((size_t)h.sig_off.back()); h.seq.resize((size_t)h.seq_off.back());
        h.op_n.resize((size_t)h.op_off.back()); h.op_t.resize((size_t)h.op_off.back());
        h.qs.resize(h.n()); h.ts.resize(h.n()); h.te.resize(h.n());
        if (flush() && dev.ok() && dev.sync() == PG_OK && dev.all_full()) {
            status = EXIT_SUCCESS; stop = true;
            fprintf(stderr, "[gmove] the record reported above lies behind the read that completes the last k-mer: the reference stops reading there (gmove.cpp:733-735); ignored\n");
        }
    }
    if (status == EXIT_SUCCESS && !stop && !flush()) status = EXIT_FAILURE;
    free(line); fclose(paf_fp);

    const double t_loop = secs(t_loop0, clk::now());
    if (status == EXIT_SUCCESS) {
        pg_result res;
        const clk::time_point tq0 = clk::now();
        const pg_status fin = need_ctx() ? dev.finish(&res) : PG_ERR_NO_DEVICE;
        t_finish = secs(tq0, clk::now());
        if (fin != PG_OK) { if (dev.ok()) fprintf(stderr, "[gmove] %s\n", dev.error()); status = EXIT_FAILURE; }
        else {
            pgh::DumpInput in{res.n_slots, res.counts, res.ev_off, res.samp_off, res.ev_len, res.ev_read, res.samples, res.read_skipped, res.n_reads,
                              [&](uint64_t first, uint64_t n, double *dst) { return dev.fetch(first, n, dst); }};
            unsigned nt = std::thread::hardware_concurrency(); if (nt > 16) nt = 16;
            const clk::time_point td0 = clk::now();
            // the "%.8f" text itself comes from the device (pg_text / pg_job_text) when it can: no -d (the ':' of -d depend on the reads),
            // every sample inside the fixed-point formatter's range. POREGEN_HOST_TEXT=1 keeps the host formatter (A/B, tests).
            pg_text_result tx;
            bool wrote = false;
            if (!opt.delimit_files && !getenv("POREGEN_HOST_TEXT") && dev.text(&tx) == PG_OK) {
                if (getenv("POREGEN_DUMP_PROBE")) fprintf(stderr, "[dump probe] pg_text: %.3f s for %.1f MB of text\n", secs(td0, clk::now()), tx.n_bytes / 1e6);
                pgh::TextInput ti{tx.n_slots, tx.slot_off, res.counts, [&](uint64_t first, uint64_t n, char *dst) { return dev.fetch_text(first, n, dst); }};
                if (!pgh::write_dump_dir_text(output_dir, slot_kmers, ti, nt, err)) { fprintf(stderr, "%s\n", err.c_str()); status = EXIT_FAILURE; }
                wrote = true;
            }
            if (!wrote && !pgh::write_dump_dir(output_dir, slot_kmers, in, opt.delimit_files != 0, opt.sample_limit, nt, err)) { fprintf(stderr, "%s\n", err.c_str()); status = EXIT_FAILURE; }
            t_dump = secs(td0, clk::now());
            if (status == EXIT_SUCCESS && (raw_model_path || dwell_model_path)) { // scripts/poregen.sh:54-85, 33-52 without the text round trip
                const clk::time_point tm0 = clk::now();
                pg_model_result mr;
                if (dev.model(&mr) != PG_OK) { fprintf(stderr, "[gmove] %s\n", dev.error()); status = EXIT_FAILURE; }
                else {
                    std::vector<uint32_t> order(res.n_slots); // the shell glob lists the dump files sorted by name
                    for (uint32_t i = 0; i < res.n_slots; i++) order[i] = i;
                    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return slot_kmers[a] < slot_kmers[b]; });
                    char a[64], b[64];
                    if (raw_model_path) {
                        FILE *fp = fopen(raw_model_path, "w");
                        if (!fp) { fprintf(stderr, "Could not open %s for writing.\n", raw_model_path); status = EXIT_FAILURE; }
                        else {
                            const long double lim = strtold(stdv_limit, nullptr);
                            for (uint32_t i : order) {
                                pg_model_format(&mr, i, PG_MODEL_TEXT_MEDIAN, a, sizeof a); pg_model_format(&mr, i, PG_MODEL_TEXT_SSTDEV, b, sizeof b);
                                const bool capped = b[0] && strcmp(b, "nan") != 0 && strtold(b, nullptr) > lim; // bc -l: "$stddev > $limit"
                                fprintf(fp, "%s\t%s\t%s\n", slot_kmers[i].c_str(), a, capped ? stdv_limit : b);
                            }
                            fclose(fp);
                        }
                    }
                    if (dwell_model_path) {
                        FILE *fp = fopen(dwell_model_path, "a"); // the script appends to this file
                        if (!fp) { fprintf(stderr, "Could not open %s for writing.\n", dwell_model_path); status = EXIT_FAILURE; }
                        else {
                            for (uint32_t i : order) { pg_model_format(&mr, i, PG_MODEL_TEXT_DWELL, a, sizeof a); fprintf(fp, "%s\t%s\n", slot_kmers[i].c_str(), a); }
                            fclose(fp);
                        }
                    }
                }
                fprintf(stderr, "\n[gmove] time: k-mer model on the device %.3f s", secs(tm0, clk::now()));
            }
            fprintf(stderr, "\n[gmove] time: file indices %.3f s next to device context %.3f s (own thread), waited %.3f s for it at the first batch\n", secs(t_setup0, t_setup1), t_ctx, t_ctx_wait);
            if (is_paf) fprintf(stderr, "[gmove] time: PAF lines %.3f s, parse + decode on the pool %.3f s, one batch from the runs %.3f s\n", t_lines, t_decode, t_concat);
            fprintf(stderr, "[gmove] time: reading + parsing %.3f s, staging + device %.3f s, download + merge %.3f s, dump files %.3f s\n",
                    t_loop - t_device, t_device, t_finish, t_dump);
            std::string where = "device " + std::to_string(device);
            if (dev.job) { where = "devices"; for (size_t i = 0; i < devices.size(); i++) where += (i ? "," : " ") + std::to_string(devices[i]); where += pg_job_uses_rccl(dev.job) ? " (RCCL all-gather)" : " (exchange through host memory)"; }
            fprintf(stderr, "[gmove] %llu reads, %llu samples, %llu events kept (%llu samples) on %s\n", (unsigned long long)res.n_reads,
                    (unsigned long long)total_samples, (unsigned long long)res.n_events, (unsigned long long)res.n_samples, where.c_str());
        }
    }
    (void)need_ctx(); // a failed run may not have reached the first use
    const clk::time_point t_end0 = clk::now();
    if (status != EXIT_SUCCESS || getenv("POREGEN_CLEAN_EXIT")) dev.destroy(); // a successful run leaves its (idle) device to the exit of the process: main.cpp
    fprintf(stderr, "[gmove] time: %.3f s from the start of gmove to the end of the output, %.3f s to release the device\n", secs(t_main0, t_end0), secs(t_end0, clk::now()));
    return status;
}
