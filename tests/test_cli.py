"""`poregen gmove` (bin/poregen): argument handling on any box; on the GPU box whole output directories are
diffed byte for byte against the CPU oracle's CLI on the reference fixtures and on synthetic SLOW5+PAF+FASTQ."""
import filecmp
import os
import subprocess

import numpy as np
import pytest

import orc
from poregen_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "bin", "poregen")
G = os.path.join(ROOT, "tests", "golden", "single_read")


def cli(args, **kw):
    return subprocess.run([BIN, "gmove"] + [str(a) for a in args], capture_output=True, text=True, **kw)


def oracle_cli(args):
    return subprocess.run([orc.CLI] + [str(a) for a in args], capture_output=True, text=True)


def assert_same_dirs(a, b):
    assert open(os.path.join(a, "freq.txt")).read() == open(os.path.join(b, "freq.txt")).read()
    cmp = filecmp.dircmp(os.path.join(a, "dump"), os.path.join(b, "dump"))
    assert not cmp.left_only and not cmp.right_only
    _, mismatch, errors = filecmp.cmpfiles(os.path.join(a, "dump"), os.path.join(b, "dump"), cmp.common_files, shallow=False)
    assert not mismatch and not errors, mismatch[:5]


# ---- any box ------------------------------------------------------------------------------------------

def test_usage_and_exit_codes(tmp_path):
    assert cli([]).returncode == 1                                                                   # test_gmove.sh 0.1
    r = cli(["--help"]); assert r.returncode == 0 and "Usage: poregen gmove" in r.stdout
    assert cli([f"{G}/reads.slow5", f"{G}/guppy_move", "--kmer_file", f"{G}/kmer_file.txt", tmp_path / "a"]).returncode == 1   # 0.3 (k=9 vs 6-mers)
    r = cli([f"{G}/reads.slow5", f"{G}/guppy_move.paf", "--file_limit", "50", tmp_path / "b"])       # 0.5
    assert r.returncode == 1 and ".paf input requires an additional .fastq file" in r.stderr
    d = tmp_path / "d"; d.mkdir(); (d / "x").write_text("x")
    r = cli([f"{G}/reads.slow5", f"{G}/guppy_move.paf", "--fastq", f"{G}/read_0.fastq", d])
    assert r.returncode == 1 and "is not empty" in r.stderr
    assert cli(["-k", "0", "a", "b", "c"]).returncode == 1
    assert cli(["--scaling", "2", f"{G}/reads.slow5", f"{G}/guppy_move.paf", "--fastq", f"{G}/read_0.fastq", tmp_path / "e"]).returncode == 1
    r = subprocess.run([BIN, "--version"], capture_output=True, text=True); assert r.returncode == 0


def _has_gpu():
    import torch
    return torch.cuda.is_available()


@pytest.mark.skipif(_has_gpu(), reason="only meaningful on a box without a GPU")
def test_cli_fails_loudly_without_gpu(tmp_path):
    r = cli(["-k", "6", f"{G}/reads.slow5", f"{G}/guppy_move.paf", tmp_path / "o", "--fastq", f"{G}/read_0.fastq"])
    assert r.returncode == 1 and "no CPU fallback" in r.stderr
    # the output layout is created before the device is needed, exactly like the reference creates it up front
    assert os.path.isdir(tmp_path / "o" / "dump")


# ---- GPU box ------------------------------------------------------------------------------------------

FIXTURE_CASES = {
    "ka1": ["-k", "6", "{G}/reads.slow5", "{G}/guppy_move.paf", "{OUT}", "--fastq", "{G}/read_0.fastq", "--kmer_file", "{G}/kmer_file.txt"],
    "ka2_scaling": ["-k", "6", "{G}/reads.slow5", "{G}/guppy_move.paf", "{OUT}", "--fastq", "{G}/read_0.fastq", "--kmer_file", "{G}/kmer_file.txt", "--scaling", "1"],
    "ka3_pamin": ["-k", "6", "{G}/reads.slow5", "{G}/guppy_move.paf", "{OUT}", "--fastq", "{G}/read_0.fastq", "--kmer_file", "{G}/kmer_file.txt", "--scaling", "1", "--pa_min", "100"],
    "t12_literal": ["-k", "6", "{G}/reads.slow5", "{G}/guppy_move.paf", "{OUT}", "--fastq", "{G}/read_0.fastq", "--kmer_file", "{G}/single_kmer_file.txt"],  # test_gmove.sh 1.2 as written (ka1 is 2.2, ka6_defaults 0.6)
    "ka5_delimit": ["-k", "6", "-d", "{G}/reads.slow5", "{G}/guppy_move.paf", "{OUT}", "--fastq", "{G}/read_0.fastq", "--kmer_file", "{G}/single_kmer_file.txt"],
    "ka6_defaults": ["{G}/reads.slow5", "{G}/guppy_move.paf", "--file_limit", "50", "{OUT}", "--fastq", "{G}/read_0.fastq"],
    "margin0_k5_all": ["-k", "5", "{G}/reads.slow5", "{G}/guppy_move.paf", "{OUT}", "--fastq", "{G}/read_0.fastq", "--file_limit", "5000", "--kmer_pick_margin", "0", "--scaling", "1"],
    "print_margin2": ["-k", "5", "{G}/reads.slow5", "{G}/guppy_move.paf", "{OUT}", "--fastq", "{G}/read_0.fastq", "--file_limit", "5000", "--kmer_pick_margin", "1", "--scaling", "1", "--margin", "2"],
    "slice": ["-k", "6", "{G}/reads.slow5", "{G}/guppy_move.paf", "{OUT}", "--fastq", "{G}/read_0.fastq", "--kmer_file", "{G}/kmer_file.txt", "--index_start", "10", "--index_end", "40", "-d"],
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(FIXTURE_CASES))
def test_fixture_dirs_equal_oracle(tmp_path, name):
    a, b = tmp_path / "gpu", tmp_path / "cpu"
    args = FIXTURE_CASES[name]
    r = cli([x.replace("{G}", G).replace("{OUT}", str(a)) for x in args]); assert r.returncode == 0, r.stderr
    o = oracle_cli([x.replace("{G}", G).replace("{OUT}", str(b)) for x in args]); assert o.returncode == 0, o.stderr
    assert_same_dirs(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,extra", [
    ("rna004", ["-k", "5", "--rna", "--scaling", "1", "--min_dur", "20", "--max_dur", "40", "--file_limit", "1024", "--sample_limit", "30"]),
    ("rna004", ["-k", "5", "--rna", "--scaling", "1", "--min_dur", "20", "--max_dur", "40", "--file_limit", "1024", "--sample_limit", "4", "-d", "--batch_reads", "37"]),
    ("dna_r10", ["-k", "6", "--scaling", "1", "--file_limit", "4096", "--sample_limit", "12", "--kmer_pick_margin", "1"]),
    ("dna_r10", ["-k", "9", "--scaling", "1", "--index_start", "1000", "--index_end", "3000", "--sample_limit", "5", "--batch_reads", "50"]),
])
def test_synthetic_files_equal_oracle(tmp_path, kind, extra):
    b = synth.make_batch(160, kind=kind, seed=77, indel_rate=0.02)
    pre = str(tmp_path / "syn")
    synth.write_files(b, pre)
    common = [pre + ".slow5", pre + ".paf", "--fastq", pre + ".fastq"] + extra
    r = cli(common + [tmp_path / "gpu"]); assert r.returncode == 0, r.stderr
    o = oracle_cli([x for x in common if x not in ("--batch_reads", "37", "50")] + [tmp_path / "cpu"]); assert o.returncode == 0, o.stderr
    assert_same_dirs(tmp_path / "gpu", tmp_path / "cpu")


TABLE_CASES = {
    "t02_defaults": ["{G}/reads.slow5", "{G}/guppy_move", "--file_limit", "50", "{OUT}"],                                         # test_gmove.sh 0.2
    "t04_kmer_file": ["-k", "6", "-m", "0", "{G}/reads.slow5", "{G}/guppy_move", "--kmer_file", "{G}/kmer_file.txt", "{OUT}"],   # 0.4 / 2.1
    "t11_single": ["-k", "6", "-m", "0", "{G}/reads.slow5", "{G}/guppy_move", "--kmer_file", "{G}/single_kmer_file.txt", "{OUT}"],  # 1.1 (KA-4)
    "t_k5_all_scaled": ["-k", "5", "{G}/reads.slow5", "{G}/guppy_move", "--file_limit", "5000", "--scaling", "1", "-d", "{OUT}"],
    "t_offsets": ["-k", "5", "-m", "2", "-s", "3", "{G}/reads.slow5", "{G}/guppy_move", "--file_limit", "5000", "--scaling", "1", "--margin", "4", "{OUT}"],
    "t_k9_slice": ["{G}/reads.slow5", "{G}/guppy_move", "--index_start", "50000", "--index_end", "60000", "--scaling", "1", "{OUT}"],
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(TABLE_CASES))
def test_table_front_end_fixture_equals_oracle(tmp_path, name):
    """Move-table front-end (src/gmove.cpp:539-706) on the reference's fixture; t11_single is KA-4."""
    a, b = tmp_path / "gpu", tmp_path / "cpu"
    args = TABLE_CASES[name]
    r = cli([x.replace("{G}", G).replace("{OUT}", str(a)) for x in args]); assert r.returncode == 0, r.stderr
    o = oracle_cli([x.replace("{G}", G).replace("{OUT}", str(b)) for x in args]); assert o.returncode == 0, o.stderr
    assert_same_dirs(a, b)
    if name == "t11_single":
        assert open(a / "dump" / "ATGTTG").read() == "143.00816337,118.41790281,115.53301191,119.24215736,114.43400585,117.18152100,113.33499979,115.53301191,128.44633310,162.51552091;"


@pytest.mark.gpu
def test_reference_invariant_table_equals_paf_on_gpu(tmp_path):
    """test_gmove.sh 2.1 == 2.2 through the product (valid at --kmer_pick_margin 0)."""
    t, p = tmp_path / "t", tmp_path / "p"
    assert cli(["-k", "6", "-m", "0", f"{G}/reads.slow5", f"{G}/guppy_move", "--kmer_file", f"{G}/kmer_file.txt", t]).returncode == 0
    assert cli(["-k", "6", f"{G}/reads.slow5", f"{G}/guppy_move.paf", "--fastq", f"{G}/read_0.fastq", "--kmer_file", f"{G}/kmer_file.txt",
                "--kmer_pick_margin", "0", p]).returncode == 0
    assert_same_dirs(t, p)


@pytest.mark.gpu
@pytest.mark.parametrize("trim,extra", [(0, ["-k", "5", "--file_limit", "1024", "--scaling", "1", "--sample_limit", "25"]),
                                        (37, ["-k", "6", "--file_limit", "4096", "--scaling", "1", "--sample_limit", "7", "-d", "-m", "1", "--batch_reads", "33"])])
def test_table_front_end_synthetic_equals_oracle(tmp_path, trim, extra):
    b = synth.make_batch(120, kind="dna_r10", seed=79)
    pre = str(tmp_path / "syn")
    synth.write_table_files(b, pre, trim=trim)
    common = [pre + ".slow5", pre + ".table"] + extra
    r = cli(common + [tmp_path / "gpu"]); assert r.returncode == 0, r.stderr
    o = oracle_cli([x for x in common if x not in ("--batch_reads", "33")] + [tmp_path / "cpu"]); assert o.returncode == 0, o.stderr
    assert_same_dirs(tmp_path / "gpu", tmp_path / "cpu")


@pytest.mark.gpu
def test_records_with_fewer_moves_than_k_have_no_events(tmp_path):
    """A move-table / SAM record whose basecall is long enough (>= 10 bases) but whose move string holds a single '1' closes no
    segment: k - 1 padding ops, fewer than k. The batch is then NOT one of whole-k matches only (PG_BATCH_ALL_MATCHES must not be
    claimed): the read simply has no events, as on the reference's table path (gmove.cpp:616-700), and the run succeeds. A PAF
    record whose ss string holds more matches than bases were fetched is undefined in the reference: both CLIs refuse it."""
    b = synth.make_batch(40, kind="dna_r10", seed=83)
    pre = str(tmp_path / "syn")
    synth.write_table_files(b, pre)
    rows = open(pre + ".table").read().split("\n")
    c = rows[7].split("\t"); c[4] = "1" + "0" * (len(c[4]) - 1); rows[7] = "\t".join(c)
    open(pre + ".table", "w").write("\n".join(rows))
    sam = open(pre + ".sam").read().split("\n")
    i7 = [i for i, ln in enumerate(sam) if ln.startswith("r7\t")][0]
    c = sam[i7].split("\t"); mv = c[11].split(","); c[11] = ",".join(mv[:3] + ["1"] + ["0"] * (len(mv) - 4)); sam[i7] = "\t".join(c)
    open(pre + ".sam", "w").write("\n".join(sam))
    extra = ["-k", "5", "--file_limit", "1024", "--scaling", "1", "--sample_limit", "9", "-d"]
    for ext in (".table", ".sam"):
        r = cli([pre + ".slow5", pre + ext] + extra + [tmp_path / ("gpu" + ext)]); assert r.returncode == 0, r.stderr
        o = oracle_cli([pre + ".slow5", pre + ext] + extra + [tmp_path / ("cpu" + ext)]); assert o.returncode == 0, o.stderr
        assert_same_dirs(tmp_path / ("gpu" + ext), tmp_path / ("cpu" + ext))
    # PAF: one match op more than fetched bases
    synth.write_files(b, pre + "p")
    lines = open(pre + "p.paf").read().split("\n")
    cols = lines[5].split("\t")
    ss = [i for i, x in enumerate(cols) if x.startswith("ss:Z:")][0]
    cols[ss] = cols[ss] + "5,"
    lines[5] = "\t".join(cols)
    open(pre + "p.paf", "w").write("\n".join(lines))
    pa = [pre + "p.slow5", pre + "p.paf", "--fastq", pre + "p.fastq", "-k", "5", "--file_limit", "1024", "--scaling", "1"]
    assert cli(pa + [tmp_path / "gpu_p"]).returncode == 1
    assert oracle_cli(pa + [tmp_path / "cpu_p"]).returncode != 0


@pytest.mark.gpu
@pytest.mark.parametrize("kf", ["single_kmer_file.txt", "kmer_file.txt"])
def test_sam_bam_front_end_fixture(tmp_path, kf):
    """SAM/BAM front-end (src/gmove.cpp:1061-1266): test_gmove.sh 1.3 / 2.3 -- BAM == table -- plus BAM == SAM == oracle(SAM)."""
    outs = {}
    for name, args in {"bam": ["-k", "6", f"{G}/reads.slow5", f"{G}/guppy_move.bam"], "sam": ["-k", "6", f"{G}/reads.slow5", f"{G}/guppy_move.sam"],
                       "table": ["-k", "6", "-m", "0", f"{G}/reads.slow5", f"{G}/guppy_move"]}.items():
        outs[name] = tmp_path / name
        r = cli(args + ["--kmer_file", f"{G}/{kf}", outs[name]]); assert r.returncode == 0, r.stderr
    o = oracle_cli(["-k", "6", f"{G}/reads.slow5", f"{G}/guppy_move.sam", "--kmer_file", f"{G}/{kf}", tmp_path / "cpu"]); assert o.returncode == 0, o.stderr
    assert_same_dirs(outs["bam"], outs["table"])      # the reference's own invariant
    assert_same_dirs(outs["bam"], outs["sam"])
    assert_same_dirs(outs["sam"], tmp_path / "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [["-k", "5", "--file_limit", "1024", "--scaling", "1", "--sample_limit", "25"],
                                   ["-k", "6", "--file_limit", "4096", "--sample_limit", "9", "-d", "-m", "1", "--batch_reads", "29"],
                                   ["-k", "5", "--file_limit", "1024", "--scaling", "1", "--pa_min", "60", "--pa_max", "170", "-d"]])
def test_sam_front_end_synthetic_skips_out_of_range_reads(tmp_path, extra):
    """On the SAM/BAM path a read with ANY out-of-range sample is skipped (gmove.cpp:1149-1160): with 0.05 % spikes about
    one read in eight survives, which also makes event acceptance signal-dependent."""
    b = synth.make_batch(160, kind="dna_r10", seed=80, spike_rate=0.0005)
    pre = str(tmp_path / "syn")
    synth.write_table_files(b, pre, trim=11)
    common = [pre + ".slow5", pre + ".sam"] + extra
    r = cli(common + [tmp_path / "gpu"]); assert r.returncode == 0, r.stderr
    o = oracle_cli([x for x in common if x not in ("--batch_reads", "29")] + [tmp_path / "cpu"]); assert o.returncode == 0, o.stderr
    assert_same_dirs(tmp_path / "gpu", tmp_path / "cpu")
    freq = [int(l.split()[1]) for l in open(tmp_path / "cpu" / "freq.txt")]
    assert 0 < sum(freq)     # some reads survive, and (below) not all of them
    t = cli([pre + ".slow5", pre + ".table"] + [x for x in extra if x not in ("--batch_reads", "29")] + [tmp_path / "tab"]); assert t.returncode == 0
    assert open(tmp_path / "tab" / "freq.txt").read() != open(tmp_path / "gpu" / "freq.txt").read()   # the table path zero-fills instead


@pytest.mark.gpu
def test_sam_missing_tags_exit_1(tmp_path):
    p = tmp_path / "x.sam"
    p.write_text("r0\t4\t*\t0\t0\t*\t*\t0\t0\tACGTACGTACGT\t*\tmv:B:c,5,1,0,1\tts:i:0\n")
    r = cli([f"{G}/reads.slow5", p, tmp_path / "o"]); assert r.returncode == 1 and "tag 'ns' is not found" in r.stderr


@pytest.mark.gpu
def test_print_margin_larger_than_window_start_is_rejected(tmp_path):
    """--margin > start of an accepted window is undefined behaviour in the reference (unsigned wrap at
    src/gmove.cpp:928-932); the oracle flags it (exit 70) and the product refuses it (exit 1)."""
    args = ["-k", "5", f"{G}/reads.slow5", f"{G}/guppy_move.paf", "--fastq", f"{G}/read_0.fastq", "--file_limit", "5000", "--kmer_pick_margin", "0", "--margin", "2"]
    r = cli(args + [tmp_path / "gpu"]); assert r.returncode == 1 and "margin > start" in r.stderr
    assert oracle_cli(args + [tmp_path / "cpu"]).returncode == 70


@pytest.mark.gpu
def test_rna_record_without_flag_exits_1(tmp_path):
    b = synth.make_batch(5, kind="rna004", seed=78)
    pre = str(tmp_path / "syn"); synth.write_files(b, pre)
    r = cli(["-k", "5", pre + ".slow5", pre + ".paf", "--fastq", pre + ".fastq", tmp_path / "o"])
    assert r.returncode == 1 and "allow_rna" in r.stderr     # src/gmove.cpp:795-797


@pytest.mark.gpu
def test_blow5_input_equals_ascii_input(tmp_path):
    """The reference's BLOW5 fixture (zlib + svb-zd) through the CLI == the same reads as ASCII SLOW5 through the oracle."""
    import ctypes as C
    h = C.CDLL((os.environ.get("PG_HOSTTEST_SO") or os.path.join(ROOT, "poregen_amd", "_pg_hosttest.so")))
    h.pgt_slow5_get.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_size_t]; h.pgt_slow5_get.restype = C.c_long
    b5 = os.path.join(ROOT, "tests", "golden", "blow5", "example.blow5")
    rng = np.random.default_rng(3)
    with open(tmp_path / "r.slow5", "w") as s5, open(tmp_path / "r.paf", "w") as paf, open(tmp_path / "r.fastq", "w") as fq:
        s5.write("#slow5_version\t0.2.0\n#num_read_groups\t1\n#read_id\tread_group\tdigitisation\toffset\trange\tsampling_rate\tlen_raw_signal\traw_signal\n")
        for i in range(1, 6):
            dor = np.zeros(3); raw = np.zeros(100000, np.int16)
            n = h.pgt_slow5_get(b5.encode(), f"r{i}".encode(), dor.ctypes.data, raw.ctypes.data, 100000)
            s5.write(f"r{i}\t0\t{dor[0]:.17g}\t{dor[1]:.17g}\t{dor[2]:.17g}\t4000\t{n}\t" + ",".join(map(str, raw[:n].tolist())) + "\n")
            nb = 900
            d = rng.integers(5, 60, nb); d[-1] += 0
            seq = "".join("ACGT"[x] for x in rng.integers(0, 4, nb))
            fq.write(f"@r{i}\n{seq}\n+\n{'I' * nb}\n")
            paf.write(f"r{i}\t{n}\t100\t{n}\t+\tr{i}\t{nb}\t0\t{nb}\t{nb}\t{nb}\t255\tss:Z:" + "".join(f"{x}," for x in d) + "\n")
    extra = ["-k", "5", "--scaling", "1", "--file_limit", "1024", "--sample_limit", "6", "--pa_min", "60", "--pa_max", "140"]
    r = cli([b5, tmp_path / "r.paf", "--fastq", tmp_path / "r.fastq", tmp_path / "gpu"] + extra); assert r.returncode == 0, r.stderr
    o = oracle_cli([tmp_path / "r.slow5", tmp_path / "r.paf", "--fastq", tmp_path / "r.fastq", tmp_path / "cpu"] + extra); assert o.returncode == 0, o.stderr
    assert_same_dirs(tmp_path / "gpu", tmp_path / "cpu")


@pytest.mark.gpu
def test_malformed_line_behind_the_completing_read(tmp_path):
    """Whole k-mer list: the reference stops reading once every k-mer is complete (gmove.cpp:733-735), so a malformed
    PAF line behind that read goes unnoticed -- exit 0, same directory as the oracle CLI. With a slice of the list the
    reference reads every line and exits on it: both CLIs fail."""
    b = synth.make_batch(200, read_len=3000, kind="rna004", seed=91)
    pre = str(tmp_path / "syn")
    synth.write_files(b, pre)
    lines = open(pre + ".paf").read().split("\n")
    cols = lines[180].split("\t")
    ss = [i for i, c in enumerate(cols) if c.startswith("ss:Z:")][0]
    cols[ss] = cols[ss].replace(",", "X", 1)                       # "Bad ss": exit(1) in the reference (gmove.cpp:834-867)
    lines[180] = "\t".join(cols)
    open(pre + ".paf", "w").write("\n".join(lines))
    whole = [pre + ".slow5", pre + ".paf", "--fastq", pre + ".fastq", "-k", "3", "--rna", "--scaling", "1", "--min_dur", "20", "--max_dur", "40",
             "--file_limit", "64", "--sample_limit", "5"]
    r = cli(whole + [tmp_path / "gpu", "--batch_reads", "64"]); assert r.returncode == 0, r.stderr
    r1 = cli(whole + [tmp_path / "gpu1", "--batch_reads", "1000"]); assert r1.returncode == 0, r1.stderr   # the bad line inside the completing batch
    o = oracle_cli(whole + [tmp_path / "cpu"]); assert o.returncode == 0, o.stderr
    assert_same_dirs(tmp_path / "gpu", tmp_path / "cpu"); assert_same_dirs(tmp_path / "gpu1", tmp_path / "cpu")
    sl = [x if x != "64" else "10" for x in whole]                  # --file_limit 10: a slice, the loop never ends early
    assert cli(sl + [tmp_path / "gpu2"]).returncode == 1
    assert oracle_cli(sl + [tmp_path / "cpu2"]).returncode != 0
    # the same from an uncompressed BLOW5: the samples are placed straight into the batch, run by run, and a run that ends at a bad
    # line leaves its later samples behind the kept prefix
    synth.write_blow5(b, pre + ".blow5", compress=False)
    wb = [pre + ".blow5"] + whole[1:]
    for name, br in (("gpu3", "64"), ("gpu4", "1000"), ("gpu5", "7")):
        r = cli(wb + [tmp_path / name, "--batch_reads", br]); assert r.returncode == 0, r.stderr
        assert_same_dirs(tmp_path / name, tmp_path / "cpu")
    assert cli([pre + ".blow5"] + sl[1:] + [tmp_path / "gpu6"]).returncode == 1


@pytest.mark.gpu
def test_bad_move_record_behind_the_completing_read(tmp_path):
    """The same on the move-table front-end (gmove.cpp:539-706): a malformed record behind the read that completes the whole list goes
    unnoticed by the reference. The completing read may sit in a batch that is already queued (--batch_reads 40) or in the LAST, still
    unflushed one (--batch_reads 1000: the valid reads in front of the bad record are submitted before the verdict); what the bad record
    left half-appended is dropped. With a slice of the list both CLIs fail."""
    b = synth.make_batch(150, kind="dna_r10", seed=97)
    pre = str(tmp_path / "syn")
    synth.write_table_files(b, pre)
    lines = open(pre + ".table").read().split("\n")
    cols = lines[120].split("\t")
    lines[120] = "\t".join(cols[:5])                                # fewer than 7 columns: exit in the reference
    open(pre + ".table", "w").write("\n".join(lines))
    whole = [pre + ".slow5", pre + ".table", "-k", "3", "--scaling", "1", "--file_limit", "64", "--sample_limit", "5"]
    o = oracle_cli(whole + [tmp_path / "cpu"]); assert o.returncode == 0, o.stderr
    for name, br in (("gpu", "40"), ("gpu1", "1000"), ("gpu2", "119")):
        r = cli(whole + [tmp_path / name, "--batch_reads", br]); assert r.returncode == 0, (br, r.stderr)
        assert_same_dirs(tmp_path / name, tmp_path / "cpu")
    sl = [x if x != "64" else "10" for x in whole]
    assert cli(sl + [tmp_path / "gpu3"]).returncode == 1
    assert oracle_cli(sl + [tmp_path / "cpu3"]).returncode != 0
    # a record that fails half-way through (moves shorter than -m asks for) leaves samples behind in the batch: they must not be submitted
    lines[120] = "\t".join(cols[:4] + ["1"] + cols[5:])
    open(pre + ".table", "w").write("\n".join(lines))
    r = cli(whole + ["-m", "1", tmp_path / "gpu4", "--batch_reads", "1000"])
    o = oracle_cli(whole + ["-m", "1", tmp_path / "cpu4"])
    assert (r.returncode == 0) == (o.returncode == 0), (r.stderr, o.stderr)
    if o.returncode == 0:
        assert_same_dirs(tmp_path / "gpu4", tmp_path / "cpu4")


@pytest.mark.gpu
def test_sample_limit_zero_reads_every_line(tmp_path):
    """--sample_limit 0: no k-mer ever completes (gmove.cpp:925-927 skips in front of 945-950), so the reference reads every
    line, writes ':' per read with -d, and still exits on an RNA-oriented record without --rna behind the first batch."""
    b = synth.make_batch(6, kind="dna_r10", seed=93)
    pre = str(tmp_path / "syn"); synth.write_files(b, pre)
    args = [pre + ".slow5", pre + ".paf", "--fastq", pre + ".fastq", "-k", "3", "--scaling", "1", "--file_limit", "64", "--sample_limit", "0", "-d"]
    r = cli(args + [tmp_path / "gpu", "--batch_reads", "1"]); assert r.returncode == 0, r.stderr
    o = oracle_cli(args + [tmp_path / "cpu"]); assert o.returncode == 0, o.stderr
    assert_same_dirs(tmp_path / "gpu", tmp_path / "cpu")
    assert open(tmp_path / "gpu" / "dump" / "AAA").read() == ":" * 6
    lines = open(pre + ".paf").read().split("\n")
    cols = lines[3].split("\t"); cols[7], cols[8] = cols[8], cols[7]      # target_start > target_end: RNA orientation
    lines[3] = "\t".join(cols)
    open(pre + ".paf", "w").write("\n".join(lines))
    r = cli(args + [tmp_path / "gpu2", "--batch_reads", "1"]); assert r.returncode == 1 and "allow_rna" in r.stderr
    assert oracle_cli(args + [tmp_path / "cpu2"]).returncode != 0


def test_devices_scale_the_default_batch(tmp_path):
    """`--devices` cuts a batch into one shard per device: the default batch is 20 000 reads PER DEVICE (a shard stays as large as the
    one-device batch), an explicit --batch_reads is taken as given. (Option handling only: the run then fails on the missing files.)"""
    env = dict(os.environ, POREGEN_BATCH_PROBE="1")
    def probe(extra):
        r = subprocess.run([BIN, "gmove", "nope.blow5", "nope.paf", str(tmp_path / "o"), "--fastq", "nope.fastq"] + extra, capture_output=True, text=True, env=env)
        line = [x for x in r.stderr.splitlines() if x.startswith("[batch probe]")]
        assert line, r.stderr
        return line[0]
    assert "batch_reads 20000 (1 device x 20000)" in probe([])
    assert "batch_reads 160000 (8 devices x 20000)" in probe(["--devices", "0,1,2,3,4,5,6,7"])
    assert "batch_reads 60000 (3 devices x 20000)" in probe(["--devices", "0,0,0"])
    assert "batch_reads 100 (3 devices x 33)" in probe(["--devices", "0,0,0", "--batch_reads", "100"])


@pytest.mark.gpu
@pytest.mark.parametrize("devopts", [["--devices", "0,0,0"], ["--devices", "0", "--exchange", "rccl"]], ids=["three_shards_host_exchange", "rccl_one_rank"])
def test_devices_option_equals_oracle(tmp_path, devopts):
    """`poregen gmove --devices`: the multi-GPU job from the C++ host (pg_job_*). On a one-GPU box: three shards on device 0
    (exchange through host memory) and the RCCL all-gather on a one-rank communicator; indels + shuffled whitelist slice +
    -d, several batches; directory byte-identical to the oracle CLI's."""
    rng = np.random.default_rng(9)
    full = ["".join(t) for t in __import__("itertools").product("ACGU", repeat=5)]
    wl = [full[i] for i in rng.permutation(len(full))[:300]]
    (tmp_path / "wl.txt").write_text("".join(k + "\n" for k in wl))
    b = synth.make_batch(260, kind="rna004", seed=94, indel_rate=0.03)
    pre = str(tmp_path / "syn"); synth.write_files(b, pre)
    args = [pre + ".slow5", pre + ".paf", "--fastq", pre + ".fastq", "-k", "5", "--rna", "--scaling", "1", "--min_dur", "20", "--max_dur", "40",
            "--kmer_file", tmp_path / "wl.txt", "--index_start", "51", "--index_end", "250", "--kmer_pick_margin", "2", "--sample_limit", "7", "-d"]
    r = cli(args + [tmp_path / "gpu", "--batch_reads", "100"] + devopts); assert r.returncode == 0, r.stderr
    assert ("RCCL all-gather" in r.stderr) == ("rccl" in devopts)
    o = oracle_cli(args + [tmp_path / "cpu"]); assert o.returncode == 0, o.stderr
    assert_same_dirs(tmp_path / "gpu", tmp_path / "cpu")
    # the whole list, filled early: the job stops reading like one context does
    args2 = [pre + ".slow5", pre + ".paf", "--fastq", pre + ".fastq", "-k", "3", "--rna", "--scaling", "1", "--min_dur", "20", "--max_dur", "40",
             "--file_limit", "64", "--sample_limit", "20", "--raw_model", tmp_path / "m_job.txt"]
    r = cli(args2 + [tmp_path / "gpu2", "--batch_reads", "60"] + devopts); assert r.returncode == 0, r.stderr
    r1 = cli(args2[:-1] + [tmp_path / "m_one.txt", tmp_path / "gpu3", "--batch_reads", "60"]); assert r1.returncode == 0, r1.stderr
    o = oracle_cli(args2[:-2] + [tmp_path / "cpu2"]); assert o.returncode == 0, o.stderr
    assert_same_dirs(tmp_path / "gpu2", tmp_path / "cpu2")
    assert open(tmp_path / "m_job.txt").read() == open(tmp_path / "m_one.txt").read()


@pytest.mark.gpu
@pytest.mark.parametrize("damage", ["truncated_record", "zstd_header", "exzd_header", "svb_block"])
def test_gmove_exits_1_on_a_corrupt_blow5(tmp_path, damage):
    """A damaged BLOW5 file ends `poregen gmove` with the reader's message and exit status 1 (src/gmove.cpp:493-503, 746-749: slow5lib's
    failures are `exit(EXIT_FAILURE)` there), whether the damage is met while the file is indexed or when the read is fetched."""
    import struct
    import zlib
    b = synth.make_batch(4, read_len=800, kind="dna_r10", seed=12)
    synth.write_paf_fastq(b, str(tmp_path / "r"))
    p = tmp_path / "r.blow5"
    synth.write_blow5(b, str(p), compress=True)
    d = bytearray(p.read_bytes())
    if damage == "truncated_record":
        d = d[:len(d) - 300]; msg = "BLOW5"
    elif damage == "zstd_header":
        d[9] = 2; msg = "zstd"   # zlib records announced as zstd (round 6 reads zstd): "zstd error in BLOW5 record", or "libzstd.so.1 was not found"
    elif damage == "exzd_header":
        d[14] = 2; msg = "signal compression other than none/svb-zd is not supported"
    else:  # the second record: a valid zlib stream whose streamvbyte block announces more values than it holds
        hlen = struct.unpack_from("<I", d, 64)[0]; pos = 68 + hlen
        sz = struct.unpack_from("<Q", d, pos)[0]; pos2 = pos + 8 + sz
        sz2 = struct.unpack_from("<Q", d, pos2)[0]
        body = bytearray(zlib.decompress(bytes(d[pos2 + 8:pos2 + 8 + sz2])))
        idl = struct.unpack_from("<H", body, 0)[0]
        struct.pack_into("<I", body, 2 + idl + 4 + 32 + 8, 10 ** 6)   # the block's count field
        z = zlib.compress(bytes(body))
        d = d[:pos2] + struct.pack("<Q", len(z)) + z + d[pos2 + 8 + sz2:]
        msg = "corrupt streamvbyte block"
    p.write_bytes(d)
    r = subprocess.run([BIN, "gmove", "-k", "5", "--file_limit", "1024", str(p), str(tmp_path / "r.paf"), "--fastq", str(tmp_path / "r.fastq"), str(tmp_path / "out")],
                       capture_output=True, text=True)
    assert r.returncode == 1 and msg in r.stderr, (r.returncode, r.stderr[-400:])


@pytest.mark.gpu
def test_whole_list_job_ramps_its_batches_and_stops_near_the_completing_read(tmp_path):
    """The reference stops reading at the read that completes the last k-mer (gmove.cpp:733-735). A whole-list job at the default batch
    size ramps its batches (2 048 reads, 4 096, ...; POREGEN_BATCH_RAMP=N: N, 2N, ...) so that it ends within a factor of two of that read
    instead of working through a first batch of 20 000: same directory as the oracle CLI, far fewer reads read than the file holds; an
    explicit --batch_reads and a slice keep their one size."""
    import re
    b = synth.make_batch(1500, read_len=3000, kind="rna004", seed=17)
    pre = str(tmp_path / "syn")
    synth.write_blow5(b, pre + ".blow5"); synth.write_paf_fastq(b, pre)
    args = [pre + ".blow5", pre + ".paf", "--fastq", pre + ".fastq", "-k", "3", "--rna", "--scaling", "1", "--min_dur", "20", "--max_dur", "40",
            "--file_limit", "64", "--sample_limit", "20"]
    synth.write_files(b, pre)                                                # (the oracle CLI reads ASCII SLOW5)
    o = oracle_cli([pre + ".slow5"] + args[1:] + [tmp_path / "cpu"]); assert o.returncode == 0, o.stderr

    def reads_read(r):
        m = re.search(r"\[gmove\] (\d+) reads, (\d+) samples", r.stderr)
        assert m, r.stderr
        return int(m.group(1))
    r = cli(args + [tmp_path / "gpu"], env=dict(os.environ, POREGEN_BATCH_RAMP="16")); assert r.returncode == 0, r.stderr
    assert_same_dirs(tmp_path / "gpu", tmp_path / "cpu")
    n_ramp = reads_read(r)
    r1 = cli(args + [tmp_path / "gpu1"]); assert r1.returncode == 0, r1.stderr          # default: first batch 2 048 > the file: one batch
    assert_same_dirs(tmp_path / "gpu1", tmp_path / "cpu")
    assert reads_read(r1) == 1500
    assert n_ramp < 600, n_ramp                                                           # 16 + 32 + ... : a small multiple of the completing read
    r2 = cli(args + [tmp_path / "gpu2", "--batch_reads", "700"], env=dict(os.environ, POREGEN_BATCH_RAMP="0")); assert r2.returncode == 0
    assert_same_dirs(tmp_path / "gpu2", tmp_path / "cpu")
    assert reads_read(r2) in (700, 1400)
    sl = [x if x != "64" else "10" for x in args]                                        # a slice reads every line, in batches of one size
    r3 = cli(sl + [tmp_path / "gpu3"], env=dict(os.environ, POREGEN_BATCH_RAMP="16")); assert r3.returncode == 0, r3.stderr
    assert reads_read(r3) == 1500
    o3 = oracle_cli([pre + ".slow5"] + sl[1:] + [tmp_path / "cpu3"]); assert o3.returncode == 0
    assert_same_dirs(tmp_path / "gpu3", tmp_path / "cpu3")


@pytest.mark.gpu
def test_gmove_reads_zstd_compressed_blow5(tmp_path):
    """`poregen gmove` on a BLOW5 file with zstd-compressed records (the reference's `make zstd=1` build, /root/reference/Makefile:12-13,67):
    the same output directory as the oracle CLI on the ASCII SLOW5 of the same reads."""
    if synth.zstd_compress(b"x") is None:
        pytest.skip("no libzstd.so.1 on this machine")
    b = synth.make_batch(120, read_len=3000, kind="rna004", seed=23)
    pre = str(tmp_path / "syn")
    synth.write_files(b, pre)
    synth.write_blow5(b, pre + ".blow5", compress="zstd")
    args = [pre + ".paf", "--fastq", pre + ".fastq", "-k", "3", "--rna", "--scaling", "1", "--min_dur", "20", "--max_dur", "40", "--file_limit", "64", "--sample_limit", "50"]
    r = cli([pre + ".blow5"] + args + [tmp_path / "gpu"]); assert r.returncode == 0, r.stderr
    o = oracle_cli([pre + ".slow5"] + args + [tmp_path / "cpu"]); assert o.returncode == 0, o.stderr
    assert_same_dirs(tmp_path / "gpu", tmp_path / "cpu")
