"""`poregen reform` against the reference's OWN expected files (test/test_reform.sh testcases 1-18; inputs and
expected outputs copied as data under tests/golden/reform/, the two 1.6 MB TSVs gzip-compressed). Host-only: runs on
any box. This is the one sub-tool for which the reference holds golden outputs, so parity here is pinned, byte for
byte, by reference-owned vectors."""
import gzip
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.environ.get("PG_REFORM_BIN") or os.path.join(ROOT, "bin", "poregen")  # PG_REFORM_BIN: the sanitizer build of the subtool (make asan)
R = os.path.join(ROOT, "tests", "golden", "reform")


def reform(*args):
    return subprocess.run([BIN, "reform"] + [str(a) for a in args], capture_output=True)


def golden(name):
    p = os.path.join(R, name)
    if os.path.exists(p):
        return open(p, "rb").read()
    return gzip.open(p + ".gz", "rb").read()


def test_usage_and_rejected_options():
    assert reform().returncode != 0                                              # testcase 1
    assert reform("-k0", "-m1", "-c", f"{R}/guppy_one_read.bam").returncode != 0  # testcase 2
    assert reform("-k0", "-m0", "-c", f"{R}/guppy_one_read.bam").returncode != 0  # testcase 3
    assert reform("-k9", "-m10", "-c", f"{R}/guppy_one_read.bam").returncode != 0  # testcase 4
    r = reform("-h"); assert r.returncode == 0 and b"Usage: poregen reform" in r.stdout


@pytest.mark.parametrize("k,m", [(1, 0), (9, 0), (9, 1), (9, 6), (9, 8)])
@pytest.mark.parametrize("fmt", ["paf", "tsv"])
def test_guppy_one_read(k, m, fmt):                                              # testcases 5-14
    args = [f"-k{k}", f"-m{m}"] + (["-c"] if fmt == "paf" else []) + [f"{R}/guppy_one_read.bam"]
    r = reform(*args)
    assert r.returncode == 0, r.stderr
    assert r.stdout == golden(f"r1k{k}m{m}.{fmt}")


@pytest.mark.parametrize("k,m", [(9, 8), (1, 0)])
@pytest.mark.parametrize("fmt", ["paf", "tsv"])
def test_dorado_sam(k, m, fmt):                                                  # testcases 15-18
    args = [f"-k{k}", f"-m{m}"] + (["-c"] if fmt == "paf" else []) + [f"{R}/slow5-dorado.sam"]
    r = reform(*args)
    assert r.returncode == 0, r.stderr
    assert r.stdout == golden(f"dr2k{k}m{m}.{fmt}")


def test_output_file_option(tmp_path):
    out = tmp_path / "o.paf"
    r = reform("-k9", "-m0", "-c", "-o", out, f"{R}/guppy_one_read.bam")
    assert r.returncode == 0 and r.stdout == b""
    assert out.read_bytes() == golden("r1k9m0.paf")


def test_missing_tags_are_errors():
    # buttery_eel.sam carries mv but neither ns nor ts: reform returns -1 (src/reform.cpp:212-215)
    r = reform("-k9", "-m0", "-c", f"{R}/buttery_eel.sam")
    assert r.returncode != 0 and b"tag 'ns' is not found" in r.stderr and r.stdout == b""


def test_two_reads_bam_paf_and_tsv_agree():
    """No golden for guppy_two_reads.bam: check the two output formats against each other (each ss duration is the
    width of the matching TSV row; columns 3/4 are the first start / last end)."""
    paf = reform("-k9", "-m0", "-c", f"{R}/guppy_two_reads.bam"); tsv = reform("-k9", "-m0", f"{R}/guppy_two_reads.bam")
    assert paf.returncode == 0 and tsv.returncode == 0
    rows = {}
    for line in tsv.stdout.decode().splitlines():
        rid, idx, s, e = line.split("\t")
        rows.setdefault(rid, []).append((int(idx), int(s), int(e)))
    lines = paf.stdout.decode().splitlines()
    assert len(lines) == len(rows) == 2
    for line in lines:
        c = line.split("\t")
        durs = [int(x) for x in c[12][5:].rstrip(",").split(",")]
        rr = rows[c[0]]
        assert [i for i, _, _ in rr] == list(range(len(rr))) and len(rr) == int(c[6]) == len(durs)
        assert durs == [e - s for _, s, e in rr]
        assert int(c[2]) == rr[0][1] and int(c[3]) == rr[-1][2]


def test_reform_reproduces_the_gmove_fixture_ss():
    """The reference's gmove fixture guppy_move.paf (test/data/raw/gmove/single_read) holds the ss string of the
    same read's BAM move table: `reform -k1 -m0 -c` of guppy_move.bam must reproduce it and columns 1-11."""
    G = os.path.join(ROOT, "tests", "golden", "single_read")
    for f in ("guppy_move.bam", "guppy_move.sam"):
        r = reform("-k1", "-m0", "-c", f"{G}/{f}")
        assert r.returncode == 0
        mine = r.stdout.decode().rstrip("\n").split("\t")
        ref = open(f"{G}/guppy_move.paf").read().rstrip("\n").split("\t")
        assert mine[:11] == ref[:11]
        assert mine[12] == [c for c in ref if c.startswith("ss:Z:")][0]


@pytest.mark.gpu
def test_reform_then_gmove_equals_fixture_paf(tmp_path):
    """reform -c | gmove --paf == gmove on the fixture PAF (device path, whole dump directory)."""
    import filecmp
    G = os.path.join(ROOT, "tests", "golden", "single_read")
    paf = tmp_path / "re.paf"
    assert reform("-k1", "-m0", "-c", "-o", paf, f"{G}/guppy_move.bam").returncode == 0
    outs = []
    for name, p in (("a", paf), ("b", f"{G}/guppy_move.paf")):
        out = tmp_path / name
        r = subprocess.run([BIN, "gmove", "-k", "6", f"{G}/reads.slow5", str(p), "--fastq", f"{G}/read_0.fastq", str(out)], capture_output=True)
        assert r.returncode == 0, r.stderr
        outs.append(out)
    assert (outs[0] / "freq.txt").read_bytes() == (outs[1] / "freq.txt").read_bytes()
    names = sorted(os.listdir(outs[0] / "dump"))
    _, mismatch, errors = filecmp.cmpfiles(outs[0] / "dump", outs[1] / "dump", names, shallow=False)
    assert not mismatch and not errors


def _sam(tmp_path, name, tags, seq="ACGTACGTACGT"):
    p = tmp_path / name
    p.write_text("@HD\tVN:1.6\n" + "\t".join(["read1", "4", "*", "0", "0", "*", "*", "0", "0", seq, "*"] + tags) + "\n")
    return p


def test_reform_error_returns(tmp_path):
    """The reference's `return -1` paths (src/reform.cpp:212-240) and the inputs on which its scans would run past the
    mv array: all end with a non-zero exit status and no complete record."""
    mv = "mv:B:c,5,1,0,1,0,0,1,1,0,1,1,0,1,1,1,0,1"
    cases = [
        (["ts:i:10", mv], b"tag 'ns' is not found"),
        (["ns:i:200", mv], b"tag 'ts' is not found"),
        (["ns:i:200", "ts:i:10"], b"NULL returned for tag mv"),
        (["ns:i:200", "ts:i:10", "mv:B:C,5,1,0,1"], b"tag 'mv' specification is incorrect"),
        (["ns:i:200", "ts:i:10", "mv:B:c,6,1,0,1,1,1,1,1,1,1,1"], b"expected stride of 5 is missing"),
        (["ns:i:200", "ts:i:10", "mv:B:c,5,0,0,0,0"], b"fewer moves than sig_move_offset + 1"),
    ]
    for i, (tags, msg) in enumerate(cases):
        r = reform("-k3", "-m0", "-c", _sam(tmp_path, f"e{i}.sam", tags))
        assert r.returncode != 0 and msg in r.stderr, (tags, r.stderr)


def test_reform_small_hand_checked_record(tmp_path):
    """12 bases, k=3 -> 10 k-mers; moves at mv indices 1,3,6,7,9,10,12,13,14,16; ts=10, ns=200, stride 5, -m 0.
    TSV rows: k-mer j spans [ts + (p_j - 1)*5, ts + (p_{j+1} - 1)*5), the last one ends at ns."""
    p = _sam(tmp_path, "ok.sam", ["ns:i:200", "ts:i:10", "mv:B:c,5,1,0,1,0,0,1,1,0,1,1,0,1,1,1,0,1"])
    pos = [1, 3, 6, 7, 9, 10, 12, 13, 14, 16]
    tsv = reform("-k3", "-m0", p)
    assert tsv.returncode == 0
    rows = [l.split("\t") for l in tsv.stdout.decode().splitlines()]
    starts = [10 + (q - 1) * 5 for q in pos]
    ends = starts[1:] + [200]
    assert rows == [["read1", str(j), str(starts[j]), str(ends[j])] for j in range(10)]
    paf = reform("-k3", "-m0", "-c", p)
    assert paf.returncode == 0
    c = paf.stdout.decode().rstrip("\n").split("\t")
    assert c[:12] == ["read1", "200", "10", "200", "+", "read1", "10", "0", "10", "10", "10", "255"]
    assert c[12] == "ss:Z:" + "".join(f"{e - s}," for s, e in zip(starts, ends))


def test_synthetic_bam_with_records_across_bgzf_blocks(tmp_path):
    """The BGZF/BAM reader on a synthetic file whose blocks are 777 bytes of payload (every record straddles several
    blocks) gives the same records as the SAM text of the same reads (reform is the CPU-side consumer of that reader)."""
    import sys
    sys.path.insert(0, ROOT)
    from poregen_amd import synth
    b = synth.make_batch(40, kind="dna_r10", seed=5, read_len=3000)
    pre = str(tmp_path / "s")
    synth.write_table_files(b, pre, trim=11)
    for blk in (777, 60000):
        synth.write_bam(b, pre + ".bam", trim=11, block_bytes=blk)
        for fmt in (["-c"], []):
            a = reform("-k5", "-m1", *fmt, pre + ".sam"); c = reform("-k5", "-m1", *fmt, pre + ".bam")
            assert a.returncode == 0 and c.returncode == 0 and a.stdout == c.stdout and len(a.stdout) > 1000
