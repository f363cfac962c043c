"""The CPU oracle against every recorded answer for this path: the known-answer vectors of SURVEY.md Appendix C
on the reference's own input fixtures, the reference tests' cross-format invariant (table == PAF at
--kmer_pick_margin 0, test/test_gmove.sh:79-80,95-96) and its exit-status expectations (test_gmove.sh:50,58,66)."""
import filecmp
import json
import os
import subprocess

import numpy as np
import pytest

import orc

G = os.path.join(os.path.dirname(__file__), "golden", "single_read")
KA = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ka_vectors.json")))


def run_cli(args, out):
    a = [x.replace("{G}", G).replace("{OUT}", str(out)) for x in args]
    return subprocess.run([orc.CLI] + a, capture_output=True, text=True)


def read_freq(out):
    return dict((l.split("\t")[0], int(l.split("\t")[1])) for l in open(os.path.join(out, "freq.txt")).read().splitlines())


def dump(out, kmer):
    return open(os.path.join(out, "dump", kmer)).read()


def test_ka1_paf_kmer_file_default_margin(tmp_path):
    out = tmp_path / "o"
    assert run_cli(KA["KA1"]["args"], out).returncode == 0
    f = read_freq(out)
    assert len(f) == KA["KA1"]["freq_lines"]
    assert sorted(k for k, v in f.items() if v == 0) == sorted(KA["KA1"]["zeros"])
    for k, v in KA["KA1"]["freq"].items():
        assert f[k] == v
    assert sum(1 for v in f.values() if v == 20) == KA["KA1"]["n_at_20"]
    assert sum(1 for v in f.values() if v == 1) == 53 - 4 - 1 - 18 - 4
    for k, v in KA["KA1"]["dump"].items():
        assert dump(out, k) == v


@pytest.mark.parametrize("name", ["KA2", "KA3"])
def test_ka23_medmad_and_zero_fill(tmp_path, name):
    out = tmp_path / "o"
    assert run_cli(KA[name]["args"], out).returncode == 0
    for k, v in KA[name]["dump"].items():
        assert dump(out, k) == v
    # the read's median / MAD themselves
    raw = np.array([int(x) for x in [l for l in open(os.path.join(G, "reads.slow5")) if not l.startswith(("#", "@"))][0].split("\t")[7].split(",")], np.int16)
    pa = (raw.astype(np.float64) + (-101.0)) * (281.345551 / 2048.0)
    pmin = 100.0 if name == "KA3" else 40.0
    x = np.where((pa < pmin) | (pa > 180.0), 0.0, pa)
    L = orc.lib()
    med = L.orc_median(x.ctypes.data, x.size)
    mad = L.orc_madf(x.ctypes.data, x.size, med)
    assert abs(med - KA[name]["median"]) < 5e-11 and abs(mad - KA[name]["mad"]) < 5e-11
    if name == "KA3":
        assert int((x == 0.0).sum()) == KA[name]["zeroed"]


def test_ka4_table_path_and_paf_margin(tmp_path):
    out = tmp_path / "t"
    assert run_cli(KA["KA4"]["args"], out).returncode == 0
    assert read_freq(out) == KA["KA4"]["freq"]
    assert dump(out, "ATGTTG") == KA["KA4"]["dump"]["ATGTTG"]
    out2 = tmp_path / "p"
    assert run_cli(KA["KA4_paf_default_margin"]["args"], out2).returncode == 0
    assert read_freq(out2) == KA["KA4_paf_default_margin"]["freq"]


def test_ka5_delimiter(tmp_path):
    out = tmp_path / "o"
    assert run_cli(KA["KA5"]["args"], out).returncode == 0
    assert dump(out, "ATGTTG") == ":"


def test_ka6_default_run(tmp_path):
    out = tmp_path / "o"
    assert run_cli(KA["KA6"]["args"], out).returncode == 0
    files = sorted(os.listdir(out / "dump"))
    assert len(files) == 50 and files[0] == "AAAAAAAAA"
    assert all(v == 0 for v in read_freq(out).values())


@pytest.mark.parametrize("kf", ["single_kmer_file.txt", "kmer_file.txt"])
def test_reference_invariant_table_equals_paf_at_margin0(tmp_path, kf):
    """test_gmove.sh cases 1.1 == 1.2 and 2.1 == 2.2 (valid with --kmer_pick_margin 0)."""
    t, p = tmp_path / "table", tmp_path / "paf"
    assert run_cli(["-k", "6", "-m", "0", "{G}/reads.slow5", "{G}/guppy_move", "{OUT}", "--kmer_file", "{G}/" + kf], t).returncode == 0
    assert run_cli(["-k", "6", "{G}/reads.slow5", "{G}/guppy_move.paf", "{OUT}", "--fastq", "{G}/read_0.fastq", "--kmer_file", "{G}/" + kf,
                    "--kmer_pick_margin", "0"], p).returncode == 0
    assert open(t / "freq.txt").read() == open(p / "freq.txt").read()
    cmp = filecmp.dircmp(t / "dump", p / "dump")
    assert not cmp.left_only and not cmp.right_only
    _, mismatch, errors = filecmp.cmpfiles(t / "dump", p / "dump", cmp.common_files, shallow=False)
    assert not mismatch and not errors


def test_ka7_exit_statuses(tmp_path):
    assert run_cli([], tmp_path / "a").returncode == 1                                                       # 0.1 help
    assert run_cli(["{G}/reads.slow5", "{G}/guppy_move", "--kmer_file", "{G}/kmer_file.txt", "{OUT}"], tmp_path / "b").returncode == 1  # 0.3
    assert run_cli(["{G}/reads.slow5", "{G}/guppy_move.paf", "--file_limit", "50", "{OUT}"], tmp_path / "c").returncode == 1            # 0.5
    d = tmp_path / "d"; d.mkdir(); (d / "x").write_text("x")
    assert run_cli(["{G}/reads.slow5", "{G}/guppy_move", "--file_limit", "50", "{OUT}"], d).returncode == 1  # non-empty dir


def test_rna_orientation_requires_flag():
    o = orc.Oracle(["ACGTA"], kmer_size=5)
    rc = o.paf_read(np.zeros(100, np.int16), 2048.0, 0.0, 281.0, 0, 10, 0, "ACGTACGTAC", "10," * 10)
    assert rc == orc.ORC_ERR_RNA_FLAG
