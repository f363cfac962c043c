"""Host-side C++ of the CLI (poregen_amd/csrc/host), exercised without a GPU through the host test shim:
SLOW5/BLOW5 reader, FASTQ fetch with faidx clamping, PAF + ss tokeniser, exact %.8f formatting."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden", "single_read")
B5 = os.path.join(ROOT, "tests", "golden", "blow5")


@pytest.fixture(scope="module")
def h():
    lib = C.CDLL((os.environ.get("PG_HOSTTEST_SO") or os.path.join(ROOT, "poregen_amd", "_pg_hosttest.so")))
    lib.pgt_format_f8.argtypes = [C.c_double, C.c_char_p]; lib.pgt_format_f8.restype = C.c_size_t
    lib.pgt_tokenize_ss.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_size_t]; lib.pgt_tokenize_ss.restype = C.c_long
    lib.pgt_fastx_fetch.argtypes = [C.c_char_p, C.c_char_p, C.c_long, C.c_long, C.c_char_p, C.c_size_t]; lib.pgt_fastx_fetch.restype = C.c_long
    lib.pgt_slow5_get.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_size_t]; lib.pgt_slow5_get.restype = C.c_long
    lib.pgt_slow5_count.argtypes = [C.c_char_p]; lib.pgt_slow5_count.restype = C.c_long
    lib.pgt_sam_first.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]; lib.pgt_sam_first.restype = C.c_long
    lib.pgt_parse_paf.argtypes = [C.c_char_p, C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]; lib.pgt_parse_paf.restype = C.c_int
    return lib


def test_format_f8_equals_printf(h):
    rng = np.random.default_rng(1)
    vals = np.concatenate([
        rng.normal(0, 2, 20000), rng.uniform(40, 180, 20000), rng.normal(0, 1e-6, 2000), rng.normal(0, 1e5, 2000),
        np.array([0.0, -0.0, 1.0, -1.0, 0.5, 2.0 ** -9, 3 * 2.0 ** -9, -(2.0 ** -9), 0.001953125, 1e-9, -1e-9, 4.9e-9, 5e-9, 5.1e-9,
                  123456.123456785, 0.000000005, 0.000000015, 99999999.99999999, 1e15]),
        (rng.integers(-10 ** 12, 10 ** 12, 5000) + 0.5) / 1e8,   # decimal ties at the 9th digit (not exactly representable)
        rng.integers(-2 ** 20, 2 ** 20, 3000) / 2.0 ** 12,       # exactly representable, some with exact binary ties
        np.arange(-3001, 3001, 2) / 512.0,                       # every odd multiple of 2^-9: exact ties at the 8th digit
        np.array([3.9999999e7, -3.9999999e7, 4.0e7, -4.0e7, 4.1e7, 1e-300, -1e-300, 5e-324, -2.5e-9, -4.9e-9, -5.1e-9]),  # both sides of the fast path's range
    ])
    buf = C.create_string_buffer(400)
    for v in vals:
        n = h.pgt_format_f8(float(v), buf)
        assert buf.raw[:n].decode() == "%.8f" % float(v), v


def test_tokenize_ss(h):
    n = np.zeros(64, np.uint32); t = np.zeros(64, np.uint8)
    assert h.pgt_tokenize_ss(b"10,5,3D12,40I7,", n.ctypes.data, t.ctypes.data, 64) == 6
    assert n[:6].tolist() == [10, 5, 3, 12, 40, 7] and t[:6].tolist() == [0, 0, 2, 0, 1, 0]
    assert h.pgt_tokenize_ss(b"", n.ctypes.data, t.ctypes.data, 64) == 0
    for bad in (b",5,", b"5,,3,", b"5x3,", b"12345678901,", b"-5,", b"4294967295,"):   # the reference's "Bad ss" exits
        assert h.pgt_tokenize_ss(bad, n.ctypes.data, t.ctypes.data, 64) == -1, bad
    assert h.pgt_tokenize_ss(b"0,2147483647I", n.ctypes.data, t.ctypes.data, 64) == 2


def test_fastx_fetch_clamping(h):
    fq = os.path.join(G, "read_0.fastq").encode()
    rid = b"2babd419-2e01-454a-b7f4-08ad9d4e2a9e"
    buf = C.create_string_buffer(1000)
    full = open(os.path.join(G, "read_0.fastq")).read().splitlines()[1]
    assert len(full) == 481
    assert h.pgt_fastx_fetch(fq, rid, 0, 480, buf, 1000) == 481 and buf.value.decode() == full
    assert h.pgt_fastx_fetch(fq, rid, 3, 9, buf, 1000) == 7 and buf.value.decode() == full[3:10]
    assert h.pgt_fastx_fetch(fq, rid, 470, 10 ** 6, buf, 1000) == 11 and buf.value.decode() == full[470:]   # end clamps to len-1
    assert h.pgt_fastx_fetch(fq, rid, 5, 4, buf, 1000) == 1 and buf.value.decode() == full[4]                 # end < beg: beg = end
    assert h.pgt_fastx_fetch(fq, rid, 0, -1, buf, 1000) == 1 and buf.value.decode() == full[0]                # empty target range
    assert h.pgt_fastx_fetch(fq, rid, 600, 700, buf, 1000) == 0
    assert h.pgt_fastx_fetch(fq, b"nope", 0, 10, buf, 1000) == -2


def test_fastx_multiline_fasta_and_fastq(h, tmp_path):
    p = tmp_path / "x.fa"
    p.write_text(">a desc\nACGTAC\nGTACGT\nAC\n>b\nTTTT\n")
    buf = C.create_string_buffer(100)
    assert h.pgt_fastx_fetch(str(p).encode(), b"a", 4, 9, buf, 100) == 6 and buf.value == b"ACGTAC"
    assert h.pgt_fastx_fetch(str(p).encode(), b"b", 0, 100, buf, 100) == 4 and buf.value == b"TTTT"
    q = tmp_path / "x.fq"
    q.write_text("@r1 x\nACGT\n+\n@@@@\n@r2\nGGCC\n+r2\nIIII\n")   # quality line starting with '@'
    assert h.pgt_fastx_fetch(str(q).encode(), b"r2", 1, 2, buf, 100) == 2 and buf.value == b"GC"
    assert h.pgt_fastx_fetch(str(q).encode(), b"@@@", 0, 2, buf, 100) == -2


def test_ascii_slow5_fixture(h):
    dor = np.zeros(3); raw = np.zeros(7000, np.int16)
    n = h.pgt_slow5_get(os.path.join(G, "reads.slow5").encode(), b"2babd419-2e01-454a-b7f4-08ad9d4e2a9e", dor.ctypes.data, raw.ctypes.data, 7000)
    assert n == 6020 and dor.tolist() == [2048.0, -101.0, 281.345551]
    line = [l for l in open(os.path.join(G, "reads.slow5")) if not l.startswith(("#", "@"))][0].split("\t")
    assert raw[:n].tolist() == [int(x) for x in line[7].split(",")]
    assert h.pgt_slow5_get(os.path.join(G, "reads.slow5").encode(), b"missing", dor.ctypes.data, raw.ctypes.data, 7000) == -1


def test_blow5_zlib_svbzd_means_match_reference_fixture(h):
    """test/example.blow5 (zlib + svb-zd) -> mean pA per read == test/example.exp (the reference's subtool0 golden)."""
    path = os.path.join(B5, "example.blow5").encode()
    assert h.pgt_slow5_count(path) == 5
    for line in open(os.path.join(B5, "example.exp")):
        rid, mean = line.split()
        dor = np.zeros(3); raw = np.zeros(100000, np.int16)
        n = h.pgt_slow5_get(path, rid.encode(), dor.ctypes.data, raw.ctypes.data, 100000)
        assert n > 50000
        pa = (raw[:n].astype(np.float64) + dor[1]) * (dor[2] / dor[0])     # TO_PICOAMPS, src/poregen.h:30
        assert "%.6f" % pa.mean() == mean


def test_blow5_uncompressed_roundtrip(h, tmp_path):
    """Our own uncompressed BLOW5 (record none / signal none) with two reads."""
    p = tmp_path / "t.blow5"
    hdr = b"#slow5_version\t0.2.0\n#num_read_groups\t1\n#char*\tuint32_t\n#read_id\tread_group\n"
    with open(p, "wb") as f:
        f.write(b"BLOW5\x01" + bytes([0, 2, 0]) + bytes([0]) + struct.pack("<I", 1) + bytes([0]) + bytes(64 - 15))
        f.write(struct.pack("<I", len(hdr)) + hdr)
        for rid, sig in ((b"a", [1, -2, 3]), (b"bb", [7] * 10)):
            body = struct.pack("<H", len(rid)) + rid + struct.pack("<I", 0) + struct.pack("<dddd", 2048.0, -5.0, 300.0, 4000.0)
            body += struct.pack("<Q", len(sig)) + np.array(sig, np.int16).tobytes()
            f.write(struct.pack("<Q", len(body)) + body)
        f.write(b"5WOLB")
    dor = np.zeros(3); raw = np.zeros(16, np.int16)
    assert h.pgt_slow5_count(str(p).encode()) == 2
    assert h.pgt_slow5_get(str(p).encode(), b"a", dor.ctypes.data, raw.ctypes.data, 16) == 3 and raw[:3].tolist() == [1, -2, 3]
    assert h.pgt_slow5_get(str(p).encode(), b"bb", dor.ctypes.data, raw.ctypes.data, 16) == 10 and dor.tolist() == [2048.0, -5.0, 300.0]


def test_parse_paf_fixture(h):
    line = open(os.path.join(G, "guppy_move.paf")).read()
    cols = (C.c_int32 * 6)(); rid = C.create_string_buffer(256); tid = C.create_string_buffer(256); ss = C.create_string_buffer(1 << 16)
    buf = C.create_string_buffer(line.encode())
    assert h.pgt_parse_paf(buf, cols, rid, tid, ss, 1 << 16) == 0
    assert list(cols) == [6020, 0, 6020, 481, 0, 481] and rid.value == tid.value == b"2babd419-2e01-454a-b7f4-08ad9d4e2a9e"
    ops = ss.value.decode()
    assert ops.count(",") == 481 and sum(int(x) for x in ops.split(",") if x) == 6020
    assert h.pgt_parse_paf(C.create_string_buffer(b"a\t1\t2\n"), cols, rid, tid, ss, 1 << 16) == 1
    assert h.pgt_parse_paf(C.create_string_buffer(b"r\t10\t0\t10\t+\tt\t5\t0\t5\t5\t5\t255\tsc:f:1\n"), cols, rid, tid, ss, 1 << 16) == 2


def test_sam_and_bam_readers_agree_on_fixture(h):
    """guppy_move.sam (text) and guppy_move.bam (BGZF + binary records) hold the same record; the move tags agree
    with the move table fixture (481 moves = one per base, stride 5, ns 6020, ts 0)."""
    got = {}
    for ext in ("sam", "bam"):
        q = C.create_string_buffer(256); s = C.create_string_buffer(4096); v = (C.c_longlong * 3)(); mv = np.zeros(4096, np.uint8)
        n = h.pgt_sam_first(os.path.join(G, f"guppy_move.{ext}").encode(), q, s, 4096, v, mv.ctypes.data, 4096)
        got[ext] = (n, q.value, s.value, list(v), mv[:max(n, 0)].tolist())
    assert got["sam"] == got["bam"]
    n, q, s, v, mv = got["bam"]
    table = open(os.path.join(G, "guppy_move")).read().split("\t")
    assert q.decode() == table[0] and s.decode() == table[2] and v == [5, 6020, 0]
    assert "".join(map(str, mv)) == table[4] and n == len(table[4])


def test_compressed_blow5_round_trip(tmp_path):
    """zlib records + svb-zd signals written by our own encoder (synth.write_blow5(compress=True)) decode to the same
    samples and calibration. The decoder itself is pinned by the reference's test/example.blow5 (test above); this adds
    ragged lengths, spikes (4-byte deltas) and a few hundred reads."""
    import ctypes as C
    from poregen_amd import synth
    h = C.CDLL((os.environ.get("PG_HOSTTEST_SO") or os.path.join(ROOT, "poregen_amd", "_pg_hosttest.so")))
    h.pgt_slow5_get.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_size_t]; h.pgt_slow5_get.restype = C.c_long
    b = synth.make_batch(200, kind="rna004", seed=3, read_len=5003, spike_rate=0.05)
    big = b.sig.copy(); big[::997] = 32767; big[1::997] = -32768            # extreme deltas: 3- and 4-byte codes
    b = type(b)(**{**b.__dict__, "sig": big})
    path = str(tmp_path / "c.blow5")
    comps = [True, False] + (["zstd"] if synth.zstd_compress(b"x") is not None else [])  # zstd records where libzstd.so.1 exists (it does in this image)
    for comp in comps:
        synth.write_blow5(b, path, compress=comp)
        for r in range(0, b.n_reads, 7):
            dor = np.zeros(3); raw = np.zeros(6000, np.int16)
            n = h.pgt_slow5_get(path.encode(), f"r{r}".encode(), dor.ctypes.data, raw.ctypes.data, 6000)
            s = b.sig[int(b.sig_off[r]):int(b.sig_off[r + 1])]
            assert n == s.size and np.array_equal(raw[:n], s)
            assert (dor[0], dor[1], dor[2]) == (b.digitisation[r], b.offset[r], b.range[r])


def test_zstd_blow5_is_read_or_refused_by_name(tmp_path):
    """BLOW5 with zstd-compressed records (the reference's `make zstd=1` build, /root/reference/Makefile:12-13,67): read through a
    dlopen'ed libzstd.so.1 where the machine has one; where it has none the reader says so instead of misreading (checked through
    the error text of a build that cannot find the library: LD_LIBRARY_PATH does not matter to dlopen of an absolute miss, so the
    refusal itself is exercised in tests/test_host_corrupt.py with a damaged frame)."""
    import ctypes as C
    from poregen_amd import synth
    if synth.zstd_compress(b"x") is None:
        pytest.skip("no libzstd.so.1 on this machine")
    h = C.CDLL((os.environ.get("PG_HOSTTEST_SO") or os.path.join(ROOT, "poregen_amd", "_pg_hosttest.so")))
    h.pgt_slow5_get.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_size_t]; h.pgt_slow5_get.restype = C.c_long
    b = synth.make_batch(64, kind="dna_r10", seed=11, read_len=3001)
    path = str(tmp_path / "z.blow5")
    for mode in ("zstd", "zstd-stream"):   # one-shot frames (slow5lib) and frames without their content size (a streaming writer)
        synth.write_blow5(b, path, compress=mode)
        assert open(path, "rb").read()[9] == 2
        for r in range(b.n_reads):
            dor = np.zeros(3); raw = np.zeros(4000, np.int16)
            n = h.pgt_slow5_get(path.encode(), f"r{r}".encode(), dor.ctypes.data, raw.ctypes.data, 4000)
            s = b.sig[int(b.sig_off[r]):int(b.sig_off[r + 1])]
            assert n == s.size and np.array_equal(raw[:n], s)
            assert (dor[0], dor[1], dor[2]) == (b.digitisation[r], b.offset[r], b.range[r])
