"""Corrupt binary inputs to the host readers (own BLOW5 / streamvbyte decoder, own BGZF inflate + BAM records, SAM text): every one
must end in the reader's message and a clean failure -- `poregen` exits 1 -- never in a crash. The reference leaves these formats to
slow5lib / htslib (src/gmove.cpp:493-503, 745, 1067-1134). `make asan_test` runs this file against the sanitizer builds."""
import ctypes as C
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from poregen_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFORM = os.environ.get("PG_REFORM_BIN") or os.path.join(ROOT, "bin", "poregen")


@pytest.fixture(scope="module")
def shim():
    h = C.CDLL(os.environ.get("PG_HOSTTEST_SO") or os.path.join(ROOT, "poregen_amd", "_pg_hosttest.so"))
    for f in (h.pgt_slow5_scan, h.pgt_sambam_scan):
        f.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]; f.restype = C.c_long
    return h


@pytest.fixture(scope="module")
def batch():
    return synth.make_batch(6, read_len=600, kind="dna_r10", seed=5)


def scan(fn, path):
    buf = C.create_string_buffer(512)
    n = fn(str(path).encode(), buf, 512)
    return n, buf.value.decode(errors="replace")  # (a message may quote bytes of the damaged file)


def test_valid_files_scan_clean(shim, batch, tmp_path):
    synth.write_blow5(batch, str(tmp_path / "a.blow5")); synth.write_blow5(batch, str(tmp_path / "z.blow5"), compress=True)
    synth.write_table_files(batch, str(tmp_path / "t")); synth.write_bam(batch, str(tmp_path / "t.bam"), block_bytes=700)
    assert scan(shim.pgt_slow5_scan, tmp_path / "a.blow5") == (6, "")
    assert scan(shim.pgt_slow5_scan, tmp_path / "z.blow5") == (6, "")
    assert scan(shim.pgt_sambam_scan, tmp_path / "t.sam") == (6, "")
    assert scan(shim.pgt_sambam_scan, tmp_path / "t.bam") == (6, "")


@pytest.mark.parametrize("compress", [False, True])
def test_truncated_blow5_record(shim, batch, tmp_path, compress):
    p = tmp_path / "a.blow5"
    synth.write_blow5(batch, str(p), compress=compress)
    data = p.read_bytes()
    for cut in (len(data) - 5 - 37, len(data) // 2, 68 + 40, 70, 30):
        q = tmp_path / f"cut{cut}.blow5"; q.write_bytes(data[:cut])
        n, err = scan(shim.pgt_slow5_scan, q)
        assert n == -1 and ("truncated BLOW5" in err or "corrupt BLOW5" in err or "zlib error" in err), (cut, n, err)


def test_blow5_record_sizes_that_lie(shim, batch, tmp_path):
    p = tmp_path / "a.blow5"; synth.write_blow5(batch, str(p))
    data = bytearray(p.read_bytes())
    hlen = struct.unpack_from("<I", data, 64)[0]
    first = 68 + hlen
    for sz in (0, 1, 3, 2 ** 63, 2 ** 64 - 1, len(data)):           # record size field of the first record
        d = bytearray(data); struct.pack_into("<Q", d, first, sz)
        q = tmp_path / "lie.blow5"; q.write_bytes(d)
        n, err = scan(shim.pgt_slow5_scan, q)
        assert n == -1 and err, (sz, n, err)
    d = bytearray(data); struct.pack_into("<H", d, first + 8, 60000)   # read id longer than the record
    (tmp_path / "id.blow5").write_bytes(d)
    n, err = scan(shim.pgt_slow5_scan, tmp_path / "id.blow5"); assert n == -1 and "corrupt BLOW5 record" in err
    d = bytearray(data)                                                # len_raw_signal beyond the record
    idl = struct.unpack_from("<H", d, first + 8)[0]
    struct.pack_into("<Q", d, first + 8 + 2 + idl + 4 + 32, 10 ** 12)
    (tmp_path / "len.blow5").write_bytes(d)
    n, err = scan(shim.pgt_slow5_scan, tmp_path / "len.blow5"); assert n == -1 and "corrupt BLOW5 record" in err
    d = bytearray(data); struct.pack_into("<I", d, 64, 2 ** 31)        # header length beyond the file
    (tmp_path / "hdr.blow5").write_bytes(d)
    n, err = scan(shim.pgt_slow5_scan, tmp_path / "hdr.blow5"); assert n == -1 and "truncated BLOW5 header" in err


def _zrec(rid, count_field, svb_payload):
    body = (struct.pack("<H", len(rid)) + rid + struct.pack("<I", 0) + struct.pack("<dddd", 2048.0, -240.0, 282.0, 4000.0)
            + struct.pack("<Q", 4 + len(svb_payload)) + struct.pack("<I", count_field) + svb_payload)
    z = zlib.compress(body)
    return struct.pack("<Q", len(z)) + z


def test_corrupt_streamvbyte_block(shim, tmp_path):
    hdr = b"#slow5_version\t0.2.0\n#num_read_groups\t1\n#char*\tuint32_t\tdouble\tdouble\tdouble\tdouble\tuint64_t\tint16_t*\n#read_id\tread_group\tdigitisation\toffset\trange\tsampling_rate\tlen_raw_signal\traw_signal\n"
    head = b"BLOW5\x01" + bytes([0, 2, 0]) + bytes([1]) + struct.pack("<I", 1) + bytes([1]) + bytes(64 - 15) + struct.pack("<I", len(hdr)) + hdr
    good = synth._svb_zd(np.arange(100, 140, dtype=np.int16))[4:]
    cases = {
        "count says 1000 values, the block holds 40": _zrec(b"r0", 1000, good),
        "control bytes ask for more data bytes than there are": _zrec(b"r0", 40, bytes([0xFF] * 10) + b"\x01\x02"),
        "count of 2^32 - 1": _zrec(b"r0", 2 ** 32 - 1, good),
        "block shorter than its own count field": struct.pack("<Q", 0),
    }
    for what, rec in cases.items():
        p = tmp_path / "svb.blow5"; p.write_bytes(head + rec + b"5WOLB")
        n, err = scan(shim.pgt_slow5_scan, p)
        assert n == -1 and ("corrupt streamvbyte block" in err or "corrupt BLOW5 record" in err or "zlib error" in err or "truncated" in err), (what, n, err)


@pytest.mark.parametrize("byte,val,msg", [(9, 3, "record compression other than none/zlib/zstd"), (14, 2, "signal compression other than none/svb-zd"), (9, 7, "record compression"), (14, 9, "signal compression")])
def test_blow5_header_announcing_unknown_compression_or_exzd(shim, batch, tmp_path, byte, val, msg):
    p = tmp_path / "a.blow5"; synth.write_blow5(batch, str(p), compress=True)
    d = bytearray(p.read_bytes()); d[byte] = val; p.write_bytes(d)
    n, err = scan(shim.pgt_slow5_scan, p)
    assert n == -1 and msg in err and "not supported" in err


def test_truncated_and_corrupt_bgzf(shim, batch, tmp_path):
    p = tmp_path / "t.bam"; synth.write_bam(batch, str(p), block_bytes=900)
    data = p.read_bytes()
    for cut in (len(data) - 28 - 10, len(data) // 2, len(data) // 3, 40, 17, 3):   # inside blocks, inside a block header
        q = tmp_path / "cut.bam"; q.write_bytes(data[:cut])
        n, err = scan(shim.pgt_sambam_scan, q)
        assert n == -1 and any(m in err for m in ("corrupt BGZF block", "truncated BAM", "zlib error in BGZF block", "not a BAM", "malformed SAM record")), (cut, n, err)  # (3 bytes are not recognisably BGZF: read as SAM text)
    d = bytearray(data); d[len(d) // 2] ^= 0xFF; d[len(d) // 2 + 1] ^= 0x55        # damaged deflate stream
    (tmp_path / "flip.bam").write_bytes(d)
    n, err = scan(shim.pgt_sambam_scan, tmp_path / "flip.bam")
    assert n == -1 and err
    d = bytearray(data); struct.pack_into("<H", d, 16, 5)                           # BSIZE smaller than the block header
    (tmp_path / "bsize.bam").write_bytes(d)
    n, err = scan(shim.pgt_sambam_scan, tmp_path / "bsize.bam"); assert n == -1 and err


def _bam_with_record(body: bytes) -> bytes:
    text = b"@HD\tVN:1.6\n"
    raw = b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", 0) + struct.pack("<i", len(body)) + body

    def block(data):
        c = zlib.compressobj(6, zlib.DEFLATED, -15); comp = c.compress(data) + c.flush()
        return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", 18 + len(comp) + 8 - 1) + comp
                + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))
    return block(raw) + block(b"")


def test_bam_record_fields_past_the_record(shim, tmp_path):
    name = b"r0\x00"
    def rec(l_seq, l_name=len(name), n_cigar=0, tags=b"", block_size=None):
        body = struct.pack("<iiBBHHHiiii", -1, -1, l_name, 0, 4680, n_cigar, 4, l_seq, -1, -1, 0) + name + bytes(2) + b"\xff" * 4 + tags
        return body
    cases = {
        "l_seq past the record": rec(10 ** 6),
        "l_seq negative": rec(-5),
        "l_read_name past the record": rec(4, l_name=250),
        "n_cigar_op past the record": rec(4, n_cigar=60000),
        "B tag whose count runs past the record": rec(4, tags=b"mvBc" + struct.pack("<i", 10 ** 6) + b"\x05\x01"),
        "Z tag without its terminator": rec(4, tags=b"xxZabc"),
        "tag of an unknown type": rec(4, tags=b"xx?abc"),
    }
    for what, body in cases.items():
        p = tmp_path / "rec.bam"; p.write_bytes(_bam_with_record(body))
        n, err = scan(shim.pgt_sambam_scan, p)
        assert n == -1 and ("corrupt BAM" in err or "unknown BAM tag" in err or "truncated BAM" in err), (what, n, err)
    p = tmp_path / "short.bam"; p.write_bytes(_bam_with_record(bytes(8)))   # block_size below the fixed fields
    n, err = scan(shim.pgt_sambam_scan, p); assert n == -1 and "truncated BAM record" in err


def test_reform_fails_cleanly_on_a_corrupt_bam(tmp_path):
    """a record-level failure of `reform` is `return -1` in the reference (src/reform.cpp:213-357: exit status 255), a file-level one exit(1)"""
    body = struct.pack("<iiBBHHHiiii", -1, -1, 3, 0, 4680, 0, 4, 10 ** 6, -1, -1, 0) + b"r0\x00" + bytes(6)
    p = tmp_path / "rec.bam"; p.write_bytes(_bam_with_record(body))
    r = subprocess.run([REFORM, "reform", "-k9", "-m0", "-c", str(p)], capture_output=True, text=True)
    assert r.returncode in (1, 255) and "corrupt BAM record" in r.stderr
    q = tmp_path / "cut.bam"; q.write_bytes(p.read_bytes()[:30])
    r = subprocess.run([REFORM, "reform", "-k9", "-m0", "-c", str(q)], capture_output=True, text=True)
    assert r.returncode in (1, 255) and ("BGZF" in r.stderr or "BAM" in r.stderr)


def test_random_damage_never_crashes(shim, batch, tmp_path):
    """2 000 damaged copies of valid files (byte flips, truncations, spliced garbage): any answer is fine, a crash is not (the sanitizer
    build turns an out-of-bounds read into one)."""
    rng = np.random.default_rng(11)
    synth.write_blow5(batch, str(tmp_path / "a.blow5")); synth.write_blow5(batch, str(tmp_path / "z.blow5"), compress=True)
    synth.write_table_files(batch, str(tmp_path / "t")); synth.write_bam(batch, str(tmp_path / "t.bam"), block_bytes=700)
    files = [("a.blow5", shim.pgt_slow5_scan), ("z.blow5", shim.pgt_slow5_scan), ("t.bam", shim.pgt_sambam_scan), ("t.sam", shim.pgt_sambam_scan)]
    for name, fn in files:
        data = (tmp_path / name).read_bytes()
        for it in range(500):
            d = bytearray(data)
            kind = it % 4
            if kind == 0:
                for _ in range(int(rng.integers(1, 6))):
                    d[int(rng.integers(0, len(d)))] = int(rng.integers(0, 256))
            elif kind == 1:
                d = d[:int(rng.integers(0, len(d)))]
            elif kind == 2:
                a = int(rng.integers(0, len(d))); d[a:a + int(rng.integers(1, 64))] = rng.integers(0, 256, int(rng.integers(1, 64)), dtype=np.uint8).tobytes()
            else:
                a = int(rng.integers(0, len(d) - 8)); struct.pack_into("<Q", d, a, int(rng.integers(0, 2 ** 63)) * 2 + 1)
            q = tmp_path / ("dmg_" + name); q.write_bytes(d)
            n, err = scan(fn, q)
            assert n >= -1 and (n >= 0 or err), (name, it, n, err)


def test_zstd_records_that_are_not_zstd_or_lie_about_their_size(shim, batch, tmp_path):
    """Record compression 2 (zstd; round 6): a header that says zstd over zlib records, a damaged frame, a truncated frame and a frame
    header announcing an absurd content size all end in the reader's message."""
    if synth.zstd_compress(b"x") is None:
        pytest.skip("no libzstd.so.1 on this machine")
    p = tmp_path / "z.blow5"; synth.write_blow5(batch, str(p), compress=True)
    d = bytearray(p.read_bytes()); d[9] = 2; p.write_bytes(d)                       # zlib records announced as zstd
    n, err = scan(shim.pgt_slow5_scan, p); assert n == -1 and "zstd error" in err, err
    synth.write_blow5(batch, str(p), compress="zstd")
    good = p.read_bytes()
    assert scan(shim.pgt_slow5_scan, p) == (6, "")
    hlen = struct.unpack_from("<I", good, 64)[0]; first = 68 + hlen
    sz = struct.unpack_from("<Q", good, first)[0]
    d = bytearray(good); d[first + 8 + sz // 2] ^= 0xFF; d[first + 8 + sz // 2 + 1] ^= 0xA5   # damaged frame body
    p.write_bytes(d); n, err = scan(shim.pgt_slow5_scan, p); assert n == -1 and ("zstd error" in err or "corrupt" in err), err
    d = bytearray(good); struct.pack_into("<Q", d, first, sz - 9)                   # frame cut short, the next "record" starts inside it
    p.write_bytes(d); n, err = scan(shim.pgt_slow5_scan, p); assert n == -1 and err
    # a frame whose header claims 2^40 bytes of content: magic, frame header descriptor 0xE0 (8-byte content size, single segment), the size
    lie = bytes.fromhex("28b52ffd") + bytes([0xE0]) + struct.pack("<Q", 1 << 40) + b"\x00" * 16
    d = bytearray(good[:first]) + struct.pack("<Q", len(lie)) + lie + b"5WOLB"
    p.write_bytes(d); n, err = scan(shim.pgt_slow5_scan, p); assert n == -1 and "zstd error" in err, err


def test_svbzd_deltas_that_overflow_int32(shim, tmp_path):
    """Crafted zig-zag deltas whose running sum leaves int32 (advisor r05): defined wrap-around, a result, no abort under UBSan."""
    hdr = b"#slow5_version\t0.2.0\n#num_read_groups\t1\n#char*\tuint32_t\tdouble\tdouble\tdouble\tdouble\tuint64_t\tint16_t*\n#read_id\tread_group\tdigitisation\toffset\trange\tsampling_rate\tlen_raw_signal\traw_signal\n"
    head = b"BLOW5\x01" + bytes([0, 2, 0]) + bytes([1]) + struct.pack("<I", 1) + bytes([1]) + bytes(64 - 15) + struct.pack("<I", len(hdr)) + hdr
    n_val = 8
    payload = bytes([0xFF, 0xFF]) + struct.pack("<8I", *([0xFFFFFFFE] * n_val))      # eight 4-byte values: zig-zag of +2^31 - 1, summed eight times
    p = tmp_path / "ovf.blow5"; p.write_bytes(head + _zrec(b"r0", n_val, payload) + b"5WOLB")
    n, err = scan(shim.pgt_slow5_scan, p)
    assert (n, err) == (1, "")
