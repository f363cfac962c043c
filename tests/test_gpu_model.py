"""pg_model (the k-mer model reduced on the device) against the CPU chain gmove oracle -> dump files -> model oracle
(oracle/model_oracle.c: tr | tail | datamash restated). Median and dwell texts must be identical; the sstdev text may
differ by one unit of its 14th significant digit (datamash sums left to right in long double, the device sums exact
integers)."""
import os
import subprocess
from decimal import Decimal

import numpy as np
import pytest

from helpers import oracle_for
from poregen_amd import synth
from poregen_amd.engine import GmoveEngine, GmoveParams, generate_kmers

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle", "model_oracle")
BIN = os.path.join(ROOT, "bin", "poregen")


def oracle_lines(dump_dir, mode, *args):
    out = subprocess.run([ORACLE, mode, str(dump_dir)] + list(args), capture_output=True, check=True).stdout.decode()
    return out.splitlines(keepends=True)


def same_number_14(a: str, b: str, ulps=1) -> bool:
    if a == b:
        return True
    if a in ("", "nan") or b in ("", "nan"):
        return False
    x, y = Decimal(a), Decimal(b)
    unit = Decimal(1).scaleb(max(x.adjusted(), y.adjusted()) - 13)
    return abs(x - y) <= unit * ulps


def exact_sstdev_text(dump_file, limit=None):
    """'%.14g' of the exact sample standard deviation of the file's values (all but the first), by rational arithmetic"""
    from decimal import getcontext
    from fractions import Fraction
    getcontext().prec = 80
    vals = [Decimal(x) for x in open(dump_file).read().replace(";", ",").strip(",").split(",")][1:]
    units = [int(v.scaleb(8)) for v in vals]
    n = len(units)
    s1 = sum(units); s2 = sum(u * u for u in units)
    var = Fraction(n * s2 - s1 * s1, n * (n - 1)) / 10**16
    sd = (Decimal(var.numerator) / Decimal(var.denominator)).sqrt()
    return sd


def compare_raw_model(mine: str, want_lines, dump_dir=None, limit=None):
    """k-mer and median columns identical. The stddev column is identical too, except where the left-to-right long double sums of
    datamash (restated by the oracle) lose the 14th digit (files whose values are nearly all equal: cancellation in x - mean):
    there the two texts differ by one unit of that digit and the DEVICE text is the correctly rounded exact value."""
    got = mine.splitlines(keepends=True)
    assert len(got) == len(want_lines)
    off = 0
    for g, w in zip(got, want_lines):
        gk, gm, gs = g.rstrip("\n").split("\t"); wk, wm, ws = w.rstrip("\n").split("\t")
        assert gk == wk and gm == wm, (g, w)
        if gs != ws:
            off += 1
            if limit is not None and (gs == limit or ws == limit):   # the two sides of the cap: both numbers sit at the limit
                assert dump_dir is not None
                exact = exact_sstdev_text(os.path.join(dump_dir, gk))
                assert abs(exact - Decimal(limit)) <= abs(exact) * Decimal("1e-13"), (g, w, exact)
                continue
            assert same_number_14(gs, ws), (g, w)
            if dump_dir is not None:
                exact = exact_sstdev_text(os.path.join(dump_dir, gk))
                unit = Decimal(1).scaleb(exact.adjusted() - 13)
                assert abs(Decimal(gs) - exact) <= unit / 2, ("device text is not the correctly rounded value", g, w, exact)
    if dump_dir is None:
        assert off <= max(2, len(got) // 50), f"{off} of {len(got)} stddev texts differ in the last digit"
    return off


def dump_from_oracle(tmp_path, o, kmers, name="dump"):
    d = tmp_path / name
    d.mkdir()
    for s, k in enumerate(kmers):
        (d / k).write_text(o.text(s))
    return d


@pytest.mark.parametrize("scaling,limit,kind", [(1, 100, "rna004"), (0, 37, "dna_r10"), (1, 1, "dna_r10")])
def test_model_equals_text_pipeline(tmp_path, scaling, limit, kind):
    rna = kind == "rna004"
    kmers = generate_kmers(5, rna=rna)
    p = dict(kmer_size=5, scaling=scaling, sample_limit=limit, rna=rna)
    b = synth.make_batch(1500, kind=kind, seed=77, indel_rate=0.02)
    o = oracle_for(kmers, **p); o.run_batch(b)
    d = dump_from_oracle(tmp_path, o, kmers)
    eng = GmoveEngine(GmoveParams(kmers=kmers, **p))
    eng.submit(b)
    m = eng.model()
    compare_raw_model(m.raw_model_lines(kmers, "3.1"), oracle_lines(d, "stats", "3.1"), d, "3.1")
    compare_raw_model(m.raw_model_lines(kmers, "0.5"), oracle_lines(d, "stats", "0.5"), d, "0.5")
    assert m.dwell_lines(kmers) == "".join(oracle_lines(d, "dwell"))
    # the same job in three batches: merged on the host, reduced from the re-uploaded arrays
    eng.reset()
    for lo, hi in ((0, 400), (400, 401), (401, 1500)):
        eng.submit(b.slice_reads(lo, hi))
    m3 = eng.model()
    assert m3.median_text == m.median_text and m3.sstdev_text == m.sstdev_text and m3.dwell_text == m.dwell_text
    assert np.array_equal(m3.sum1, m.sum1) and np.array_equal(m3.sum2_lo, m.sum2_lo) and np.array_equal(m3.n_values, m.n_values)
    # numeric fields agree with the texts
    has = m.n_values > 0
    assert np.allclose(m.median[has], [float(t) for t, h in zip(m.median_text, has) if h], rtol=1e-13, atol=0)
    # keep_first: one more value per non-empty file
    mk = eng.model(keep_first=True)
    res = eng.finish()
    per_slot = np.array([int(res.samp_off[int(res.ev_off[s + 1])] - res.samp_off[int(res.ev_off[s])]) for s in range(len(kmers))], dtype=np.uint64)
    assert np.array_equal(mk.n_values, per_slot) and np.array_equal(m.n_values, np.where(per_slot > 0, per_slot - 1, 0))
    for s in np.flatnonzero(per_slot)[:50]:
        v = np.sort(np.array([int(Decimal("%.8f" % x).scaleb(8)) for x in res.slot_values(int(s))], dtype=np.int64))
        n = v.size
        assert mk.mid_lo[s] == v[(n - 1) // 2] and mk.mid_hi[s] == v[n // 2]
        dd = (v - mk.origin[s]).astype(object)
        assert int(mk.sum1[s]) == sum(dd) and (int(mk.sum2_hi[s]) << 64) + int(mk.sum2_lo[s]) == sum(x * x for x in dd)
    eng.close()


def test_model_tiny_files_one_value_and_empty(tmp_path):
    """every event two samples long, one event per k-mer: `tail -n +2` leaves ONE value (sstdev nan); untouched k-mers are empty"""
    kmers = generate_kmers(5)
    b = synth.make_batch(4, kind="dna_r10", seed=5, read_len=4000)
    b.op_n[:] = 2
    p = dict(kmer_size=5, scaling=0, sample_limit=1, min_dur=1)
    o = oracle_for(kmers, **p); o.run_batch(b)
    d = dump_from_oracle(tmp_path, o, kmers)
    eng = GmoveEngine(GmoveParams(kmers=kmers, **p))
    eng.submit(b)
    m = eng.model()
    want = oracle_lines(d, "stats", "3.1")
    assert m.raw_model_lines(kmers, "3.1") == "".join(want)
    assert any(l.endswith("\tnan\n") for l in want) and any(l.endswith("\t\t\n") for l in want)
    assert m.dwell_lines(kmers) == "".join(oracle_lines(d, "dwell"))
    mk = eng.model(keep_first=True)
    assert set(np.unique(mk.n_values)) == {0, 2}
    eng.close()


@pytest.mark.parametrize("extra", [dict(scaling=1), dict(scaling=0, pa_min=100.0), dict(scaling=0, sig_move_offset=1)])
def test_model_large_slots_and_big_limit(tmp_path, extra):
    """a 64-k-mer list at sample_limit 4000: ~100 k values per slot (the 1024-thread kernel, candidates parked in LDS), even and
    odd counts; with pa_min = 100 about half of the samples are zero-filled: the spike at 0.0 overflows the candidate buffer and
    the generic select runs"""
    kmers = generate_kmers(3, rna=True)
    p = dict(kmer_size=3, sample_limit=4000, rna=True, min_dur=5, max_dur=70, **extra)
    b = synth.make_batch(4000, kind="rna004", seed=123)
    o = oracle_for(kmers, **p); o.run_batch(b)
    d = dump_from_oracle(tmp_path, o, kmers)
    eng = GmoveEngine(GmoveParams(kmers=kmers, **p))
    eng.submit(b)
    m = eng.model()
    assert int(m.n_values.max()) > 50000
    compare_raw_model(m.raw_model_lines(kmers, "3.1"), oracle_lines(d, "stats", "3.1"), d, "3.1")
    assert m.dwell_lines(kmers) == "".join(oracle_lines(d, "dwell"))
    eng.close()


def test_model_rejects_what_it_cannot_represent():
    """pa_max far above the fixed-point range: a kept sample of 5e7 pA must be an error, not a wrong number"""
    from poregen_amd.engine import PgError
    kmers = generate_kmers(5)
    b = synth.make_batch(20, kind="dna_r10", seed=9)
    b.range[:] = b.range * 1e6                 # pA values around 1e8
    eng = GmoveEngine(GmoveParams(kmers=kmers, kmer_size=5, scaling=0, sample_limit=5, pa_min=-1e300, pa_max=1e300))
    eng.submit(b)
    with pytest.raises(PgError) as ei:
        eng.model()
    assert "4e7" in str(ei.value)
    eng.close()


def test_cli_raw_model_and_dwell_model(tmp_path):
    b = synth.make_batch(300, kind="rna004", seed=31)
    pre = str(tmp_path / "in")
    synth.write_files(b, pre)
    out = tmp_path / "out"
    raw, dwell = tmp_path / "raw_model", tmp_path / "dwell_times"
    r = subprocess.run([BIN, "gmove", "-k", "5", "--rna", "--scaling", "1", "--file_limit", "1024", "--sample_limit", "50", "--min_dur", "19",
                        "--max_dur", "51", pre + ".slow5", pre + ".paf", str(out), "--fastq", pre + ".fastq", "--raw_model", str(raw),
                        "--dwell_model", str(dwell), "--stdv_limit", "0.9"], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    compare_raw_model(raw.read_text(), oracle_lines(out / "dump", "stats", "0.9"), out / "dump", "0.9")
    assert dwell.read_text() == "".join(oracle_lines(out / "dump", "dwell"))
    assert "\t0.9\n" in raw.read_text()
    # -d and --raw_model exclude each other; a bad limit is refused before any work
    r = subprocess.run([BIN, "gmove", "-k", "5", "-d", pre + ".slow5", pre + ".paf", str(tmp_path / "o2"), "--fastq", pre + ".fastq", "--raw_model", str(raw)], capture_output=True)
    assert r.returncode != 0 and b"cannot be combined with -d" in r.stderr
    r = subprocess.run([BIN, "gmove", "-k", "5", pre + ".slow5", pre + ".paf", str(tmp_path / "o3"), "--fastq", pre + ".fastq", "--raw_model", str(raw), "--stdv_limit", "abc"], capture_output=True)
    assert r.returncode != 0 and not (tmp_path / "o3").exists()


def test_model_device_on_the_writer_of_a_sharded_job():
    """Three ranks emulated on one GPU (count -> bases -> collect per shard); the writer's view of the job (per k-mer the ranks'
    kept events in rank order, as dist.gather_kept delivers it) reduced with pg_model_device equals the model of the unsharded
    job; empty tensors are accepted."""
    import torch
    from poregen_amd import dist as pgdist
    kmers = generate_kmers(5, rna=True)
    p = dict(kmer_size=5, scaling=1, sample_limit=60, rna=True)
    b = synth.make_batch(900, kind="rna004", seed=404, indel_rate=0.01)
    eng = GmoveEngine(GmoveParams(kmers=kmers, **p))
    eng.submit(b)
    whole = eng.model()
    shards = [b.slice_reads(0, 300), b.slice_reads(300, 650), b.slice_reads(650, 900)]
    engs = [GmoveEngine(GmoveParams(kmers=kmers, **p)) for _ in shards]
    cnts = [e.count(s) for e, s in zip(engs, shards)]
    base = np.zeros(len(kmers), np.uint64)
    for e, c in zip(engs, cnts):
        e.collect(base.copy()); base = base + c
    results = [e.finish() for e in engs]
    counts = sum(r.counts.astype(np.int64) for r in results)
    lens = np.concatenate([r.ev_len[int(r.ev_off[s]):int(r.ev_off[s + 1])] for s in range(len(kmers)) for r in results] + [np.zeros(0, np.uint32)])
    vals = np.concatenate([r.slot_values(s) for s in range(len(kmers)) for r in results] + [np.zeros(0)])
    dev = torch.device("cuda:0")
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).astype(dt)).to(dev)
    got = engs[0].model_device(t(counts, np.int64), t(lens, np.int32), t(vals, np.float64))
    assert got.median_text == whole.median_text and got.sstdev_text == whole.sstdev_text and got.dwell_text == whole.dwell_text
    assert np.array_equal(got.sum1, whole.sum1) and np.array_equal(got.mid_lo, whole.mid_lo)
    assert got.raw_model_lines(kmers) == whole.raw_model_lines(kmers)
    empty = engs[0].model_device(t(np.zeros(len(kmers)), np.int64), t(np.zeros(0), np.int32), t(np.zeros(0), np.float64))
    assert not empty.n_values.any() and all(x == "" for x in empty.median_text + empty.dwell_text)
    for e in engs + [eng]:
        e.close()


def test_model_rejects_a_spread_beyond_the_moment_sums():
    """values 20 000 pA apart inside one k-mer file: beyond the 2^40-unit deviation the exact moment sums accept"""
    from poregen_amd.engine import PgError
    kmers = generate_kmers(3)
    b = synth.make_batch(30, kind="dna_r10", seed=9)
    b.range[::2] = b.range[::2] * 250           # every other read: pA values around 25 000
    eng = GmoveEngine(GmoveParams(kmers=kmers, kmer_size=3, scaling=0, sample_limit=500, pa_min=-1e300, pa_max=1e300))
    eng.submit(b)
    with pytest.raises(PgError) as ei:
        eng.model()
    assert "2^40" in str(ei.value)
    eng.close()


def _units(x):
    return int(Decimal("%.8f" % float(x)).scaleb(8))


def test_model_one_wave_kernel_on_constructed_files():
    """Files built value by value for the one-wave kernels (<= 1024 values and < 256 events; <= 2048 and < 512): sizes around the lane / row /
    kernel boundaries, even
    and odd counts, duplicates, products x * 1e8 that land on a half (the tie the FMA's error decides), a spread beyond 2^31 units
    (64-bit selection keys), values of 2.2e7 and more (outside the 2^52 conversion: the file goes to the 256-thread kernel) and a file
    of 255 / 256 events (the last one-wave size / the first that is not). Middle order statistics and moments against Python integers."""
    import torch
    rng = np.random.default_rng(20251005)
    files = []   # (values, event lengths)
    def add(vals, lens=None):
        vals = np.asarray(vals, dtype=np.float64)
        if lens is None:
            k = max(1, len(vals) // 7)
            cuts = np.sort(rng.choice(np.arange(1, len(vals)), size=min(k, len(vals) - 1), replace=False)) if len(vals) > 1 else np.zeros(0, int)
            lens = np.diff(np.concatenate([[0], cuts, [len(vals)]]))
        assert int(np.sum(lens)) == len(vals)
        files.append((vals, np.asarray(lens, dtype=np.int64)))
    for n in (1, 2, 3, 63, 64, 65, 127, 128, 129, 700, 1023, 1024):
        add(rng.normal(0.0, 1.5, n))
    add(np.round(rng.normal(0.0, 1.0, 500), 2))                                   # many equal values
    add(np.full(300, 1.25))                                                       # one value
    add(np.concatenate([np.full(200, -0.5), np.full(200, 0.5)]))                  # the two middles on different values
    ties = (np.arange(1, 400, 2) + 0.5) * 1e-8                                    # x * 1e8 = k + 1/2 up to the product's rounding
    add(np.concatenate([ties, -ties]))
    add(rng.normal(0.0, 1.0, 600) * 40.0)                                         # spread > 2^31 units
    add(2.3e7 + rng.normal(0.0, 1.0, 100))                                        # beyond the 2^52 conversion
    add(np.concatenate([-3.9e7 + rng.normal(0.0, 50.0, 333), [-3.9e7 - 5000.0]]))
    add(np.concatenate([2.2e7 - 5000.0 + rng.normal(0.0, 1.0, 64), [2.2e7 + 1.0]]))       # ... by its last value only
    add(rng.normal(90.0, 10.0, 255 * 3), np.full(255, 3))                         # 255 events: one wave; first value dropped below
    add(rng.normal(90.0, 10.0, 256 * 3), np.full(256, 3))                         # 256 events: 256 threads
    add(rng.normal(0.0, 1.0, 1025))                                               # 1024 values after the first is dropped
    add(rng.normal(0.0, 1.0, 1026))                                               # 1025: the 32-row wave kernel
    for n in (1500, 2047, 2049, 2050, 3000):                                      # ... up to 2048 values, then 256 threads
        add(rng.normal(0.0, 2.0, n))
    add(rng.normal(90.0, 10.0, 511 * 3), np.full(511, 3))                         # 511 events: the last 32-row size
    add(rng.normal(90.0, 10.0, 512 * 3), np.full(512, 3))
    add(np.concatenate([rng.normal(0.0, 1.0, 1800), [9000.0]]) + 2.4e7)            # a 32-row file handed on
    add(rng.normal(0.0, 1.0, 5000))                                               # 1024 threads
    counts = np.array([len(l) for _, l in files], np.int64)
    lens = np.concatenate([l for _, l in files])
    vals = np.concatenate([v for v, _ in files])
    dev = torch.device("cuda:0")
    eng = GmoveEngine(GmoveParams(kmers=generate_kmers(5)[:len(files)], kmer_size=5, scaling=0, sample_limit=10))
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).astype(dt)).to(dev)
    for keep_first in (False, True):
        got = eng.model_device(t(counts, np.int64), t(lens, np.int32), t(vals, np.float64), keep_first=keep_first)
        for s, (v, l) in enumerate(files):
            u = [_units(x) for x in (v if keep_first else v[1:])]
            assert int(got.n_values[s]) == len(u), s
            if u:
                su = sorted(u); n = len(u)
                assert (int(got.mid_lo[s]), int(got.mid_hi[s])) == (su[(n - 1) // 2], su[n // 2]), (s, keep_first)
                assert int(got.origin[s]) == u[0], s
                d = [x - u[0] for x in u]
                assert int(got.sum1[s]) == sum(d), s
                assert (int(got.sum2_hi[s]) << 64) + int(got.sum2_lo[s]) == sum(x * x for x in d), s
            dw = sorted([int(x) - 1 for x in l] + [0])
            assert int(got.dwell_n[s]) == len(dw) and float(got.dwell_median[s]) == (dw[(len(dw) - 1) // 2] + dw[len(dw) // 2]) / 2.0, s
    eng.close()
