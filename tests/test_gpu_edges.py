"""Edge cases through the C ABI: empty and ragged batches, long reads, degenerate parameters."""
import numpy as np
import pytest

import orc
from helpers import assert_result_equals_oracle, oracle_for
from poregen_amd import synth
from poregen_amd.engine import Batch, GmoveEngine, GmoveParams, PgError, generate_kmers

pytestmark = pytest.mark.gpu


def concat(batches):
    def cat(name): return np.concatenate([getattr(b, name) for b in batches])
    def offs(name):
        out = [np.zeros(1, np.uint64)]; base = np.uint64(0)
        for b in batches:
            o = getattr(b, name); out.append(o[1:] + base); base = base + o[-1]
        return np.concatenate(out)
    return Batch(n_reads=sum(b.n_reads for b in batches), sig=cat("sig"), sig_off=offs("sig_off"), digitisation=cat("digitisation"),
                 offset=cat("offset"), range=cat("range"), query_start=cat("query_start"), target_start=cat("target_start"),
                 target_end=cat("target_end"), seq=cat("seq"), seq_off=offs("seq_off"), op_n=cat("op_n"), op_t=cat("op_t"), op_off=offs("op_off"))


def run(batches, kmers, **p):
    eng = GmoveEngine(GmoveParams(kmers=kmers, **p))
    for b in batches:
        eng.submit(b)
    res = eng.finish()
    eng.close()
    return res


def test_empty_batch_and_empty_between_batches():
    kmers = generate_kmers(5)
    p = dict(kmer_size=5, scaling=1, sample_limit=10)
    empty = synth.make_batch(0, kind="dna_r10")
    res = run([empty], kmers, **p)
    assert res.counts.sum() == 0 and res.samples.size == 0 and res.n_reads == 0
    b = synth.make_batch(60, kind="dna_r10", seed=31)
    o = oracle_for(kmers, **p); o.run_batch(b)
    res = run([b.slice_reads(0, 20), empty, b.slice_reads(20, 60), empty], kmers, **p)
    assert_result_equals_oracle(res, o, sample_limit=10)


def test_ragged_lengths_long_read_and_unaligned_signal_starts():
    """Reads of 41 ... 250 000 samples back to back (signal starts are not 16-byte aligned), one read with ~50 000 ops."""
    parts = [synth.make_batch(1, read_len=L, kind="dna_r10", seed=40 + i) for i, L in enumerate([41, 4000, 250000, 77, 1234, 9999, 16, 100003])]
    b = concat(parts)
    assert any(int(x) % 8 for x in b.sig_off[1:-1])
    kmers = generate_kmers(5)
    for p in (dict(kmer_size=5, scaling=1, sample_limit=1000), dict(kmer_size=5, scaling=0, sample_limit=3, margin=2, kmer_pick_margin=3)):
        o = oracle_for(kmers, **p); rcs = o.run_batch(b)
        assert set(rcs) <= {0, 1, 2}   # 2: every k-mer complete, the reference stops reading
        assert_result_equals_oracle(run([b], kmers, **p), o, sample_limit=p["sample_limit"])


def test_all_reads_skipped_and_short_sequences():
    b = synth.make_batch(30, kind="rna004", seed=50)
    # fetched sequence shorter than k for every read: target range of 3 bases
    b.target_start[:] = 3; b.target_end[:] = 0
    seq = np.concatenate([b.seq[int(b.seq_off[r]):int(b.seq_off[r]) + 3] for r in range(b.n_reads)])
    b = Batch(**{**b.__dict__, "seq": seq, "seq_off": (np.arange(b.n_reads + 1, dtype=np.uint64) * np.uint64(3))})
    kmers = generate_kmers(5, rna=True)
    res = run([b], kmers, kmer_size=5, rna=True, scaling=1)
    assert res.counts.sum() == 0 and np.all(res.read_skipped == 1)


@pytest.mark.parametrize("p", [
    dict(kmer_size=1, scaling=1, sample_limit=50, kmer_pick_margin=0),
    dict(kmer_size=12, scaling=0, sample_limit=2),
    dict(kmer_size=5, scaling=1, sample_limit=0),
    dict(kmer_size=5, scaling=1, sample_limit=10 ** 9),
    dict(kmer_size=5, scaling=1, sample_limit=4, min_dur=0, max_dur=10 ** 6),
])
def test_degenerate_parameters(p):
    b = synth.make_batch(80, kind="dna_r10", seed=60)
    k = p["kmer_size"]
    kmers = generate_kmers(k) if k < 12 else sorted({synth.seq_string(b, r)[i:i + k] for r in range(b.n_reads) for i in range(0, 200, 7)})
    o = oracle_for(kmers, **p); rcs = o.run_batch(b)
    assert set(rcs) <= {0, 1, 2}
    assert_result_equals_oracle(run([b], kmers, **p), o, sample_limit=p["sample_limit"])


def test_single_slot_and_whitelist_of_absent_kmers():
    b = synth.make_batch(100, kind="dna_r10", seed=61)
    for kmers in (["ACGTA"], ["NNNNN", "ACGUA", "ACGTA"]):   # k-mers that can never match get (empty) slots
        p = dict(kmer_size=5, scaling=1, sample_limit=7)
        o = oracle_for(kmers, **p); o.run_batch(b)
        assert_result_equals_oracle(run([b], kmers, **p), o, sample_limit=7)


def test_inputs_outside_the_references_defined_behaviour_are_errors():
    b = synth.make_batch(10, kind="dna_r10", seed=62)
    kmers = generate_kmers(5)
    bad = Batch(**{**b.__dict__, "query_start": b.query_start.copy()}); bad.query_start[4] = -1
    eng = GmoveEngine(GmoveParams(kmers=kmers, kmer_size=5))
    with pytest.raises(PgError) as ei:
        eng.submit(bad); eng.sync()
    assert ei.value.status == -3 and "read 4" in ei.value.text          # assert(query_start < len), gmove.cpp:752
    eng.close()
    ops = b.op_t.copy(); ops[int(b.op_off[2]) + 1] = 7                  # not ',', 'I' or 'D'
    eng = GmoveEngine(GmoveParams(kmers=kmers, kmer_size=5))
    with pytest.raises(PgError) as ei:
        eng.submit(Batch(**{**b.__dict__, "op_t": ops})); eng.sync()
    assert ei.value.status == -3 and "read 2" in ei.value.text
    eng.close()
    with pytest.raises(PgError):
        GmoveEngine(GmoveParams(kmers=kmers, kmer_size=5, kmer_pick_margin=-1))
    with pytest.raises(PgError):
        GmoveEngine(GmoveParams(kmers=kmers, kmer_size=5, sig_move_offset=6))


def test_margin_beyond_start_only_matters_for_kept_events():
    """--margin larger than an event's window start is undefined in the reference only where the window is printed
    (gmove.cpp:928-941), i.e. for an event that is KEPT. An accepted event whose k-mer is already complete is never
    looked at: with sample_limit 1 every later read starts with such events (window start 0 < margin) and the job is
    well defined. (Found by tools/fuzz_gpu.py: the check used to sit in front of the sample_limit cut.)"""
    p = dict(kmer_size=3, rna=True, scaling=1, sample_limit=1, kmer_pick_margin=1, min_dur=20, max_dur=70, margin=3)
    kmers = generate_kmers(3, rna=True)
    checked = 0
    for seed in range(40):
        b = synth.make_batch(7, read_len=3000, kind="rna004", seed=1000 + seed)
        o = oracle_for(kmers, **p)
        if min(o.run_batch(b)) < 0:  # the first read's own first kept event has start < margin: undefined, skip
            continue
        eng = GmoveEngine(GmoveParams(kmers=kmers, **p))
        eng.submit(b)
        res = eng.finish()
        eng.close()
        assert_result_equals_oracle(res, o, check_text_slots=2, sample_limit=1)
        checked += 1
    assert checked >= 5


def test_errors_behind_the_completing_read_do_not_count_for_the_whole_list():
    """The reference stops reading PAF lines once every k-mer of the WHOLE list is complete (gmove.cpp:733-735): a
    malformed line behind that point is never looked at. PG_FLAG_STOP_WHEN_FULL (stop_when_full) gives the batch API the
    same verdict; without it -- the reference's behaviour for a slice of the list, where its loop never ends early --
    the malformed read fails the job."""
    b = synth.make_batch(300, read_len=3000, kind="rna004", seed=77)
    p = dict(kmer_size=3, rna=True, scaling=1, sample_limit=5, min_dur=20, max_dur=40)
    kmers = generate_kmers(3, rna=True)
    bad = Batch(**{**b.__dict__, "query_start": b.query_start.copy()}); bad.query_start[250] = -1   # far behind completion
    o = oracle_for(kmers, **p)
    rcs = o.run_batch(bad)
    assert min(rcs) >= 0 and rcs[-1] == orc.ORC_STOPPED and len(rcs) < 250   # the oracle (whole list) never reaches read 250
    eng = GmoveEngine(GmoveParams(kmers=kmers, stop_when_full=True, **p))
    eng.submit(bad)
    res = eng.finish()
    assert_result_equals_oracle(res, o, check_text_slots=2, sample_limit=5)
    # the job was complete after that batch: a further batch is not looked at either, whatever it holds
    eng.submit(bad.slice_reads(240, 260))
    res2 = eng.finish()
    assert np.array_equal(res2.counts, res.counts) and np.array_equal(res2.samples.view(np.uint64), res.samples.view(np.uint64))
    eng.close()
    # a higher rank of a multi-GPU job whose lower ranks already completed every k-mer: its shard is never read
    e0 = GmoveEngine(GmoveParams(kmers=kmers, stop_when_full=True, **p))
    c0 = e0.count(b.slice_reads(0, 200)); e0.collect(np.zeros_like(c0)); assert e0.all_slots_full()
    e1 = GmoveEngine(GmoveParams(kmers=kmers, stop_when_full=True, **p))
    e1.count(bad.slice_reads(200, 300)); e1.collect(c0.copy())
    r1 = e1.finish()
    assert int(r1.counts.sum()) == 0
    e0.close(); e1.close()
    # the same malformed read INSIDE the part the reference reads: an error with or without the flag
    early = Batch(**{**b.__dict__, "query_start": b.query_start.copy()}); early.query_start[0] = -1
    for flag in (True, False):
        eng = GmoveEngine(GmoveParams(kmers=kmers, stop_when_full=flag, **p))
        with pytest.raises(PgError):
            eng.submit(early); eng.sync()
        eng.close()
    # a slice of the list: the reference reads every line, the oracle reports the malformed one, and so does the engine
    o2 = oracle_for(kmers, index_start=1, index_end=10, **p)
    assert min(o2.run_batch(bad)) < 0
    eng = GmoveEngine(GmoveParams(kmers=kmers[:10], **p))
    with pytest.raises(PgError) as ei:
        eng.submit(bad); eng.sync()
    assert "read 250" in ei.value.text
    eng.close()


# ---- round 2: device batches are never recognised by their address; pg_batch.n_ops is verified on the device -------------

def _dev_batches_same_shape():
    """Two DIFFERENT device batches with the same n_reads and the same array sizes except the ss ops: torch's caching
    allocator hands the second one the addresses of the first once that is freed."""
    b1 = synth.make_batch(120, kind="rna004", seed=501)
    b2 = synth.make_batch(120, kind="rna004", seed=502, indel_rate=0.05)  # more ss ops than b1 at the same n_reads
    assert int(b1.op_off[-1]) != int(b2.op_off[-1])
    return b1, b2


@pytest.mark.parametrize("known_n_ops", [True, False], ids=["n_ops_given", "n_ops_read_back"])
def test_device_batches_of_equal_n_reads_at_the_same_address(known_n_ops):
    import torch
    p = dict(kmer_size=5, rna=True, scaling=1, min_dur=20, max_dur=40, sample_limit=30)
    kmers = generate_kmers(5, rna=True)
    eng = GmoveEngine(GmoveParams(kmers=kmers, **p))
    dev = torch.device("cuda:0")
    # the offsets of every batch live in ONE device buffer: same address, same n_reads, different content -- what a caching
    # allocator produces for consecutive shards of equal shape
    op_off_buf = torch.empty(121, dtype=torch.int64, device=dev)
    for hb in (*_dev_batches_same_shape(), *_dev_batches_same_shape()[::-1]):   # small op count first, then large, then back
        o = oracle_for(kmers, **p); o.run_batch(hb)
        db = hb.to_device(dev)
        op_off_buf.copy_(db.op_off); db.op_off = op_off_buf
        torch.cuda.synchronize()
        if not known_n_ops:
            db.n_ops = 0
        eng.reset(); eng.submit(db)
        assert_result_equals_oracle(eng.finish(), o, sample_limit=30)
    eng.close()


@pytest.mark.parametrize("delta", [-7, 5, 4096])
def test_wrong_n_ops_of_a_device_batch_is_an_error_not_a_fault(delta):
    import torch
    from poregen_amd.engine import PgError
    p = dict(kmer_size=5, rna=True, scaling=1, min_dur=20, max_dur=40, sample_limit=30)
    kmers = generate_kmers(5, rna=True)
    hb = synth.make_batch(150, kind="rna004", seed=503)
    eng = GmoveEngine(GmoveParams(kmers=kmers, **p))
    db = hb.to_device(torch.device("cuda:0"))
    db.n_ops += delta
    with pytest.raises(PgError) as ei:
        eng.submit(db); eng.finish()
    assert ei.value.status == -2 and "n_ops" in ei.value.text          # PG_ERR_INVALID_ARG
    db.n_ops -= delta                                                  # the context is usable afterwards
    o = oracle_for(kmers, **p); o.run_batch(hb)
    eng.reset(); eng.submit(db)
    assert_result_equals_oracle(eng.finish(), o, sample_limit=30)
    eng.close()


def test_n_ops_too_large_with_the_ops_at_the_end_of_an_allocation():
    """pgmove.h: "no kernel touches memory behind n_ops" must hold for the ARRAYS too when the caller's n_ops is too large: op_t and
    op_n sit at the very end of allocations of their own (exactly one 2 MiB / 8 MiB segment each), so a kernel that trusted n_ops + 64
    before verifying it would read past the mapping. Expected: PG_ERR_INVALID_ARG, and the context stays usable."""
    import torch
    from poregen_amd.engine import PgError
    p = dict(kmer_size=5, rna=True, scaling=1, min_dur=20, max_dur=40, sample_limit=30)
    kmers = generate_kmers(5, rna=True)
    hb = synth.make_batch(150, kind="rna004", seed=505)
    dev = torch.device("cuda:0")
    db = hb.to_device(dev)
    n = int(hb.op_off[-1])
    seg_t = torch.empty(2 << 20, dtype=torch.uint8, device=dev)          # whole segments of the caching allocator
    seg_n = torch.empty((8 << 20) // 4, dtype=torch.int32, device=dev)
    pad = (n + 15) // 16 * 16                                            # the arrays keep their 16-byte alignment
    t_view, n_view = seg_t[-pad:][:n], seg_n[-pad:][:n]
    t_view.copy_(db.op_t[:n]); n_view.copy_(db.op_n[:n])
    db.op_t, db.op_n = t_view, n_view
    torch.cuda.synchronize()
    eng = GmoveEngine(GmoveParams(kmers=kmers, **p))
    db.n_ops = n + 64
    with pytest.raises(PgError) as ei:
        eng.submit(db); eng.finish()
    assert ei.value.status == -2 and "n_ops" in ei.value.text
    db.n_ops = n
    o = oracle_for(kmers, **p); o.run_batch(hb)
    eng.reset(); eng.submit(db)
    assert_result_equals_oracle(eng.finish(), o, sample_limit=30)
    eng.close()


def test_sample_limit_zero_never_completes_a_kmer():
    """gmove.cpp:925-927 skips every event at limit 0 before 945-950 could count it: no k-mer ever completes, the loop never
    ends early, every read is looked at (and can fail the job), every read gets its ':' with -d."""
    from poregen_amd.engine import PgError
    p = dict(kmer_size=3, rna=True, scaling=1, min_dur=20, max_dur=40, sample_limit=0)
    kmers = generate_kmers(3, rna=True)
    hb = synth.make_batch(60, kind="rna004", seed=504)
    o = oracle_for(kmers, delimit=True, **p)
    assert all(rc == 0 for rc in o.run_batch(hb))                      # never ORC_STOPPED
    eng = GmoveEngine(GmoveParams(kmers=kmers, stop_when_full=True, **p))
    for lo in range(0, 60, 20):
        eng.submit(hb.slice_reads(lo, lo + 20))
        assert not eng.all_slots_full()
    res = eng.finish()
    assert int(res.counts.sum()) == 0
    assert_result_equals_oracle(res, o, delimit=True, sample_limit=0)
    # an RNA record without --rna still fails the job although "all k-mers hold sample_limit events"
    eng2 = GmoveEngine(GmoveParams(kmers=generate_kmers(3), stop_when_full=True, **dict(p, rna=False)))
    with pytest.raises(PgError) as ei:
        eng2.submit(hb.slice_reads(0, 20)); eng2.finish()
    assert ei.value.status == -4
    eng.close(); eng2.close()


def _tiny_reads_batch(n_reads, seed, k, rna, max_ops=6, op_len=(18, 45), indel_every=0):
    """Reads of 1 .. max_ops ss ops (matches; every indel_every-th read gets an insertion): several reads inside one group of
    four op indices, more than a thousand reads inside one tile of 4096 -- the rare paths of the op-parallel event kernel."""
    rng = np.random.default_rng(seed)
    sig, sig_off, seq, seq_off, op_n, op_t, op_off = [], [0], [], [0], [], [], [0]
    ts, te, qs = [], [], []
    for r in range(n_reads):
        nm = int(rng.integers(1, max_ops + 1))
        ops = [(int(rng.integers(*op_len)), 0) for _ in range(nm)]
        if indel_every and r % indel_every == 1 and nm >= 2:
            ops.insert(1, (int(rng.integers(3, 9)), 1))
        L = sum(n for n, _ in ops) + int(rng.integers(0, 5))
        raw = np.clip(np.rint(rng.normal(900, 60, L)), 300, 1500).astype(np.int16)
        bases = rng.integers(0, 4, nm)
        s = synth.BASES[bases[::-1]] if rna else synth.BASES[bases]
        sig.append(raw); sig_off.append(sig_off[-1] + L); seq.append(s); seq_off.append(seq_off[-1] + nm)
        op_n += [n for n, _ in ops]; op_t += [t for _, t in ops]; op_off.append(op_off[-1] + len(ops))
        ts.append(nm if rna else 0); te.append(0 if rna else nm); qs.append(0)
    n = n_reads
    return Batch(n_reads=n, sig=np.concatenate(sig), sig_off=np.asarray(sig_off, np.uint64), digitisation=np.full(n, 2048.0),
                 offset=np.full(n, -240.0), range=np.full(n, 281.0), query_start=np.asarray(qs, np.int32), target_start=np.asarray(ts, np.int32),
                 target_end=np.asarray(te, np.int32), seq=np.concatenate(seq).astype(np.uint8), seq_off=np.asarray(seq_off, np.uint64),
                 op_n=np.asarray(op_n, np.uint32), op_t=np.asarray(op_t, np.uint8), op_off=np.asarray(op_off, np.uint64)).validate_host()


@pytest.mark.parametrize("k,rna,max_ops,indel_every", [(1, False, 3, 0), (2, True, 4, 0), (3, False, 6, 0), (2, False, 5, 7), (1, True, 2, 0)])
def test_tiny_reads_many_per_group_and_per_tile(k, rna, max_ops, indel_every):
    _tiny_reads_case(k, rna, max_ops, indel_every, 0)


@pytest.mark.parametrize("k,rna,max_ops,indel_every,move_offset", [(2, True, 4, 0, 1), (3, False, 6, 0, 0), (2, False, 5, 7, 2), (6, False, 9, 0, 1), (6, True, 8, 5, 0)])
def test_tiny_reads_through_the_dense_kernels(k, rna, max_ops, indel_every, move_offset, monkeypatch):
    """The same with PGMOVE_DENSE_MIN=0: k_rank_emit2 (its LDS table of a tile's first 128 reads, events that name a read beyond it or none
    at all) and, for k = 6, the partitioned ranking, with a window that is the one of match i + sig_move_offset (found by the fuzzer:
    an event naming read 129 of its tile read the table out of bounds)."""
    monkeypatch.setenv("PGMOVE_DENSE_MIN", "0")
    _tiny_reads_case(k, rna, max_ops, indel_every, move_offset)


def _tiny_reads_case(k, rna, max_ops, indel_every, move_offset):
    """Reads far shorter than a 256-op block: three or four reads inside one thread's group of four ops, > 1024 reads inside one
    tile of 4096 ops. Direct path, the forced generic path and the oracle agree bit for bit."""
    import torch
    b = _tiny_reads_batch(9000, 700 + k, k, rna, max_ops=max_ops, indel_every=indel_every)
    limit = 1000000  # no k-mer ever completes: every read of the batch reaches the output
    p = dict(kmer_size=k, rna=rna, scaling=1, sample_limit=limit, kmer_pick_margin=0, min_dur=5, max_dur=70, sig_move_offset=move_offset)
    kmers = generate_kmers(k, rna=rna)
    o = oracle_for(kmers, **p)
    rcs = o.run_batch(b)
    assert min(rcs) >= 0 and max(rcs) <= 1, "the generator must stay inside the reference's defined behaviour"
    for split in (False, True):
        for dev in (False, True):
            eng = GmoveEngine(GmoveParams(kmers=kmers, split_walk=split, **p))
            eng.submit(b.to_device(torch.device("cuda:0")) if dev else b)
            assert_result_equals_oracle(eng.finish(), o, check_text_slots=2, sample_limit=limit)
            eng.close()


@pytest.mark.parametrize("lanes", ["0", "1", "8"])
def test_calibration_outside_the_reciprocal_domain_still_divides_exactly(monkeypatch, lanes):
    """The dense gathers replace (x - median) / MAD by a reciprocal taken once per event (pg_select.h: pg_div_by_recip), proved exact
    for normal numbers end to end. A read whose calibration leaves that domain (offset 1e-300: pA values and differences of 1e-300 next
    to a MAD of 148) must switch the batch to the division itself: the doubles stay the oracle's bit for bit."""
    monkeypatch.setenv("PGMOVE_DENSE_MIN", "0")
    monkeypatch.setenv("PGMOVE_GATHER_LANES", lanes)
    b = synth.make_batch(40, kind="dna_r10", seed=77)
    r = 7
    a, e = int(b.sig_off[r]), int(b.sig_off[r + 1])
    n = e - a
    sig = b.sig.copy()
    q = max(2, n // 50)
    n_neg = n // 2 - q - q // 2
    v = np.concatenate([np.full(n_neg, -100), np.full(q, 30000), np.full(q, 0), np.full(n - n_neg - 2 * q, 100)]).astype(np.int16)
    v = v[np.random.default_rng(3).permutation(n)]  # 0: pA = 1e-300, the read's median; 30000: above pa_max, zero-filled, x - median = -1e-300
    # the median must be one of the tiny values: as many samples below them as above (the zero-filled ones count below)
    lo = int((v == -100).sum() + (v == 30000).sum()); hi = int((v == 100).sum())
    assert lo <= n // 2 < lo + q, (lo, hi, q, n)
    sig[a:e] = v
    dig = b.digitisation.copy(); off = b.offset.copy(); rg = b.range.copy()
    dig[r] = 1.0; off[r] = 1e-300; rg[r] = 1.0
    b = Batch(**{**b.__dict__, "sig": sig, "digitisation": dig, "offset": off, "range": rg})
    kmers = generate_kmers(5)
    p = dict(kmer_size=5, scaling=1, sample_limit=10 ** 6, pa_min=-180.0, pa_max=180.0, min_dur=1, max_dur=10 ** 5)
    o = oracle_for(kmers, **p); o.run_batch(b)
    eng = GmoveEngine(GmoveParams(kmers=kmers, **p))
    eng.submit(b)
    res = eng.finish()
    eng.close()
    assert_result_equals_oracle(res, o, sample_limit=p["sample_limit"])
    vals = np.concatenate([o.values(s) for s in range(o.n_slots) if o.counts()[s]])
    assert np.any((vals != 0) & (np.abs(vals) < 1e-290)), "the case must reach the tiny quotients"


def _long_batch(seed=91):
    """reads of 3 000 ... 1 000 000 samples (the 10^6 one and three more above the split threshold), ragged signal starts"""
    L = np.array([3000, 40001, 1_000_000, 4000, 32768, 32769, 250_000, 77, 65_537], np.int64)
    return synth.make_ragged_fast(L, kind="dna_r10", seed=seed)


@pytest.mark.parametrize("mode", ["default", "one_stream", "device_batch", "few_helpers", "no_split"])
def test_long_reads_split_across_waves_equal_the_oracle(monkeypatch, mode):
    """A read above 32 768 samples is binned by several waves (PgLongState: the read's own wave + helpers, histograms summed in global
    memory, the last slice runs the selection). Median / MAD and every kept sample stay the oracle's bit for bit: with room for every
    helper, with room for some (the others' reads stay on one wave), with none, fed from the host and from the device."""
    if mode == "few_helpers":
        monkeypatch.setenv("PGMOVE_LONG_HELPERS", "70")   # the 10^6-sample read wants 61, the 250 000 one 15, 65 537: 4, 40 001: 2
    if mode == "no_split":
        monkeypatch.setenv("PGMOVE_NO_LONG_SPLIT", "1")
    b = _long_batch()
    kmers = generate_kmers(5)
    p = dict(kmer_size=5, scaling=1, sample_limit=3000)
    o = oracle_for(kmers, **p); o.run_batch(b)
    q = dict(p, profile=True) if mode != "default" else p
    if mode == "one_stream":
        q = dict(p, overlap=False)
    eng = GmoveEngine(GmoveParams(kmers=kmers, **q))
    if mode == "device_batch":
        import torch
        eng.submit(b.to_device(torch.device("cuda", 0)))
    else:
        eng.submit(b)
    res = eng.finish()
    st = eng.kernel_stats()
    eng.close()
    assert_result_equals_oracle(res, o, sample_limit=p["sample_limit"])
    split = st.get("long_reads_split", (0, 0.0))[0]
    if mode == "no_split":
        assert split == 0
    elif mode == "few_helpers":
        assert 1 <= split < 5 and st["long_helpers_short_batches"][0] == 1
    else:
        assert split == 5, st   # 40 001, 10^6, 32 769, 250 000, 65 537 (32 768 itself is not above the threshold)


def test_long_histograms_are_zero_before_the_statistics_stream_uses_them():
    """Advisor r05 (high): in the default two-stream mode the zero-fill of a freshly grown long-read histogram buffer was queued on the chain's
    stream while the statistics that add into it ran on the second one. Several thousand helpers (a fill of tens of MB) from a HOST batch, then a
    second batch that reuses the buffer as the first left it: medians, MADs and every kept sample are the oracle's."""
    L = np.full(400, 200_000, np.int64); L[::7] = 150_001; L[3::11] = 40_000          # 12 / 9 / 2 helper slices per read: ~4 300 helpers
    b1 = synth.make_ragged_fast(L, kind="dna_r10", seed=61)
    b2 = synth.make_ragged_fast(L[::-1].copy(), kind="dna_r10", seed=62)
    kmers = generate_kmers(5)
    p = dict(kmer_size=5, scaling=1, sample_limit=4000)
    o = oracle_for(kmers, **p); o.run_batch(b1); o.run_batch(b2)
    eng = GmoveEngine(GmoveParams(kmers=kmers, **p))                                   # the library's default: PG_FLAG_OVERLAP
    eng.submit(b1); eng.submit(b2)
    res = eng.finish()
    st = eng.kernel_stats()
    eng.close()
    assert_result_equals_oracle(res, o, sample_limit=p["sample_limit"])
    assert st.get("long_reads_split", (0, 0.0))[0] == 2 * L.size and st.get("long_helpers_short_batches", (0, 0.0))[0] == 0, st


def test_back_to_back_steps_with_long_reads_keep_their_helper_counters_apart():
    """The bench loop's shape (reset + submit again and again, no host wait in between) on a batch with long reads, two streams: the helper
    reservations of step i + 1 (statistics stream) and the zeroing by step i's k_batch_init (chain's stream) are only ordered through the
    gather two steps back -- round 6 keeps the counters in a ring of four so that this is enough. The last step's result is the oracle's."""
    import torch
    L = np.array([3000, 70_000, 5000, 300_000, 40_001, 4000, 120_000, 33_000, 2500, 65_537] * 3, np.int64)
    b = synth.make_ragged_fast(L, kind="dna_r10", seed=77)
    kmers = generate_kmers(5)
    p = dict(kmer_size=5, scaling=1, sample_limit=2000)
    o = oracle_for(kmers, **p); o.run_batch(b)
    eng = GmoveEngine(GmoveParams(kmers=kmers, **p))
    d = b.to_device(torch.device("cuda", 0))
    for _ in range(40):
        eng.reset(); eng.submit(d)
    res = eng.finish()
    st = eng.kernel_stats()
    eng.close()
    assert_result_equals_oracle(res, o, sample_limit=p["sample_limit"])
    assert st.get("long_helpers_short_batches", (0, 0.0))[0] == 0, st
