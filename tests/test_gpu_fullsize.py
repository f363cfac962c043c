"""Full-size runs (BASELINE.json shapes) checked through size-independent properties and through the oracle on
the prefix of the job that decides the output (the reference stops reading once every k-mer is complete)."""
import numpy as np
import pytest

import orc
from helpers import assert_result_equals_oracle, oracle_for
from poregen_amd import synth
from poregen_amd.engine import GmoveEngine, GmoveParams, generate_kmers

pytestmark = pytest.mark.gpu

_BIG = {}


def big_rna_batch():
    """400 000 RNA004 reads x 4 000 samples (BASELINE configs[2] / configs[4]'s size), generated once per test session."""
    if "b" not in _BIG:
        _BIG["b"] = synth.make_batch_fast(400000, kind="rna004", seed=20251003 + 2)
    return _BIG["b"]


def check_invariants(res, sample_limit, n_reads):
    assert np.all(res.counts <= sample_limit)
    assert np.array_equal(np.diff(res.ev_off.astype(np.int64)), res.counts.astype(np.int64))
    assert np.array_equal(np.diff(res.samp_off.astype(np.int64)), res.ev_len.astype(np.int64))
    assert res.samples.size == int(res.samp_off[-1]) and np.all(np.isfinite(res.samples))
    assert res.ev_read.size == 0 or int(res.ev_read.max()) < n_reads
    for s in np.flatnonzero(res.counts > 1)[:200]:   # PAF-line order inside every k-mer file
        r = res.ev_read[int(res.ev_off[s]):int(res.ev_off[s + 1])]
        assert np.all(np.diff(r.astype(np.int64)) >= 0)


def test_config1_full_size_rna_k5():
    """50 000 reads x 4 000 samples, k=5, --rna --scaling 1, dur 20/40, sample_limit 100, all 1024 k-mers."""
    b = synth.make_batch_fast(50000, kind="rna004", seed=20251003 + 1)
    p = dict(kmer_size=5, rna=True, scaling=1, min_dur=20, max_dur=40, sample_limit=100)
    kmers = generate_kmers(5, rna=True)
    eng = GmoveEngine(GmoveParams(kmers=kmers, **p))
    eng.submit(b)
    res = eng.finish()
    check_invariants(res, 100, b.n_reads)
    assert int(res.counts.sum()) == 1024 * 100 and eng.all_slots_full()
    # the oracle stops (like the reference, gmove.cpp:733-735) once all k-mers are complete: its output over the
    # prefix it reads is the output of the whole job
    o = oracle_for(kmers, **p)
    rcs = o.run_batch(b.slice_reads(0, 12000))
    assert rcs[-1] == orc.ORC_STOPPED
    assert_result_equals_oracle(res, o, check_text_slots=4, sample_limit=100)
    # same job in three uneven batches, and with lazy statistics: identical bits
    eng.reset()
    for lo, hi in ((0, 700), (700, 21000), (21000, 50000)):
        eng.submit(b.slice_reads(lo, hi))
    res2 = eng.finish()
    assert np.array_equal(res2.samples.view(np.uint64), res.samples.view(np.uint64)) and np.array_equal(res2.ev_off, res.ev_off)
    eng.close()
    lz = GmoveEngine(GmoveParams(kmers=kmers, lazy_stats=True, **p))
    lz.submit(b)
    res3 = lz.finish()
    assert np.array_equal(res3.samples.view(np.uint64), res.samples.view(np.uint64))
    lz.close()


def test_config3_dna_k9_all_slots():
    """DNA, k=9 (262 144 slots: generic radix-sort ranking), sample_limit 1000, homopolymer-rich reads."""
    b = synth.make_batch(3000, kind="dna_r10", seed=20251003 + 3, homopolymer_frac=0.1)
    p = dict(kmer_size=9, scaling=1, sample_limit=1000)
    kmers = generate_kmers(9)
    o = oracle_for(kmers, **p)
    o.run_batch(b)
    eng = GmoveEngine(GmoveParams(kmers=kmers, **p))
    eng.submit(b.slice_reads(0, 1234)); eng.submit(b.slice_reads(1234, 3000))
    res = eng.finish()
    eng.close()
    check_invariants(res, 1000, b.n_reads)
    oc = o.counts()
    assert np.array_equal(res.counts, oc)
    assert int((oc == 1000).sum()) >= 1          # the homopolymer k-mers reach the cap: the cut runs
    # EVERY slot's values and window lengths (the product's stream is k-mer-major: one comparison each)
    assert np.array_equal(res.ev_len, o.all_event_lens())
    assert np.array_equal(res.samples.view(np.uint64), o.all_values().view(np.uint64))


@pytest.mark.parametrize("dense", [False, True], ids=["sparse_gather", "chunked_gather"])
def test_k9_one_region_holds_most_events(dense, monkeypatch):
    """k = 9 with reads that are 70 % one long homopolymer: AAAAAAAAA alone accepts most events (cut at the cap in the middle of a tile),
    and its REGION of the two-level partition (high digit 0: the k-mers AAAAA....) holds more than half of all accepted events --
    region skew in k_part_bases / k_part_scatter / k_region_place, where the uniform workloads have 1/512 per region. All slots against
    the oracle (gmove.cpp:922-950)."""
    if dense:
        monkeypatch.setenv("PGMOVE_DENSE_MIN", "0")
    b = synth.make_batch(400, kind="dna_r10", seed=20251003 + 33)
    seq = b.seq.copy()
    for r in range(b.n_reads):
        a, e = int(b.seq_off[r]), int(b.seq_off[r + 1])
        seq[a + (e - a) // 10: a + (e - a) // 10 + (7 * (e - a)) // 10] = ord("A")
    b.seq = seq
    p = dict(kmer_size=9, scaling=1, sample_limit=1000)
    kmers = generate_kmers(9)
    o = oracle_for(kmers, **p)
    o.run_batch(b)
    big = dict(p); big["sample_limit"] = 2 ** 31 - 1
    ou = oracle_for(kmers, **big)
    ou.run_batch(b)
    acc = ou.counts().astype(np.int64)
    assert acc[:512].sum() > acc.sum() // 2 and acc[0] > 20 * 1000     # region 0 holds most accepted events, slot 0 is far above the cap
    eng = GmoveEngine(GmoveParams(kmers=kmers, **p))
    eng.submit(b.slice_reads(0, 150)); eng.submit(b.slice_reads(150, 400))
    res = eng.finish()
    eng.close()
    check_invariants(res, 1000, b.n_reads)
    assert np.array_equal(res.counts, o.counts()) and int(res.counts[0]) == 1000
    assert np.array_equal(res.ev_len, o.all_event_lens())
    assert np.array_equal(res.samples.view(np.uint64), o.all_values().view(np.uint64))


def test_config3_full_size_properties():
    """50 000 DNA reads, k=9: properties only at full size (counts vs limit, ordering, batch-split invariance)."""
    b = synth.make_batch_fast(50000, kind="dna_r10", seed=20251003 + 3, homopolymer_frac=0.1)   # SURVEY 8(d) cfg 3: 10 % homopolymer-rich reads
    p = dict(kmer_size=9, scaling=1, sample_limit=1000)
    kmers = generate_kmers(9)
    eng = GmoveEngine(GmoveParams(kmers=kmers, **p))
    eng.submit(b)
    res = eng.finish()
    check_invariants(res, 1000, b.n_reads)
    assert int((res.counts == 1000).sum()) >= 4      # the four homopolymer 9-mers (5 000 reads x 32 events each) sit at the cap
    # the oracle on the prefix that completes them (the slots that reach the cap inside 6 000 reads are final there)
    o = oracle_for(kmers, **p)
    o.run_batch(b.slice_reads(0, 6000))
    oc = o.counts()
    full = np.flatnonzero(oc == 1000)
    assert full.size >= 4
    for sl in full:
        assert np.array_equal(res.slot_values(int(sl)).view(np.uint64), o.values(int(sl)).view(np.uint64)), int(sl)
    for sl in np.flatnonzero(oc)[:: 997]:            # and a sample of the open ones: the oracle's values are a prefix of the job's
        ov = o.values(int(sl))
        assert np.array_equal(res.slot_values(int(sl))[:ov.size].view(np.uint64), ov.view(np.uint64)), int(sl)
    eng.reset()
    eng.submit(b.slice_reads(0, 20000)); eng.submit(b.slice_reads(20000, 50000))
    res2 = eng.finish()
    assert np.array_equal(res2.counts, res.counts) and np.array_equal(res2.samples.view(np.uint64), res.samples.view(np.uint64))
    eng.close()


def test_config2_large_limit_rank_shards():
    """BASELINE configs[2] at one GPU's size: 50 000 RNA reads, k=5, nearly every accepted event kept. The 8-rank job
    reaches sample_limit 5000 with 400 000 reads; 50 000 reads accept 2060-2240 events per k-mer, so the limit is
    scaled to 2100: about half of the k-mers hit the cap, the others do not. One run versus four contiguous 'rank'
    shards with the count exchange (pg_count -> bases -> pg_collect): the concatenation of the shards' streams in rank
    order is the single run, bit for bit."""
    b = synth.make_batch_fast(50000, kind="rna004", seed=20251003 + 2)
    p = dict(kmer_size=5, rna=True, scaling=1, min_dur=20, max_dur=40, sample_limit=2100)
    kmers = generate_kmers(5, rna=True)
    eng = GmoveEngine(GmoveParams(kmers=kmers, **p))
    eng.submit(b)
    res = eng.finish()
    eng.close()
    check_invariants(res, 2100, b.n_reads)
    assert int(res.counts.max()) == 2100 and int(res.counts.min()) < 2100
    bounds = [0, 9000, 25000, 25001, 50000]
    shards = [b.slice_reads(lo, hi) for lo, hi in zip(bounds[:-1], bounds[1:])]
    engs = [GmoveEngine(GmoveParams(kmers=kmers, **p)) for _ in shards]
    cnts = [e.count(s) for e, s in zip(engs, shards)]
    base = np.zeros_like(cnts[0])
    parts = []
    for e, c in zip(engs, cnts):
        e.collect(base.copy())
        parts.append(e.finish())
        base += c
        e.close()
    assert np.array_equal(np.minimum(base, 2100), res.counts)  # freq.txt of the job = min(sum of accepted, limit)
    assert sum(int(r.counts.sum()) for r in parts) == int(res.counts.sum())
    for s in list(range(0, 1024, 37)) + [int(np.argmax(res.counts)), int(np.argmin(res.counts))]:
        vals = np.concatenate([r.slot_values(s) for r in parts])
        assert np.array_equal(vals.view(np.uint64), res.slot_values(s).view(np.uint64)), s
    # the first shard's prefix against the oracle (slots that the oracle completes within that prefix)
    o = oracle_for(kmers, **p)
    o.run_batch(b.slice_reads(0, 3000))
    oc = o.counts()
    for s in range(0, 1024, 53):
        n = int(oc[s])
        ov = o.values(s)
        assert np.array_equal(res.slot_values(s)[:ov.size].view(np.uint64), ov.view(np.uint64)), s


def test_config4_whitelist_slice_indels_rank_shards():
    """BASELINE configs[4] at one GPU's size: 30 000 RNA reads whose ss strings carry 2 % deletions and 2 % insertions,
    a shuffled whitelist of 300 5-mers over ACGU of which the slice [51, 250] is dumped, --kmer_pick_margin 2. The
    reference reads every line here (a slice never completes the whole list), so the oracle runs on a prefix only for
    the k-mers it fills there; the whole job is checked through the rank-shard identity and the invariants."""
    rng = np.random.default_rng(4)
    full = generate_kmers(5, rna=True)
    wl = [full[i] for i in rng.permutation(len(full))[:300]]
    b = synth.make_batch(30000, kind="rna004", seed=20251003 + 4, indel_rate=0.04)
    p = dict(kmer_size=5, rna=True, scaling=1, min_dur=20, max_dur=40, kmer_pick_margin=2, sample_limit=100)
    kmers = wl[50:250]
    eng = GmoveEngine(GmoveParams(kmers=kmers, **p))
    eng.submit(b)
    res = eng.finish()
    eng.close()
    check_invariants(res, 100, b.n_reads)
    assert int(res.counts.min()) == 100                     # 30 000 reads fill every whitelisted k-mer
    # three uneven 'ranks' with the count exchange
    bounds = [0, 1000, 17000, 30000]
    engs = [GmoveEngine(GmoveParams(kmers=kmers, **p)) for _ in bounds[1:]]
    cnts = [e.count(b.slice_reads(lo, hi)) for e, lo, hi in zip(engs, bounds[:-1], bounds[1:])]
    base = np.zeros_like(cnts[0])
    parts = []
    for e, c in zip(engs, cnts):
        e.collect(base.copy()); parts.append(e.finish()); base += c; e.close()
    for s in range(len(kmers)):
        vals = np.concatenate([r.slot_values(s) for r in parts])
        assert np.array_equal(vals.view(np.uint64), res.slot_values(s).view(np.uint64)), s
    # the oracle on the first 4 000 reads: k-mers it completes there have their final content
    o = oracle_for(wl, index_start=51, index_end=250, **p)
    o.run_batch(b.slice_reads(0, 4000))
    oc = o.counts()
    done = np.flatnonzero(oc == 100)
    assert done.size > 20
    for s in done:
        assert np.array_equal(res.slot_values(int(s)).view(np.uint64), o.values(int(s)).view(np.uint64)), int(s)


def test_config2_full_size_8_rank_shards():
    """BASELINE configs[2] AS STATED: 400 000 RNA reads x 4 000 samples (1.6e9 samples), k=5, sample_limit 5000, sharded 8
    contiguous ways. The eight 'ranks' are eight contexts on this one GPU; every one counts its shard straight into its row
    of the all_gather receive buffer (pg_count with a device output) and collects with pg_collect_gathered(world=8), exactly
    the calls a rank of the 8-GPU job makes around its RCCL all_gather. Checked: the invariants, the concatenation of the
    ranks' per-k-mer streams in rank order == the single unsharded run bit for bit (src/gmove.cpp:925-950: the first
    sample_limit events in PAF-line order), freq.txt == min(sum of counts, limit), and the oracle on a prefix of the job for
    the leading events of every k-mer."""
    import torch
    from poregen_amd.dist import shard_bounds
    limit, world = 5000, 8
    b = big_rna_batch()
    p = dict(kmer_size=5, rna=True, scaling=1, min_dur=20, max_dur=40, sample_limit=limit)
    kmers = generate_kmers(5, rna=True)
    eng = GmoveEngine(GmoveParams(kmers=kmers, **p))
    eng.submit(b)
    res = eng.finish()
    eng.close()
    check_invariants(res, limit, b.n_reads)
    assert int(res.counts.min()) == limit and int(res.counts.sum()) == 1024 * limit   # 400 000 reads fill every 5-mer to the cap
    dev = torch.device("cuda:0")
    gather_buf = torch.zeros(world * len(kmers), dtype=torch.int64, device=dev)     # what ncclAllGather fills on every rank
    engs, keep_alive = [], []
    for g in range(world):
        lo, hi = shard_bounds(b.n_reads, world, g)
        e = GmoveEngine(GmoveParams(kmers=kmers, defer_stats=True, **p))
        sh = b.slice_reads(lo, hi)
        e.count(sh, out=gather_buf[g * len(kmers):(g + 1) * len(kmers)])
        e.sync()
        engs.append(e); keep_alive.append(sh)
    parts = []
    for g, e in enumerate(engs):
        e.stats()
        e.collect_gathered(gather_buf, world, g)
        parts.append(e.finish())
        tot, freq = (t.cpu().numpy() for t in e.job_totals())
        assert np.array_equal(freq.astype(np.uint64), res.counts) and np.array_equal(np.minimum(tot, limit), freq)
        e.close(); keep_alive[g] = None
    assert sum(int(r.counts.sum()) for r in parts) == int(res.counts.sum())
    lo_reads = [shard_bounds(b.n_reads, world, g)[0] for g in range(world)]
    for s in range(len(kmers)):
        vals = np.concatenate([r.slot_values(s) for r in parts])
        assert np.array_equal(vals.view(np.uint64), res.slot_values(s).view(np.uint64)), s
        a, e_ = int(res.ev_off[s]), int(res.ev_off[s + 1])
        reads = np.concatenate([r.ev_read[int(r.ev_off[s]):int(r.ev_off[s + 1])].astype(np.int64) + lo_reads[g] for g, r in enumerate(parts)])
        assert np.array_equal(reads, res.ev_read[a:e_].astype(np.int64)), s
    # the oracle on the first 4 000 reads: it fills no k-mer there, so its streams are the leading events of the job's
    o = oracle_for(kmers, **p)
    o.run_batch(b.slice_reads(0, 4000))
    oc = o.counts()
    assert int(oc.max()) < limit and int(oc.min()) > 0
    for s in range(len(kmers)):
        ov = o.values(s)
        assert np.array_equal(res.slot_values(s)[:ov.size].view(np.uint64), ov.view(np.uint64)), s
        assert np.array_equal(res.ev_len[int(res.ev_off[s]):int(res.ev_off[s]) + int(oc[s])], o.event_lens(s)), s


def test_sample_limit_5000_at_oracle_size():
    """configs[2]'s limit where the oracle can run the whole job: 9 000 RNA reads, k=3 (64 k-mers, ~8 000 accepted events
    each), sample_limit 5000 -- every k-mer reaches the cap; bit-exact against the oracle, one batch and three."""
    b = synth.make_batch_fast(9000, kind="rna004", seed=20251003 + 22)
    p = dict(kmer_size=3, rna=True, scaling=1, min_dur=20, max_dur=40, sample_limit=5000)
    kmers = generate_kmers(3, rna=True)
    o = oracle_for(kmers, **p)
    rcs = o.run_batch(b)
    assert int(o.counts().min()) == 5000 and rcs[-1] == orc.ORC_STOPPED
    eng = GmoveEngine(GmoveParams(kmers=kmers, stop_when_full=True, **p))
    eng.submit(b)
    assert_result_equals_oracle(eng.finish(), o, check_text_slots=2, sample_limit=5000)
    eng.reset()
    for lo, hi in ((0, 2500), (2500, 2501), (2501, 9000)):
        eng.submit(b.slice_reads(lo, hi))
    assert_result_equals_oracle(eng.finish(), o, check_text_slots=0, sample_limit=5000)
    eng.close()


def test_config4_full_size_8_rank_shards():
    """BASELINE configs[4] AS STATED: 400 000 RNA reads x 4 000 samples whose ss strings carry 2 % deletions (nD, n in 1..3) and 2 %
    insertions (nI, n in 5..40), a shuffled whitelist of 300 5-mers over ACGU of which the slice [51, 250] is dumped,
    --kmer_pick_margin 2, sharded 8 contiguous ways. Nearly every read holds an I or a D op, so all of them take the generic walk
    (src/gmove.cpp:204-211, 841-847). The eight 'ranks' are eight contexts on this one GPU around the calls of the 8-GPU job
    (pg_count into a row of the all_gather buffer, pg_collect_gathered(world=8)); the same input also goes through pg_job with
    devices [0] * 8 (what `poregen gmove --devices` runs). Checked: invariants, shard concatenation == the single run bit for bit,
    the job layer == the single run, the oracle on a prefix for the k-mers it completes there (a slice never completes the WHOLE
    list, so the reference reads every line and no prefix decides the whole output)."""
    import torch
    from poregen_amd.dist import shard_bounds
    from poregen_amd.engine import GmoveJob
    world = 8
    b = synth.add_indels_fast(big_rna_batch(), del_rate=0.02, ins_rate=0.02, seed=20251003 + 4)
    assert int((b.op_t == 1).sum()) > 500000 and int((b.op_t == 2).sum()) > 1000000
    rng = np.random.default_rng(4)
    full = generate_kmers(5, rna=True)
    wl = [full[i] for i in rng.permutation(len(full))[:300]]
    kmers = wl[50:250]
    p = dict(kmer_size=5, rna=True, scaling=1, min_dur=20, max_dur=40, kmer_pick_margin=2, sample_limit=100)
    eng = GmoveEngine(GmoveParams(kmers=kmers, **p))
    eng.submit(b)
    res = eng.finish()
    eng.close()
    check_invariants(res, 100, b.n_reads)
    assert int(res.counts.min()) == 100
    dev = torch.device("cuda:0")
    gather_buf = torch.zeros(world * len(kmers), dtype=torch.int64, device=dev)
    engs, keep_alive = [], []
    for g in range(world):
        lo, hi = shard_bounds(b.n_reads, world, g)
        e = GmoveEngine(GmoveParams(kmers=kmers, defer_stats=True, **p))
        sh = b.slice_reads(lo, hi)
        e.count(sh, out=gather_buf[g * len(kmers):(g + 1) * len(kmers)])
        e.sync()
        engs.append(e); keep_alive.append(sh)
    parts = []
    for g, e in enumerate(engs):
        e.stats()
        e.collect_gathered(gather_buf, world, g)
        parts.append(e.finish())
        tot, freq = (t.cpu().numpy() for t in e.job_totals())
        assert np.array_equal(freq.astype(np.uint64), res.counts) and np.array_equal(np.minimum(tot, 100), freq)
        e.close(); keep_alive[g] = None
    lo_reads = [shard_bounds(b.n_reads, world, g)[0] for g in range(world)]
    for s in range(len(kmers)):
        vals = np.concatenate([r.slot_values(s) for r in parts])
        assert np.array_equal(vals.view(np.uint64), res.slot_values(s).view(np.uint64)), s
        a, e_ = int(res.ev_off[s]), int(res.ev_off[s + 1])
        reads = np.concatenate([r.ev_read[int(r.ev_off[s]):int(r.ev_off[s + 1])].astype(np.int64) + lo_reads[g] for g, r in enumerate(parts)])
        assert np.array_equal(reads, res.ev_read[a:e_].astype(np.int64)), s
    assert sum(int(r.counts.sum()) for r in parts[1:]) == 0   # (the first shard's 50 000 reads fill the slice: the others place nothing)
    del parts
    # the same input through the job layer: eight shards on device 0, exchange through host memory
    job = GmoveJob(GmoveParams(kmers=kmers, **p), [0] * world)
    job.submit(b)
    rj = job.finish()
    for name in ("counts", "ev_off", "ev_len", "ev_read", "samp_off"):
        assert np.array_equal(getattr(rj, name), getattr(res, name)), name
    assert np.array_equal(rj.samples.view(np.uint64), res.samples.view(np.uint64))
    job.close()
    # the oracle on the first 4 000 reads: k-mers it completes there have their final content
    o = oracle_for(wl, index_start=51, index_end=250, **p)
    o.run_batch(b.slice_reads(0, 4000))
    oc = o.counts()
    done = np.flatnonzero(oc == 100)
    assert done.size > 20
    for s in done:
        assert np.array_equal(res.slot_values(int(s)).view(np.uint64), o.values(int(s)).view(np.uint64)), int(s)


@pytest.mark.parametrize("lanes", ["0", "1", "8"], ids=["gather_wave", "gather_evpair", "gather_chunks8"])
@pytest.mark.parametrize("extra", ["", "exact_samples", "unfused_cut"])
def test_k9_every_gather_form_and_both_cut_paths(lanes, extra, monkeypatch):
    """The three dense gathers (k_gather_wave, k_gather_evpair, the lane-group form) on the SAME k = 9 job -- the library picks one by the
    mean window, so the others would never see these inputs -- each also with the sample buffer sized exactly (its total then comes from
    the coarse chunk sums, in front of the gather) and with the sample_limit cut as a launch of its own (k_slot_cut instead of
    k_region_scan_cut). All slots against the oracle."""
    monkeypatch.setenv("PGMOVE_DENSE_MIN", "0")
    monkeypatch.setenv("PGMOVE_GATHER_LANES", lanes)
    if extra == "exact_samples":
        monkeypatch.setenv("PGMOVE_SAMPLES_EXACT", "1")
    if extra == "unfused_cut":
        monkeypatch.setenv("PGMOVE_NO_FUSED_CUT", "1")
    b = synth.make_batch(300, kind="dna_r10", seed=20251003 + 41, homopolymer_frac=0.2)
    p = dict(kmer_size=9, scaling=1, sample_limit=40)
    kmers = generate_kmers(9)
    o = oracle_for(kmers, **p)
    o.run_batch(b)
    eng = GmoveEngine(GmoveParams(kmers=kmers, **p))
    eng.submit(b.slice_reads(0, 100)); eng.submit(b.slice_reads(100, 300))
    res = eng.finish()
    eng.close()
    check_invariants(res, 40, b.n_reads)
    assert np.array_equal(res.counts, o.counts())
    assert np.array_equal(res.ev_len, o.all_event_lens())
    assert np.array_equal(res.samples.view(np.uint64), o.all_values().view(np.uint64))


@pytest.mark.parametrize("table", ["no_affine", "permuted_list", "sliced_list"])
def test_k9_slot_tables_that_are_looked_up(table, monkeypatch):
    """Since round 4 a generated k-mer list is COMPUTED (slot = code + constant); the look-up branch of k_events<2>, and the mixed case -- one
    table affine, the other looked up -- run only for other lists: PGMOVE_NO_AFFINE=1 (read per pg_create), a --kmer_file in another
    order, a slice whose other spelling's table is sparse. Dense kernels, all slots against the oracle."""
    monkeypatch.setenv("PGMOVE_DENSE_MIN", "0")
    b = synth.make_batch(200, kind="dna_r10", seed=20251003 + 42)
    full = generate_kmers(9)
    p = dict(kmer_size=9, scaling=1, sample_limit=30)
    if table == "no_affine":
        monkeypatch.setenv("PGMOVE_NO_AFFINE", "1")
        kmers, o = full, oracle_for(full, **p)
    elif table == "permuted_list":
        rng = np.random.default_rng(8)
        kmers = [full[i] for i in rng.permutation(len(full))[:60000]]
        o = oracle_for(kmers, **p)
    else:
        kmers = full[1000:200000]
        o = oracle_for(full, index_start=1001, index_end=200000, **p)
    o.run_batch(b)
    eng = GmoveEngine(GmoveParams(kmers=kmers, **p))
    eng.submit(b)
    res = eng.finish()
    eng.close()
    assert np.array_equal(res.counts, o.counts())
    assert np.array_equal(res.ev_len, o.all_event_lens())
    assert np.array_equal(res.samples.view(np.uint64), o.all_values().view(np.uint64))
