"""GPU parity: libpgmove (through the C ABI) vs the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest

from helpers import assert_result_equals_oracle, oracle_for
from poregen_amd import synth
from poregen_amd.engine import GmoveEngine, GmoveParams, generate_kmers

pytestmark = pytest.mark.gpu


def run_engine(batches, **p):
    eng = GmoveEngine(GmoveParams(**p))
    for b in batches:
        eng.submit(b)
    res = eng.finish()
    eng.close()
    return res


CASES = [
    # BASELINE.json config 1 shape at oracle-friendly size: RNA004, k=5, med-MAD, dur 20/40
    dict(name="rna_k5_medmad", kind="rna004", n=600, gen={}, p=dict(kmer_size=5, rna=True, scaling=1, min_dur=20, max_dur=40, sample_limit=100)),
    dict(name="rna_k5_noscale_limit7", kind="rna004", n=300, gen={}, p=dict(kmer_size=5, rna=True, scaling=0, min_dur=20, max_dur=40, sample_limit=7)),
    # DNA, defaults of the reference except k
    dict(name="dna_k5_medmad", kind="dna_r10", n=300, gen={}, p=dict(kmer_size=5, scaling=1, sample_limit=1000)),
    dict(name="dna_k9", kind="dna_r10", n=200, gen=dict(homopolymer_frac=0.1), p=dict(kmer_size=9, scaling=1, sample_limit=3)),
    # indels + pick margin (config 4 shape)
    dict(name="rna_indel_margin2", kind="rna004", n=300, gen=dict(indel_rate=0.02), p=dict(kmer_size=5, rna=True, scaling=1, min_dur=20, max_dur=40, kmer_pick_margin=2)),
    dict(name="dna_indel_margin0", kind="dna_r10", n=200, gen=dict(indel_rate=0.03), p=dict(kmer_size=6, scaling=1, kmer_pick_margin=0, sample_limit=50)),
    dict(name="dna_indel_margin5_off1", kind="dna_r10", n=200, gen=dict(indel_rate=0.03), p=dict(kmer_size=6, scaling=0, kmer_pick_margin=5, sig_move_offset=1, sample_limit=50)),
    # zero-fill quirk in the middle of the value range, print margin
    dict(name="rna_pamin100", kind="rna004", n=200, gen={}, p=dict(kmer_size=5, rna=True, scaling=1, pa_min=100.0, min_dur=20, max_dur=40)),
    dict(name="dna_margin3", kind="dna_r10", n=150, gen={}, p=dict(kmer_size=5, scaling=1, margin=3, sample_limit=20)),
]


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_batch_matches_oracle(case):
    b = synth.make_batch(case["n"], kind=case["kind"], seed=20251003 + len(case["name"]), **case["gen"])
    p = case["p"]
    kmers = generate_kmers(p["kmer_size"], rna=p.get("rna", False))
    o = oracle_for(kmers, **p)
    rcs = o.run_batch(b)
    assert all(rc in (0, 1, 2) for rc in rcs), rcs  # 2 = every k-mer complete, the reference stops reading
    res = run_engine([b], kmers=kmers, **p)
    assert_result_equals_oracle(res, o, sample_limit=p.get("sample_limit", 100))


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_batch_matches_oracle_through_the_dense_kernels(case, monkeypatch):
    """The same cases with PGMOVE_DENSE_MIN=0: however few events may be kept, the job takes the kernels built for MANY kept events
    (k_rank_emit2: four tiles per workgroup, packed counters; the chunked gather with its offset scan inside), and -- PGMOVE_LSD_SORT=1
    on the k = 9 case -- the round-1 LSD radix sort that stays for more than 2^20 slots."""
    monkeypatch.setenv("PGMOVE_DENSE_MIN", "0")
    b = synth.make_batch(case["n"], kind=case["kind"], seed=20251003 + len(case["name"]), **case["gen"])
    p = case["p"]
    kmers = generate_kmers(p["kmer_size"], rna=p.get("rna", False))
    o = oracle_for(kmers, **p)
    o.run_batch(b)
    res = run_engine([b.slice_reads(0, case["n"] // 3), b.slice_reads(case["n"] // 3, case["n"])], kmers=kmers, **p)
    assert_result_equals_oracle(res, o, sample_limit=p.get("sample_limit", 100))
    if p["kmer_size"] == 9:
        monkeypatch.setenv("PGMOVE_LSD_SORT", "1")
        res = run_engine([b], kmers=kmers, **p)
        assert_result_equals_oracle(res, o, sample_limit=p.get("sample_limit", 100))


def test_multi_batch_equals_single_batch_and_oracle():
    b = synth.make_batch(500, kind="rna004", seed=11)
    p = dict(kmer_size=5, rna=True, scaling=1, min_dur=20, max_dur=40, sample_limit=40)
    kmers = generate_kmers(5, rna=True)
    o = oracle_for(kmers, **p)
    o.run_batch(b)
    parts = [b.slice_reads(0, 100), b.slice_reads(100, 101), b.slice_reads(101, 380), b.slice_reads(380, 500)]
    res = run_engine(parts, kmers=kmers, **p)
    assert_result_equals_oracle(res, o, sample_limit=40)


def test_kmer_whitelist_slice_and_delimiters():
    # config 4: shuffled whitelist over ACGU, slice [51, 250] of it
    rng = np.random.default_rng(5)
    full = generate_kmers(5, rna=True)
    wl = [full[i] for i in rng.permutation(len(full))[:300]]
    b = synth.make_batch(250, kind="rna004", seed=12, indel_rate=0.02)
    p = dict(kmer_size=5, rna=True, scaling=1, min_dur=20, max_dur=40, kmer_pick_margin=2, sample_limit=5)
    o = oracle_for(wl, index_start=51, index_end=250, delimit=True, **p)
    o.run_batch(b)
    res = run_engine([b], kmers=wl[50:250], **p)
    assert_result_equals_oracle(res, o, delimit=True, sample_limit=5)


def test_two_phase_shards_equal_single_run():
    """pg_count / pg_collect with explicit bases: two contiguous shards on one GPU reproduce the whole."""
    b = synth.make_batch(400, kind="rna004", seed=13)
    p = dict(kmer_size=5, rna=True, scaling=1, min_dur=20, max_dur=40, sample_limit=30)
    kmers = generate_kmers(5, rna=True)
    o = oracle_for(kmers, **p)
    o.run_batch(b)
    shards = [b.slice_reads(0, 170), b.slice_reads(170, 400)]
    engs = [GmoveEngine(GmoveParams(kmers=kmers, **p)) for _ in shards]
    cnts = [e.count(s) for e, s in zip(engs, shards)]
    base = np.zeros_like(cnts[0])
    results = []
    for e, c in zip(engs, cnts):
        e.collect(base.copy())
        results.append(e.finish())
        base += c
    total = sum(int(r.counts.sum()) for r in results)
    assert total == int(o.counts().sum())
    for s in range(len(kmers)):
        vals = np.concatenate([r.slot_values(s) for r in results])
        assert np.array_equal(vals.view(np.uint64), o.values(s).view(np.uint64))
    for e in engs:
        e.close()


def test_batches_merged_on_the_device(monkeypatch):
    """Several batches: their kept samples stay on the device and pg_finish merges them there (one download). Small jobs take the host
    merge by default; PGMOVE_HOLD_MIN_BYTES=1 sends this one through the device path -- same result as one batch, and as the oracle."""
    monkeypatch.setenv("PGMOVE_HOLD_MIN_BYTES", "1")
    b = synth.make_batch(600, kind="rna004", seed=41)
    p = dict(kmer_size=5, rna=True, scaling=1, min_dur=20, max_dur=40, sample_limit=40)
    kmers = generate_kmers(5, rna=True)
    o = oracle_for(kmers, **p)
    o.run_batch(b)
    for cuts in ([0, 250, 600], [0, 100, 101, 380, 600], [0, 599, 600]):
        eng = GmoveEngine(GmoveParams(kmers=kmers, **p))
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            eng.submit(b.slice_reads(lo, hi))
        res = eng.finish()
        assert_result_equals_oracle(res, o, sample_limit=40)
        res2 = eng.finish()                                   # the merged view again, nothing new collected
        assert np.array_equal(res2.samples.view(np.uint64), res.samples.view(np.uint64))
        eng.submit(b.slice_reads(0, 50))                      # a batch behind a finish: the mixed case (falls back to the host merge)
        res3 = eng.finish()
        assert res3.n_reads == 650 and int(res3.counts.sum()) >= int(res.counts.sum())
        eng.close()


def test_collect_with_a_base_below_the_running_count():
    """k_rank_scan works the last useful tile out against the context's running count; pg_collect with a SMALLER base of the caller's
    own (here: zero, after a first batch has filled the running count) must still place every event that base allows."""
    b1 = synth.make_batch(300, kind="rna004", seed=31)
    b2 = synth.make_batch(300, kind="rna004", seed=32)
    p = dict(kmer_size=5, rna=True, scaling=1, min_dur=20, max_dur=40, sample_limit=12)
    kmers = generate_kmers(5, rna=True)
    o = oracle_for(kmers, **p)
    o.run_batch(b2)                                # the second batch on its own, base 0
    eng = GmoveEngine(GmoveParams(kmers=kmers, **p))
    eng.submit(b1); eng.sync()                     # running counts: most k-mers full
    eng.count(b2)
    eng.collect(np.zeros(len(kmers), dtype=np.uint64))
    res = eng.finish()
    for s in range(len(kmers)):                    # batch 1's events come first in every slot; batch 2's must be the oracle's
        want = o.values(s)
        got = res.slot_values(s)
        assert np.array_equal(got[got.size - want.size:].view(np.uint64), want.view(np.uint64)), s
    eng.close()


def test_device_resident_batch_and_reset():
    import torch
    b = synth.make_batch(300, kind="rna004", seed=14)
    p = dict(kmer_size=5, rna=True, scaling=1, min_dur=20, max_dur=40, sample_limit=25)
    kmers = generate_kmers(5, rna=True)
    o = oracle_for(kmers, **p)
    o.run_batch(b)
    eng = GmoveEngine(GmoveParams(kmers=kmers, **p))
    db = b.to_device(torch.device("cuda:0"))
    for _ in range(3):  # the same batch after reset must give identical results (determinism)
        eng.reset()
        eng.submit(db)
        res = eng.finish()
        assert_result_equals_oracle(res, o, sample_limit=25)
    eng.close()


def test_lazy_stats_same_output():
    b = synth.make_batch(300, kind="rna004", seed=15)
    p = dict(kmer_size=5, rna=True, scaling=1, min_dur=20, max_dur=40, sample_limit=10)
    kmers = generate_kmers(5, rna=True)
    o = oracle_for(kmers, **p)
    o.run_batch(b)
    res = run_engine([b], kmers=kmers, lazy_stats=True, **p)
    assert_result_equals_oracle(res, o, sample_limit=10)


def test_error_paths():
    from poregen_amd.engine import PgError
    b = synth.make_batch(20, kind="rna004", seed=16)
    kmers = generate_kmers(5, rna=True)
    eng = GmoveEngine(GmoveParams(kmers=kmers, kmer_size=5, rna=False))  # RNA-oriented records without --rna
    with pytest.raises(PgError) as ei:
        eng.submit(b)
        eng.sync()  # per-read errors surface at the next synchronisation point
    assert ei.value.status == -4
    eng.close()


@pytest.mark.parametrize("resident", [False, True], ids=["produced_on_the_stream", "resident_batches"])
def test_stream_ordered_count_collect_on_torch_stream(resident):
    """pg_set_stream: count -> torch op on the counts -> collect, all ordered on one torch stream, no host sync. resident: the shards
    are complete before the first step and say so (PG_BATCH_RESIDENT): the statistics stream does not wait for the caller's stream."""
    import torch
    b = synth.make_batch(300, kind="rna004", seed=17)
    p = dict(kmer_size=5, rna=True, scaling=1, min_dur=20, max_dur=40, sample_limit=12)
    kmers = generate_kmers(5, rna=True)
    o = oracle_for(kmers, **p)
    o.run_batch(b)
    dev = torch.device("cuda:0")
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        eng = GmoveEngine(GmoveParams(kmers=kmers, **p))
        eng.use_torch_stream(side)
        shards = [b.slice_reads(0, 120).to_device(dev), b.slice_reads(120, 300).to_device(dev)]
        if resident:
            torch.cuda.synchronize()
            for sh in shards:
                sh.resident = True
        cnt = torch.empty(len(kmers), dtype=torch.int64, device=dev)
        base = torch.zeros(len(kmers), dtype=torch.int64, device=dev)
        results = []
        for sh in shards:
            eng.count(sh, out=cnt)
            eng.collect(base.clone())      # device-side base, produced by torch ops on the same stream
            base = base + cnt
            results.append(eng.finish())
            eng.reset()
        eng.close()
    for s in range(len(kmers)):
        vals = np.concatenate([r.slot_values(s) for r in results])
        assert np.array_equal(vals.view(np.uint64), o.values(s).view(np.uint64))


@pytest.mark.parametrize("fence", ["no_system_fence", "system_fence"])
@pytest.mark.parametrize("mode", ["overlap", "overlap_tail"])
def test_overlap_mode_same_bits(mode, fence, monkeypatch):
    """Both kinds of event between the streams (advisor r05): the default hipEventDisableSystemFence events -- med / MAD / calibration written on
    the statistics stream and read by the gather on the other rest on the device-scope release / acquire every kernel boundary has -- and
    the plain events (PGMOVE_EVENT_SYSTEM_FENCE=1, read when the context is created).
    PG_FLAG_OVERLAP_TAIL: the same second stream, forked behind the counting kernels (next to pg_collect's small launches).
    PG_FLAG_OVERLAP: the statistics kernels of a batch on a second stream (double-buffered med/MAD), joined before
    the gather. Several batches back to back, host- and device-resident: the same bits as the oracle."""
    import torch
    if fence == "system_fence":
        monkeypatch.setenv("PGMOVE_EVENT_SYSTEM_FENCE", "1")
    b = synth.make_batch(700, kind="rna004", seed=31, indel_rate=0.02)
    p = dict(kmer_size=5, rna=True, scaling=1, min_dur=20, max_dur=40, sample_limit=60)
    kmers = generate_kmers(5, rna=True)
    o = oracle_for(kmers, **p)
    o.run_batch(b)
    bounds = [0, 90, 91, 300, 520, 700]
    for on_device in (False, True):
        eng = GmoveEngine(GmoveParams(kmers=kmers, **{mode: True}, **p))
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            part = b.slice_reads(lo, hi)
            eng.submit(part.to_device(torch.device("cuda:0")) if on_device else part)
        res = eng.finish()
        eng.close()
        assert_result_equals_oracle(res, o, check_text_slots=2, sample_limit=60)


@pytest.mark.parametrize("k", [5, 9])
def test_gather_beside_the_next_batch_same_bits(k, monkeypatch):
    """PGMOVE_GATHER_SIDE=1: the chunked gather of a batch on a third stream while the main stream already runs the next batch's chain
    (records, sample offsets, totals and chunk sums per statistics slot). Different device-resident batches back to back without a
    synchronisation in between, as bench.py's steps are: each job's result equals its own oracle run, whatever was in flight beside it."""
    import torch
    monkeypatch.setenv("PGMOVE_GATHER_SIDE", "1")
    monkeypatch.setenv("PGMOVE_DENSE_MIN", "0")
    kind = "rna004" if k == 5 else "dna_r10"
    p = dict(kmer_size=k, rna=k == 5, scaling=1, sample_limit=40 if k == 5 else 3)
    if k == 5:
        p.update(min_dur=20, max_dur=40)
    kmers = generate_kmers(k, rna=k == 5)
    dev = torch.device("cuda:0")
    hosts = [synth.make_batch(n, kind=kind, seed=77 + i) for i, n in enumerate((260, 90, 200))]
    shards = [h.to_device(dev) for h in hosts]
    eng = GmoveEngine(GmoveParams(kmers=kmers, **p))
    for last in (2, 0, 1):
        for i in (0, 1, 2, 0, 1, 2):       # a train of jobs, none of them waited for
            eng.reset(); eng.submit(shards[i])
        eng.reset(); eng.submit(shards[last])
        res = eng.finish()
        o = oracle_for(kmers, **p)
        o.run_batch(hosts[last])
        assert_result_equals_oracle(res, o, sample_limit=p["sample_limit"])
    eng.close()


def test_split_walk_same_bits():
    """PG_FLAG_DEBUG_SPLIT_WALK: the ss walk and the event filter as two launches (k_walk<false> + k_events) instead of the
    fused kernel every other test runs. Indels, pick margin, move offset, DNA and RNA orientation, skipped reads."""
    for kind, rna, extra in (("rna004", True, dict(min_dur=20, max_dur=40)), ("dna_r10", False, dict(sig_move_offset=1))):
        b = synth.make_batch(500, kind=kind, seed=41, indel_rate=0.03)
        p = dict(kmer_size=5, rna=rna, scaling=1, sample_limit=30, kmer_pick_margin=2, **extra)
        kmers = generate_kmers(5, rna=rna)
        o = oracle_for(kmers, **p)
        o.run_batch(b)
        for split in (True, False):
            eng = GmoveEngine(GmoveParams(kmers=kmers, split_walk=split, **p))
            for lo, hi in ((0, 123), (123, 500)):
                eng.submit(b.slice_reads(lo, hi))
            res = eng.finish()
            eng.close()
            assert_result_equals_oracle(res, o, check_text_slots=2, sample_limit=30)


@pytest.mark.parametrize("margin", [0, 7, 100, 101, 150])
def test_fused_walk_long_reads_and_wide_pick_margins(margin):
    """The fused walk/event kernel keeps a 512-match window of a read in LDS: reads of more than 512 ss ops go through it in
    tiles of 256 events (window re-filled per tile, reaching `margin` matches back), margins above 100 take the two-launch
    form. Long DNA reads (about 1600 ops) with indels, both forms, against the oracle."""
    b = synth.make_batch(40, read_len=20000, kind="dna_r10", seed=43 + margin, indel_rate=0.03)
    assert int(np.diff(b.op_off).max()) > 1024
    p = dict(kmer_size=6, scaling=1, sample_limit=50, kmer_pick_margin=margin, sig_move_offset=2)
    kmers = generate_kmers(6)
    o = oracle_for(kmers, **p)
    o.run_batch(b)
    for split in (False, True):
        eng = GmoveEngine(GmoveParams(kmers=kmers, split_walk=split, **p))
        eng.submit(b.slice_reads(0, 17)); eng.submit(b.slice_reads(17, 40))
        res = eng.finish()
        eng.close()
        assert_result_equals_oracle(res, o, check_text_slots=2, sample_limit=50)


def test_deferred_statistics_same_bits():
    """PG_FLAG_DEFER_STATS: pg_count leaves the statistics of every read to pg_stats (called between count and collect) or,
    when that call is missing, to pg_collect; pg_submit and several batches included. Same bits as the oracle each way."""
    import torch
    b = synth.make_batch(600, kind="rna004", seed=37, indel_rate=0.02)
    p = dict(kmer_size=5, rna=True, scaling=1, min_dur=20, max_dur=40, sample_limit=40)
    kmers = generate_kmers(5, rna=True)
    o = oracle_for(kmers, **p)
    o.run_batch(b)
    bounds = [0, 200, 201, 600]
    for mode in ("stats", "collect", "submit", "twice"):
        for on_device in (False, True):
            eng = GmoveEngine(GmoveParams(kmers=kmers, defer_stats=True, **p))
            for lo, hi in zip(bounds[:-1], bounds[1:]):
                part = b.slice_reads(lo, hi)
                part = part.to_device(torch.device("cuda:0")) if on_device else part
                if mode == "submit":
                    eng.submit(part)
                    continue
                eng.count(part)
                if mode != "collect":
                    eng.stats()
                if mode == "twice":
                    eng.stats()   # nothing left to place: a no-op
                eng.collect()
            res = eng.finish()
            eng.close()
            assert_result_equals_oracle(res, o, check_text_slots=2, sample_limit=40)
    eng = GmoveEngine(GmoveParams(kmers=kmers, defer_stats=True, **p))
    with pytest.raises(Exception):
        eng.stats()               # no counted batch
    eng.close()


def test_collect_gathered_three_ranks_on_one_gpu():
    """pg_collect_gathered: the all_gather's receive buffer (world x n_slots) goes in as it is and the library sums the
    rows below its rank on the device. Three 'ranks' (engines) on one GPU reproduce the single run / the oracle."""
    import torch
    b = synth.make_batch(450, kind="rna004", seed=23)
    p = dict(kmer_size=5, rna=True, scaling=1, min_dur=20, max_dur=40, sample_limit=25)
    kmers = generate_kmers(5, rna=True)
    o = oracle_for(kmers, **p)
    o.run_batch(b)
    dev = torch.device("cuda:0")
    bounds = [0, 140, 141, 450]
    shards = [b.slice_reads(lo, hi).to_device(dev) for lo, hi in zip(bounds[:-1], bounds[1:])]
    engs = [GmoveEngine(GmoveParams(kmers=kmers, **p)) for _ in shards]
    allc = torch.empty(len(shards) * len(kmers), dtype=torch.int64, device=dev)
    for g, (e, sh) in enumerate(zip(engs, shards)):
        e.count(sh, out=allc[g * len(kmers):(g + 1) * len(kmers)])
    for e in engs:
        e.sync()
    results = []
    for g, e in enumerate(engs):
        e.collect_gathered(allc, len(shards), g)
        results.append(e.finish())
        tot, freq = (t.cpu().numpy() for t in e.job_totals())   # every rank holds the job's totals and freq.txt column
        assert np.array_equal(tot, allc.view(3, -1).sum(0).cpu().numpy()) and np.array_equal(freq.astype(np.uint64), o.counts())
        e.close()
    for s in range(len(kmers)):
        vals = np.concatenate([r.slot_values(s) for r in results])
        assert np.array_equal(vals.view(np.uint64), o.values(s).view(np.uint64))
    with pytest.raises(Exception):
        GmoveEngine(GmoveParams(kmers=kmers, **p)).collect_gathered(allc, 3, 3)


def test_sharded_step_over_rccl_single_rank(tmp_path):
    """Both placements of the statistics: queued by pg_stats between the issue of the all_gather and the wait for it (as
    bench.py runs N > 1), and inside pg_count. dist.sharded_step on the RCCL ("nccl") backend, stream-ordered as bench.py runs it at N > 1: count ->
    all_gather_into_tensor -> pg_collect_gathered on one torch stream. A one-rank group is all a one-GPU box allows; it
    still runs the collective, the receive-buffer hand-over and the stream ordering."""
    import torch
    import torch.distributed as dist
    from poregen_amd import dist as pgdist
    b = synth.make_batch(300, kind="rna004", seed=29)
    p = dict(kmer_size=5, rna=True, scaling=1, min_dur=20, max_dur=40, sample_limit=15)
    kmers = generate_kmers(5, rna=True)
    o = oracle_for(kmers, **p)
    o.run_batch(b)
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", init_method=f"file://{tmp_path}/rdv", rank=0, world_size=1, device_id=dev)
    try:
        side = torch.cuda.Stream(device=dev)
        for defer in (True, False):
            with torch.cuda.stream(side):
                eng = GmoveEngine(GmoveParams(kmers=kmers, defer_stats=defer, **p))
                eng.use_torch_stream(side)
                total = pgdist.sharded_step(eng, b.to_device(dev), stream_ordered=True)
                freq = pgdist.merged_freq(total, 15, engine=eng)   # produced on the device by pg_collect_gathered
                res = eng.finish()
                assert freq is eng.job_totals()[1] and np.array_equal(freq.cpu().numpy().astype(np.uint64), o.counts())
                assert np.array_equal(freq.cpu().numpy(), np.minimum(total.cpu().numpy(), 15))
                total = total.clone()   # the step's totals alias the engine's buffers: keep a copy beyond close()
                eng.close()
            assert np.array_equal(np.minimum(total.cpu().numpy().astype(np.uint64), 15), o.counts())
            assert_result_equals_oracle(res, o, check_text_slots=2, sample_limit=15)
        # the single-writer end of the job over the same group: device tensors that alias the library's buffers
        eng2 = GmoveEngine(GmoveParams(kmers=kmers, **p))
        eng2.submit(b.to_device(dev))
        counts, ev_len, samples = eng2.kept_tensors()
        g = pgdist.gather_kept(counts, ev_len, samples)
        assert np.array_equal(g[0].cpu().numpy().astype(np.uint64), o.counts())
        assert np.array_equal(g[1].cpu().numpy(), np.concatenate([o.event_lens(s) for s in range(len(kmers))]).astype(np.int32))
        assert np.array_equal(g[2].cpu().numpy().view(np.uint64), np.concatenate([o.values(s) for s in range(len(kmers))]).view(np.uint64))
        # the k-mer model of the whole job on the writing rank, from the gathered device tensors (pg_model_device)
        mg = eng2.model_device(*g)
        m1 = eng2.model()
        assert mg.median_text == m1.median_text and mg.sstdev_text == m1.sstdev_text and mg.dwell_text == m1.dwell_text
        assert np.array_equal(mg.n_values, m1.n_values) and int(m1.n_values.sum()) > 1000
        eng2.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("pa", [(40.0, 180.0), (100.0, 180.0), (-50.0, 95.0)])
def test_mad_fallback_search_path(pa):
    """debug_narrow shrinks the exact MAD candidate window to one code, so the full 64-lane search runs for
    most reads; both paths must give the same bits as the oracle (also with zero-filled samples mid-range)."""
    b = synth.make_batch(300, kind="rna004", seed=18)
    p = dict(kmer_size=5, rna=True, scaling=1, min_dur=20, max_dur=40, sample_limit=15, pa_min=pa[0], pa_max=pa[1])
    kmers = generate_kmers(5, rna=True)
    o = oracle_for(kmers, **p)
    o.run_batch(b)
    for narrow in (False, True):
        res = run_engine([b], kmers=kmers, debug_narrow=narrow, **p)
        assert_result_equals_oracle(res, o, sample_limit=15)


@pytest.mark.parametrize("pa", [(0.0, 250.0), (-300.0, 1000.0), (-5000.0, 5000.0)])
def test_wide_pa_windows_use_the_wider_histograms(pa):
    """[pa_min, pa_max] wider than 1024 raw codes: 2048-bin LDS histogram, then the 65536-bin global-memory one."""
    b = synth.make_batch(120, kind="rna004", seed=19, spike_rate=0.05)
    p = dict(kmer_size=5, rna=True, scaling=1, min_dur=20, max_dur=40, sample_limit=8, pa_min=pa[0], pa_max=pa[1])
    kmers = generate_kmers(5, rna=True)
    o = oracle_for(kmers, **p)
    o.run_batch(b)
    res = run_engine([b], kmers=kmers, **p)
    assert_result_equals_oracle(res, o, sample_limit=8)


@pytest.mark.parametrize("case", [c for c in CASES if c["name"] in ("rna_k5_medmad", "rna_k5_noscale_limit7", "dna_k9", "rna_pamin100", "dna_margin3")], ids=lambda c: c["name"])
def test_device_text_equals_printf(case, monkeypatch):
    """pg_text: the dump files' bytes produced on the device == '%.8f,' ... '%.8f;' of the oracle-checked doubles (src/gmove.cpp:938-944), for
    one batch (samples still in the context's buffer) and for three (merged on the device first: PGMOVE_HOLD_MIN_BYTES=1 keeps even these
    small batches there)."""
    monkeypatch.setenv("PGMOVE_HOLD_MIN_BYTES", "1")
    b = synth.make_batch(case["n"], kind=case["kind"], seed=20251003 + len(case["name"]), **case["gen"])
    p = case["p"]
    kmers = generate_kmers(p["kmer_size"], rna=p.get("rna", False))
    for parts in ([b], [b.slice_reads(0, 40), b.slice_reads(40, 41), b.slice_reads(41, case["n"])]):
        eng = GmoveEngine(GmoveParams(kmers=kmers, **p))
        for part in parts:
            eng.submit(part)
        text = eng.text()
        res = eng.finish()
        nz = [s for s in range(len(kmers)) if res.counts[s]]
        for s in nz[:300] + nz[-50:]:
            assert text[s] == res.slot_text(s).encode(), s
        assert sum(len(t) for t in text) == sum(len(res.slot_text(s)) for s in nz) if len(nz) <= 2000 else True
        assert all(len(text[s]) == 0 for s in range(len(kmers)) if not res.counts[s])
        eng.close()
