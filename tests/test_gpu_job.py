"""pg_job_*: one job over several GPUs from one process (include/pgmove.h). A one-GPU box allows: several shards on ONE device
(the exchange through host memory) and the RCCL all-gather on a one-rank communicator. Both against the CPU oracle."""
import numpy as np
import pytest

from helpers import assert_result_equals_oracle, oracle_for
from poregen_amd import _abi, synth
from poregen_amd.engine import GmoveEngine, GmoveJob, GmoveParams, PgError, generate_kmers

pytestmark = pytest.mark.gpu


def _whitelist_case():
    rng = np.random.default_rng(6)
    full = generate_kmers(5, rna=True)
    wl = [full[i] for i in rng.permutation(len(full))[:300]]
    b = synth.make_batch(420, kind="rna004", seed=601, indel_rate=0.03)
    p = dict(kmer_size=5, rna=True, scaling=1, min_dur=20, max_dur=40, kmer_pick_margin=2, sample_limit=9)
    return wl, b, p


@pytest.mark.parametrize("devices,exchange", [([0, 0, 0], _abi.PG_JOB_EXCHANGE_AUTO), ([0], _abi.PG_JOB_EXCHANGE_RCCL), ([0, 0], _abi.PG_JOB_EXCHANGE_HOST)],
                         ids=["three_shards_on_one_device", "rccl_one_rank", "two_shards_host"])
def test_job_equals_oracle_whitelist_indels_delimiters(devices, exchange):
    """configs[4] shape at oracle size (shuffled whitelist slice, indels, pick margin 2, -d) in three uneven batches."""
    wl, b, p = _whitelist_case()
    o = oracle_for(wl, index_start=51, index_end=250, delimit=True, **p)
    o.run_batch(b)
    job = GmoveJob(GmoveParams(kmers=wl[50:250], **p), devices, exchange)
    assert job.uses_rccl == (exchange == _abi.PG_JOB_EXCHANGE_RCCL)
    for lo, hi in ((0, 150), (150, 151), (151, 420)):
        job.submit(b.slice_reads(lo, hi))
    res = job.finish()
    assert_result_equals_oracle(res, o, delimit=True, sample_limit=9)
    res2 = job.finish()                       # a repeated call returns the same merged view
    assert np.array_equal(res2.samples.view(np.uint64), res.samples.view(np.uint64))
    job.close()


def test_job_equals_single_context_and_stops_when_full():
    """All 64 3-mers fill up inside the second batch: the job reports it like a context does, the streams are identical, and
    a batch with fewer reads than shards (empty shards) is fine."""
    b = synth.make_batch(300, kind="rna004", seed=602)
    p = dict(kmer_size=3, rna=True, scaling=1, min_dur=20, max_dur=40, sample_limit=40)
    kmers = generate_kmers(3, rna=True)
    eng = GmoveEngine(GmoveParams(kmers=kmers, stop_when_full=True, **p))
    job = GmoveJob(GmoveParams(kmers=kmers, stop_when_full=True, **p), [0, 0, 0, 0])
    full_e, full_j = [], []
    for lo, hi in ((0, 2), (2, 120), (120, 300)):
        part = b.slice_reads(lo, hi)
        eng.submit(part); job.submit(part)
        full_e.append(eng.all_slots_full()); full_j.append(job.all_slots_full())
    assert full_e == full_j and full_j[-1]
    re, rj = eng.finish(), job.finish()
    for name in ("counts", "ev_off", "ev_len", "ev_read", "samp_off", "read_skipped"):
        assert np.array_equal(getattr(re, name), getattr(rj, name)), name
    assert np.array_equal(re.samples.view(np.uint64), rj.samples.view(np.uint64))
    me, mj = eng.model(), job.model()
    assert me.median_text == mj.median_text and me.sstdev_text == mj.sstdev_text and me.dwell_text == mj.dwell_text
    eng.close(); job.close()


@pytest.mark.parametrize("host_merge", [False, True, "peer"], ids=["device_concat", "host_merge", "peer_copies"])
def test_job_output_side_on_the_first_device(host_merge, monkeypatch):
    """pg_job_finish_deferred / _fetch_samples / _text / _model: the shards' kept samples concatenated on the job's first device (here
    four shards of one GPU, three uneven batches; PGMOVE_HOLD_MIN_BYTES=1 keeps even these small batches on the device inside every
    shard) against the oracle, against pg_job_finish, and against printf. PGMOVE_JOB_HOST_MERGE=1: the fallback through the host."""
    monkeypatch.setenv("PGMOVE_HOLD_MIN_BYTES", "1")
    if host_merge == "peer":  # the branch a multi-GPU job takes for shards on other devices: staging buffer on the first device + hipMemcpyPeerAsync
        monkeypatch.setenv("PGMOVE_JOB_FORCE_PEER", "1")
    elif host_merge:
        monkeypatch.setenv("PGMOVE_JOB_HOST_MERGE", "1")
    wl, b, p = _whitelist_case()
    p = dict(p, sample_limit=40)
    o = oracle_for(wl, index_start=51, index_end=250, **p)
    o.run_batch(b)
    job = GmoveJob(GmoveParams(kmers=wl[50:250], **p), [0, 0, 0, 0])
    for lo, hi in ((0, 150), (150, 151), (151, 420)):
        job.submit(b.slice_reads(lo, hi))
    res_d = job.finish_deferred(piece=1000)       # samples fetched in ranges from the device
    assert_result_equals_oracle(res_d, o, sample_limit=40)
    text = job.text()
    for s in range(200):
        assert text[s] == res_d.slot_text(s).encode(), s
    m = job.model()
    res = job.finish()                            # the same merged view, samples downloaded as a whole
    assert np.array_equal(res.samples.view(np.uint64), res_d.samples.view(np.uint64))
    for name in ("counts", "ev_off", "ev_len", "ev_read", "samp_off", "read_skipped"):
        assert np.array_equal(getattr(res, name), getattr(res_d, name)), name
    job.close()
    eng = GmoveEngine(GmoveParams(kmers=wl[50:250], **p))
    eng.submit(b)
    me = eng.model()
    assert me.median_text == m.median_text and me.sstdev_text == m.sstdev_text and me.dwell_text == m.dwell_text
    eng.close()


def test_rank_level_early_out_skips_statistics():
    """Nothing behind the completing read is touched by the reference (src/gmove.cpp:733-735). Four shards on device 0 (host exchange):
    all 64 3-mers are complete inside the first shards of batch 1, so the later shards -- whose base already fills every k-mer -- queue
    no statistics kernel at all, and neither does ANY shard for a batch submitted after the job was complete. Same output as one context."""
    b = synth.make_batch(600, kind="rna004", seed=605)
    p = dict(kmer_size=3, rna=True, scaling=1, min_dur=20, max_dur=40, sample_limit=40)
    kmers = generate_kmers(3, rna=True)
    eng = GmoveEngine(GmoveParams(kmers=kmers, **p))
    job = GmoveJob(GmoveParams(kmers=kmers, profile=True, **p), [0, 0, 0, 0])
    first, second = b.slice_reads(0, 400), b.slice_reads(400, 600)
    eng.submit(first); job.submit(first); job.sync()
    assert job.all_slots_full()
    ks = [job.kernel_stats(g) for g in range(4)]
    launches = [k.get("k_read_stats", (0, 0.0))[0] for k in ks]
    assert launches[0] == 1 and launches[3] == 0 and launches[2] == 0, launches   # 100 reads fill 64 k-mers x 40 events
    eng.submit(second); job.submit(second); job.sync()
    assert [job.kernel_stats(g).get("k_read_stats", (0, 0.0))[0] for g in range(4)] == launches  # complete before the batch: no shard computes statistics
    re, rj = eng.finish(), job.finish()
    for name in ("counts", "ev_off", "ev_len", "ev_read", "samp_off", "read_skipped"):
        assert np.array_equal(getattr(re, name), getattr(rj, name)), name
    assert np.array_equal(re.samples.view(np.uint64), rj.samples.view(np.uint64))
    eng.close(); job.close()


def test_rank_level_early_out_decided_on_the_device(monkeypatch):
    """The same rule where the table of accepted events never comes to the host (the RCCL exchange; here the host exchange is told to take
    the device's rule, PGMOVE_JOB_DEVICE_RULE=1, because an RCCL communicator wants distinct devices): a shard with rows below it queues
    its statistics behind the table, and a flag computed there cancels them INSIDE the batch that completes the job -- the launches happen,
    every read's record says "skip", no sample is read (gmove.cpp:733-735). Same output as one context."""
    monkeypatch.setenv("PGMOVE_JOB_DEVICE_RULE", "1")
    b = synth.make_batch(600, kind="rna004", seed=605)
    p = dict(kmer_size=3, rna=True, scaling=1, min_dur=20, max_dur=40, sample_limit=40)
    kmers = generate_kmers(3, rna=True)
    eng = GmoveEngine(GmoveParams(kmers=kmers, **p))
    job = GmoveJob(GmoveParams(kmers=kmers, profile=True, **p), [0, 0, 0, 0])
    first, second = b.slice_reads(0, 400), b.slice_reads(400, 600)
    eng.submit(first); job.submit(first); job.sync()
    assert job.all_slots_full()
    ks = [job.kernel_stats(g) for g in range(4)]
    cancelled = [k.get("stats_cancelled_on_device", (0, 0.0))[0] for k in ks]
    assert cancelled[0] == 0 and cancelled[2] == 1 and cancelled[3] == 1, cancelled   # 100 reads fill 64 k-mers x 40 events: shards 2, 3 lie behind them
    assert all(k["k_read_stats"][0] == 1 for k in ks)   # the launches are there (queued before the table is known); shards 2, 3 read records only
    eng.submit(second); job.submit(second); job.sync()  # complete before the batch: the host knows, nothing is queued at all
    assert [job.kernel_stats(g).get("k_read_stats", (0, 0.0))[0] for g in range(4)] == [k["k_read_stats"][0] for k in ks]
    re, rj = eng.finish(), job.finish()
    for name in ("counts", "ev_off", "ev_len", "ev_read", "samp_off", "read_skipped"):
        assert np.array_equal(getattr(re, name), getattr(rj, name)), name
    assert np.array_equal(re.samples.view(np.uint64), rj.samples.view(np.uint64))
    eng.close(); job.close()


def test_job_rccl_large_limit_many_batches():
    """The RCCL path (one-rank communicator) with the running total carried in row 0 of the receive buffer over five batches."""
    b = synth.make_batch_fast(2500, kind="rna004", seed=603)
    p = dict(kmer_size=5, rna=True, scaling=1, min_dur=20, max_dur=40, sample_limit=100)
    kmers = generate_kmers(5, rna=True)
    o = oracle_for(kmers, **p)
    o.run_batch(b)
    job = GmoveJob(GmoveParams(kmers=kmers, **p), [0], _abi.PG_JOB_EXCHANGE_RCCL)
    for lo in range(0, 2500, 500):
        job.submit(b.slice_reads(lo, lo + 500))
    assert_result_equals_oracle(job.finish(), o, sample_limit=100)
    job.close()


def test_job_errors():
    kmers = generate_kmers(3)
    with pytest.raises(PgError) as ei:
        GmoveJob(GmoveParams(kmers=kmers, kmer_size=3), [0, 0], _abi.PG_JOB_EXCHANGE_RCCL)
    assert ei.value.status == _abi.PG_ERR_INVALID_ARG and "distinct" in ei.value.text
    with pytest.raises(PgError) as ei:
        GmoveJob(GmoveParams(kmers=kmers, kmer_size=3), [0, 99])
    assert ei.value.status == _abi.PG_ERR_NO_DEVICE
    # an RNA-oriented record without --rna in the LAST shard fails the job, with the shard named
    b = synth.make_batch(30, kind="dna_r10", seed=604)
    b.target_start[27], b.target_end[27] = b.target_end[27], b.target_start[27]
    job = GmoveJob(GmoveParams(kmers=kmers, kmer_size=3, scaling=1), [0, 0, 0])
    with pytest.raises(PgError) as ei:
        job.submit(b); job.finish()
    assert ei.value.status == _abi.PG_ERR_RNA_FLAG and "shard 2" in ei.value.text
    job.close()


@pytest.mark.parametrize("devices,exchange,resident", [([0, 0, 0], _abi.PG_JOB_EXCHANGE_HOST, True), ([0], _abi.PG_JOB_EXCHANGE_RCCL, True), ([0, 0], _abi.PG_JOB_EXCHANGE_HOST, False)],
                         ids=["three_device_shards_one_gpu", "rccl_one_rank_device_shard", "two_host_shards"])
def test_job_submit_shards_device_resident(devices, exchange, resident):
    """pg_job_submit_shards: every rank's shard handed over as a batch of its own, already resident on its device (nothing is cut, staged
    or copied by the host: the device-resident N-GPU step of the C++ host) or on the host. Same streams as the oracle over the
    concatenation, in two batches, with an empty shard in the second."""
    import torch
    wl, b, p = _whitelist_case()
    o = oracle_for(wl, index_start=51, index_end=250, delimit=True, **p)
    o.run_batch(b)
    job = GmoveJob(GmoveParams(kmers=wl[50:250], **p), devices, exchange)
    n = len(devices)
    dev = torch.device("cuda", 0)
    keep = []
    for lo, hi in ((0, 200), (200, 420)):
        cuts = [lo + (hi - lo) * g // n for g in range(n + 1)]
        if lo == 200 and n > 1:
            cuts[1] = cuts[0]               # rank 0's shard of the second batch is empty
        shards = [b.slice_reads(cuts[g], cuts[g + 1]) for g in range(n)]
        if resident:
            shards = [s.to_device(dev) for s in shards]
        keep.append(shards)
        job.submit_shards(shards)
    res = job.finish()
    assert_result_equals_oracle(res, o, delimit=True, sample_limit=9)
    with pytest.raises(PgError):
        job.submit_shards(keep[0][:-1] if n > 1 else keep[0] + keep[0])   # one batch per device, no more, no fewer
    if resident:
        # a "device" shard whose samples live in pinned HOST memory: refused by name (hipPointerGetAttributes), not a fault in the first kernel
        bad = keep[0][0]
        host_sig = torch.zeros(bad.sig.numel(), dtype=bad.sig.dtype).pin_memory()
        lie = type(bad)(**{**bad.__dict__, "sig": host_sig})
        with pytest.raises(PgError, match="device"):
            job.submit_shards([lie] + keep[0][1:])
    job.close()


def test_job_reset_starts_a_new_job():
    b = synth.make_batch(200, kind="rna004", seed=603)
    p = dict(kmer_size=4, rna=True, scaling=1, min_dur=20, max_dur=40, sample_limit=25)
    kmers = generate_kmers(4, rna=True)
    o = oracle_for(kmers, **p); o.run_batch(b)
    job = GmoveJob(GmoveParams(kmers=kmers, **p), [0, 0])
    job.submit(b.slice_reads(0, 90)); job.submit(b.slice_reads(90, 200))
    first = job.finish()
    assert_result_equals_oracle(first, o, sample_limit=25)
    job.reset()
    job.submit(b)                      # the same reads as one batch of a NEW job: nothing of the first job's counts is left
    again = job.finish()
    assert_result_equals_oracle(again, o, sample_limit=25)
    job.close()
