"""Host-side logic of the Python layer (no GPU): k-mer list, slice reconciliation, generator, shard bounds,
dump text assembly."""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

import numpy as np

import orc
from poregen_amd import dist, synth
from poregen_amd.engine import Result, generate_kmers, reconcile_slice


def test_generate_kmers_matches_oracle():
    for k, rna in ((1, False), (3, True), (5, False)):
        L = orc.lib()
        L.orc_generate_kmers.restype = C.POINTER(C.c_char_p); L.orc_generate_kmers.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_size_t)]
        n = C.c_size_t()
        arr = L.orc_generate_kmers(k, int(rna), C.byref(n))
        ref = [arr[i].decode() for i in range(n.value)]
        assert generate_kmers(k, rna) == ref and len(ref) == 4 ** k


def test_reconcile_slice():
    assert reconcile_slice(1024, 1, 5000, 5000) == (1, 1024)      # --file_limit 5000 at k=5 (README)
    assert reconcile_slice(262144, 1, 500, 500) == (1, 500)       # defaults at k=9
    assert reconcile_slice(300, 51, 250, 200) == (51, 250)        # config 4 slice
    assert reconcile_slice(53, 1, 500, 500) == (1, 53)            # kmer_file with 53 lines, defaults


def test_generators_are_deterministic_and_well_formed():
    a = synth.make_batch(40, kind="rna004", seed=3, indel_rate=0.02)
    b = synth.make_batch(40, kind="rna004", seed=3, indel_rate=0.02)
    for name in ("sig", "op_n", "op_t", "seq", "sig_off", "op_off", "seq_off"):
        assert np.array_equal(getattr(a, name), getattr(b, name))
    for r in range(a.n_reads):
        ops_n = a.op_n[int(a.op_off[r]):int(a.op_off[r + 1])]; ops_t = a.op_t[int(a.op_off[r]):int(a.op_off[r + 1])]
        assert int(ops_n[ops_t != 2].sum()) == int(a.sig_off[r + 1] - a.sig_off[r])          # matches + insertions cover the signal
        assert int((ops_t == 0).sum() + ops_n[ops_t == 2].sum()) == int(a.seq_off[r + 1] - a.seq_off[r])  # matches + deleted bases = sequence
        assert a.target_start[r] > a.target_end[r] == 0
    f = synth.make_batch_fast(300, kind="dna_r10", seed=4)
    assert f.n_samples == 300 * 4000 and np.all(f.target_start == 0)
    assert int(f.op_n.sum()) == f.n_samples


def test_shard_bounds_cover_in_order():
    for n, w in ((10, 3), (400000, 8), (7, 8)):
        b = [dist.shard_bounds(n, w, r) for r in range(w)]
        assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


def test_slot_text_delimiters():
    # two slots, three reads; slot 0 closes (sample_limit 2) during read 1, slot 1 never fills; read 2 is skipped
    res = Result(counts=np.array([2, 1], np.uint64), ev_off=np.array([0, 2, 3], np.uint64), ev_len=np.array([1, 2, 1], np.uint32),
                 ev_read=np.array([0, 1, 1], np.uint32), samp_off=np.array([0, 1, 3, 4], np.uint64),
                 samples=np.array([1.0, 2.5, -0.125, 100.123456789]), read_skipped=np.array([0, 0, 1], np.uint8), n_reads=3)
    assert res.slot_text(0) == "1.00000000;2.50000000,-0.12500000;"
    assert res.slot_text(0, delimit=True, sample_limit=2) == "1.00000000;:2.50000000,-0.12500000;"
    assert res.slot_text(1, delimit=True, sample_limit=2) == ":100.12345679;:"


def test_big_host_vectors_ask_for_huge_pages():
    """pg_hostmem.h: a SampleVec of 4 MB or more is a mapping of its own with MADV_HUGEPAGE (the "hg" VmFlag in /proc/self/smaps);
    small ones come from the heap. (A typo once compiled the madvise call out: results are identical, only this shows it.)"""
    import ctypes as C
    import os
    import pytest
    if not os.path.exists("/sys/kernel/mm/transparent_hugepage/enabled"):
        pytest.skip("kernel without transparent huge pages: madvise(MADV_HUGEPAGE) fails there, which the code ignores by design")
    h = C.CDLL((os.environ.get("PG_HOSTTEST_SO") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "poregen_amd", "_pg_hosttest.so")))
    h.pgt_samplevec_hugepage.argtypes = [C.c_size_t, C.POINTER(C.c_long)]
    kb = C.c_long(0)
    big = h.pgt_samplevec_hugepage(1 << 20, C.byref(kb))   # 8 MB
    if big < 0:
        import pytest
        pytest.skip("/proc/self/smaps not readable")
    assert big == 1
    assert h.pgt_samplevec_hugepage(1000, C.byref(kb)) == 0


def test_ragged_generator_is_consistent_and_the_oracle_accepts_it():
    """synth.make_ragged_fast (bench.py's ragged_mode): per read the ops add up to the read's samples, the sequence has a base per op, both
    orientations run through the oracle without an error."""
    import numpy as np
    import orc
    from helpers import oracle_for
    from poregen_amd import synth
    from poregen_amd.engine import generate_kmers
    L = synth.ragged_lengths(600_000, seed=9, huge=150_000, huge_frac=0.02)
    assert int(L.sum()) == 600_000 and L.min() >= 2000
    for kind, rna in (("dna_r10", False), ("rna004", True)):
        b = synth.make_ragged_fast(L, kind=kind, seed=10)
        assert b.n_reads == L.size and np.array_equal(np.diff(b.sig_off.astype(np.int64)), L)
        for r in range(b.n_reads):
            a, e = int(b.op_off[r]), int(b.op_off[r + 1])
            assert int(b.op_n[a:e].sum()) == int(L[r]) and int(b.seq_off[r + 1] - b.seq_off[r]) == e - a
        kmers = generate_kmers(5, rna=rna)
        o = oracle_for(kmers, kmer_size=5, scaling=1, sample_limit=30, rna=rna)
        assert set(o.run_batch(b)) <= {orc.ORC_OK, orc.ORC_STOPPED}
        assert int(o.counts().sum()) > 1000


def test_where_a_ranks_statistics_go_relative_to_the_count_exchange():
    """pg_job_rule.h (pg_job.hip): in front of the wait for the exchanged table unless completion of every k-mer below the rank is
    plausible in this batch. Hand-made tables for BASELINE configs[2] (8 ranks x 50 000 reads, 6.5 M ops per rank, 1024 k-mers) at limit
    100 / 5000 and configs[3] (k = 9, 15.9 M ops per rank, 262 144 k-mers, limit 1000). PROVISIONAL: unmeasured on more than one GPU."""
    import ctypes as C
    h = C.CDLL(os.environ.get("PG_HOSTTEST_SO") or os.path.join(ROOT, "poregen_amd", "_pg_hosttest.so"))
    h.pgt_job_stats_rule.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_uint64, C.c_char_p]

    def place(rank, ops, reads, n_slots, limit, have_batch=False, full_prev=0, mode=b""):
        o = np.asarray(ops, np.uint64); r = np.asarray(reads, np.uint64)
        return h.pgt_job_stats_rule(rank, o.ctypes.data, r.ctypes.data, n_slots, limit, int(have_batch), full_prev, mode)
    FRONT, BEHIND = 0, 1
    c2_ops, c2_reads = [6_500_000] * 8, [50_000] * 8
    # configs[2], limit 100 and 5000: 6 348 ops per k-mer and rank -- one rank below is enough to make completion plausible at both limits
    assert [place(g, c2_ops, c2_reads, 1024, 100) for g in range(8)] == [FRONT] + [BEHIND] * 7
    assert [place(g, c2_ops, c2_reads, 1024, 5000) for g in range(8)] == [FRONT] + [BEHIND] * 7
    assert [place(g, c2_ops, c2_reads, 1024, 20000) for g in range(8)] == [FRONT] * 4 + [BEHIND] * 4     # 20 000 / 6 348 = 3.15: from rank 4 on
    # configs[3]: 60.7 ops per k-mer and rank against a limit of 1000: no rank of 8 (it would take 17)
    k9_ops, k9_reads = [15_900_000] * 8, [50_000] * 8
    assert [place(g, k9_ops, k9_reads, 262144, 1000) for g in range(8)] == [FRONT] * 8
    # ... unless the earlier batches left only a few k-mers open: 97 % complete -> 30 per open k-mer are enough
    assert place(1, k9_ops, k9_reads, 262144, 1000, have_batch=True, full_prev=int(262144 * 0.97)) == BEHIND
    assert place(1, k9_ops, k9_reads, 262144, 1000, have_batch=True, full_prev=int(262144 * 0.90)) == FRONT
    # an op count the host was not told (0 with reads in the shard) counts as plausible; an empty shard below does not
    assert place(2, [15_900_000, 0], [50_000, 50_000], 262144, 1000) == BEHIND
    assert place(2, [15_900_000, 0], [50_000, 0], 262144, 1000) == FRONT
    # rank 0 and sample_limit 0 (no k-mer ever completes) never wait; the overrides
    assert place(0, c2_ops, c2_reads, 1024, 100, mode=b"behind") == FRONT and place(3, c2_ops, c2_reads, 1024, 0, mode=b"behind") == FRONT
    assert place(3, c2_ops, c2_reads, 1024, 100, mode=b"front") == FRONT and place(3, k9_ops, k9_reads, 262144, 1000, mode=b"behind") == BEHIND
    assert place(3, c2_ops, c2_reads, 1024, 100, mode=b"auto") == BEHIND
