"""Shared checks: run a host Batch through the CPU oracle and compare an engine Result with it bit for bit."""
import numpy as np

import orc
from poregen_amd.engine import GmoveParams, generate_kmers


def oracle_for(kmers_full, index_start=1, index_end=None, **p):
    """p uses GmoveParams names."""
    return orc.Oracle(
        kmers_full, index_start=index_start, index_end=index_end,
        kmer_size=p.get("kmer_size", 9), sig_move_offset=p.get("sig_move_offset", 0), scaling=p.get("scaling", 0),
        signal_print_margin=p.get("margin", 0), sample_limit=p.get("sample_limit", 100), max_dur=p.get("max_dur", 70),
        min_dur=p.get("min_dur", 5), pa_min=p.get("pa_min", 40.0), pa_max=p.get("pa_max", 180.0),
        kmer_pick_margin=p.get("kmer_pick_margin", 2), flag_rna=int(p.get("rna", False)), delimit_files=int(p.get("delimit", False)))


def assert_result_equals_oracle(res, o, check_text_slots=8, delimit=False, sample_limit=None):
    """Integer indexing bit-exact; sample doubles bit-exact (stronger than the 1e-5 of BASELINE.json)."""
    oc = o.counts()
    assert res.counts.dtype == np.uint64 and np.array_equal(res.counts, oc), (res.counts[:16], oc[:16])
    checked = 0
    for s in range(o.n_slots):
        a, b = int(res.ev_off[s]), int(res.ev_off[s + 1])
        assert b - a == oc[s]
        if oc[s] == 0:
            continue
        assert np.array_equal(res.ev_len[a:b], o.event_lens(s)), f"slot {s} event lengths"
        ov = o.values(s)
        gv = res.slot_values(s)
        assert gv.size == ov.size
        assert np.array_equal(gv.view(np.uint64), ov.view(np.uint64)), f"slot {s} ({o.kmer(s)}): sample bits differ"
        if checked < check_text_slots:
            assert res.slot_text(s, delimit=delimit, sample_limit=sample_limit) == o.text(s), f"slot {s} text"
            checked += 1
    if delimit:  # empty slots still receive ':' per processed read
        for s in range(min(o.n_slots, 4)):
            assert res.slot_text(s, delimit=True, sample_limit=sample_limit) == o.text(s)
