"""Median / MAD selection pinned by the REFERENCE's own code: tests/golden/ksmall_vectors.json holds, for 216 seeded reads, what
/root/reference/src/ksort.h:233-259 (ks_ksmall_double, compiled where it lies: oracle/ref_selection.c) returns through the
calc_median / calc_madf recipe of src/gmove.cpp:142-184, 751-771. Checked here: the oracle's restatement, the product's shared
host/device selection arithmetic (pg_select.h, host build) and -- on the GPU -- k_read_stats itself."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import ksmall_vectors as kv
import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_selection.so")


@pytest.fixture(scope="module")
def fixture(golden_dir):
    with open(os.path.join(golden_dir, "ksmall_vectors.json")) as f:
        doc = json.load(f)
    assert len(doc["vectors"]) == len(kv.specs()) == 216
    return doc["vectors"]


def raw_of(e):
    raw = kv.make_raw(e)
    assert kv.crc(raw) == e["crc"], f"{e['id']}: tests/ksmall_vectors.py no longer produces the vector the fixture was computed on"
    if "raw" in e:
        assert raw.tolist() == e["raw"]
    return raw


def test_fixture_vectors_cover_the_cases_the_verdict_names(fixture):
    ns = {e["n"] for e in fixture}
    assert {1, 2, 4000, 4001, 100000, 100001} <= ns
    kinds = {e["kind"] for e in fixture}
    assert {"ties", "allzero", "halfzero_eq", "wide", "huge"} <= kinds
    # clamped and unclamped MADs both occur, and zero medians (the zero-filled class wins) too
    assert any(e["mad"] != e["madf"] for e in fixture) and any(e["mad"] == e["madf"] for e in fixture)
    assert any(e["med"] == kv.bits(0.0) for e in fixture)


def test_oracle_selection_equals_the_reference_quickselect(fixture):
    """orc_median / orc_madf (oracle/gmove_oracle.c:170-187) on the reference-pinned vectors, bit for bit."""
    L = orc.lib()
    for e in fixture:
        x = kv.pa_zero_filled(raw_of(e), e)
        med = L.orc_median(x.ctypes.data, x.size)
        madf = L.orc_madf(x.ctypes.data, x.size, med)
        assert kv.bits(med) == e["med"], e["id"]
        assert kv.bits(madf) == e["madf"], e["id"]
        assert kv.bits(madf if madf > 1.0 else 1.0) == e["mad"], e["id"]


def test_product_selection_arithmetic_equals_the_reference_quickselect(fixture):
    """pg_select.h compiled for the host (the code k_read_stats' selection runs) on the same vectors."""
    h = C.CDLL(os.environ.get("PG_HOSTTEST_SO") or os.path.join(ROOT, "poregen_amd", "_pg_hosttest.so"))
    h.pgt_medmad.argtypes = [C.c_void_p, C.c_uint64] + [C.c_double] * 5 + [C.POINTER(C.c_double)] * 3
    for e in fixture:
        raw = raw_of(e)
        dig, off, rg = kv.CALS[e["cal"]]
        med, mad, mr = C.c_double(), C.c_double(), C.c_double()
        rc = h.pgt_medmad(raw.ctypes.data, raw.size, dig, off, rg, e["pa_min"], e["pa_max"], C.byref(med), C.byref(mad), C.byref(mr))
        assert rc == 0, e["id"]
        assert kv.bits(med.value) == e["med"], e["id"]
        if e["n"] > 1:
            assert kv.bits(mr.value * 1.4826) == e["madf"], e["id"]
        assert kv.bits(mad.value) == e["mad"], e["id"]


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref is built in the build container only (needs /root/reference)")
def test_fixture_is_what_the_compiled_reference_code_returns_today(fixture):
    """The committed expected values against oracle/_ref/libref_selection.so run live, plus fresh random vectors through oracle and reference."""
    lib = C.CDLL(REF_SO)
    lib.ref_read_medmad.argtypes = [C.c_void_p, C.c_size_t] + [C.c_double] * 5 + [C.POINTER(C.c_double)]
    lib.ref_read_medmad.restype = None
    lib.ref_median.argtypes = [C.c_void_p, C.c_size_t]; lib.ref_median.restype = C.c_double
    lib.ref_madf.argtypes = [C.c_void_p, C.c_size_t, C.c_double]; lib.ref_madf.restype = C.c_double
    out = (C.c_double * 3)()
    for e in fixture:
        raw = raw_of(e)
        dig, off, rg = kv.CALS[e["cal"]]
        lib.ref_read_medmad(raw.ctypes.data, raw.size, dig, off, rg, e["pa_min"], e["pa_max"], out)
        assert (kv.bits(out[0]), kv.bits(out[1]), kv.bits(out[2])) == (e["med"], e["madf"], e["mad"]), e["id"]
    L = orc.lib()
    rng = np.random.default_rng(6)
    for trial in range(300):
        n = int(rng.integers(1, 3000))
        x = np.round(rng.normal(90, rng.choice([0.0, 0.5, 10.0]), n), int(rng.integers(0, 3)))
        x[rng.random(n) < rng.choice([0.0, 0.5, 0.9])] = 0.0
        m = lib.ref_median(x.ctypes.data, n)
        assert kv.bits(m) == kv.bits(L.orc_median(x.ctypes.data, n))
        assert kv.bits(lib.ref_madf(x.ctypes.data, n, m)) == kv.bits(L.orc_madf(x.ctypes.data, n, m))


# ------------------------------------------------------------------------------------------------------------------------------
def _batch_of(vectors):
    """Every vector of >= 16 samples as one DNA read with a minimal alignment (12 bases, 12 one-sample matches): the statistics see the
    whole signal whatever the events do (gmove.cpp:751-771 run before the walk)."""
    from poregen_amd.engine import Batch
    raws = [raw_of(e) for e in vectors]
    n = len(raws)
    sig_off = np.zeros(n + 1, np.uint64)
    sig = np.concatenate(raws)   # back to back: odd lengths put the following reads at every alignment inside a 16-byte vector
    sig_off[1:] = np.cumsum([r.size for r in raws])
    cal = np.array([kv.CALS[e["cal"]] for e in vectors], np.float64)
    seq = np.tile(np.frombuffer(b"ACGTTGCAAGCT", np.uint8), n)
    return Batch(n_reads=n, sig=sig, sig_off=sig_off, digitisation=np.ascontiguousarray(cal[:, 0]), offset=np.ascontiguousarray(cal[:, 1]),
                 range=np.ascontiguousarray(cal[:, 2]), query_start=np.zeros(n, np.int32), target_start=np.zeros(n, np.int32),
                 target_end=np.full(n, 12, np.int32), seq=seq, seq_off=(np.arange(n + 1, dtype=np.uint64) * np.uint64(12)),
                 op_n=np.ones(12 * n, np.uint32), op_t=np.zeros(12 * n, np.uint8), op_off=(np.arange(n + 1, dtype=np.uint64) * np.uint64(12))).validate_host()


@pytest.mark.gpu
@pytest.mark.parametrize("pa", [(40.0, 180.0), (-1e9, 1e9)], ids=["pa40_180", "pa_all"])
@pytest.mark.parametrize("mode", ["default", "one_stream", "no_long_split"])
def test_k_read_stats_equals_the_reference_quickselect(fixture, pa, mode, monkeypatch):
    """The device statistics (k_read_stats, its wide / huge launches and the long-read split) on the reference-pinned vectors:
    pg_last_batch_device's d_med / d_mad bit for bit against ks_ksmall_double's answers."""
    import torch
    from poregen_amd.engine import GmoveEngine, GmoveParams, generate_kmers
    if mode == "no_long_split":
        monkeypatch.setenv("PGMOVE_NO_LONG_SPLIT", "1")
    vec = [e for e in fixture if e["n"] >= 16 and (e["pa_min"], e["pa_max"]) == pa]
    assert len(vec) >= 15
    b = _batch_of(vec)
    kmers = generate_kmers(5)
    p = GmoveParams(kmers=kmers, kmer_size=5, scaling=1, sample_limit=100, min_dur=1, max_dur=70, pa_min=pa[0], pa_max=pa[1], overlap=(False if mode == "one_stream" else None))
    eng = GmoveEngine(p)
    try:
        eng.submit(b)
        eng.sync()
        v = eng.device_view()

        class _A:
            def __init__(self, ptr, n): self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i8", "data": (int(ptr), False), "version": 2}
        med = torch.as_tensor(_A(v.d_med, len(vec)), device="cuda").cpu().numpy().view(np.uint64)
        mad = torch.as_tensor(_A(v.d_mad, len(vec)), device="cuda").cpu().numpy().view(np.uint64)
        for i, e in enumerate(vec):
            assert "%016x" % med[i] == e["med"], (e["id"], mode)
            assert "%016x" % mad[i] == e["mad"], (e["id"], mode)
    finally:
        eng.close()
