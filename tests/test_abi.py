"""The C-ABI library loads and exports every symbol include/pgmove.h declares; host-only entry points behave;
without a GPU the product fails loudly instead of falling back to a CPU path."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from poregen_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_match_header():
    hdr = open(os.path.join(ROOT, "include", "pgmove.h")).read()
    declared = set(re.findall(r"\b(pg_[a-z_]+)\s*\(", hdr))
    assert declared == set(_abi.EXPORTS), declared ^ set(_abi.EXPORTS)
    lib = _abi.load()
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert b"gfx950" in lib.pg_version()


def test_struct_sizes_and_defaults():
    lib = _abi.load()
    p = _abi.PgParams()
    lib.pg_default_params(C.byref(p))
    assert p.struct_size == C.sizeof(_abi.PgParams)
    # init_opt defaults (src/poregen.cpp:209-237, src/poregen.h:30-43); scaling effectively 0 (gmove.cpp:229)
    assert (p.kmer_size, p.sample_limit, p.max_dur, p.min_dur, p.kmer_pick_margin, p.scaling) == (9, 100, 70, 5, 2, 0)
    assert (p.pa_min, p.pa_max) == (40.0, 180.0)


def test_slot_tables():
    lib = _abi.load()
    kmers = [b"ACGTA", b"AAAAA", b"ACGUA", b"ACNNA", b"UUUTT"]
    arr = (C.c_char_p * len(kmers))(*kmers)
    tt = np.empty(4 ** 5, np.int32); tu = np.empty(4 ** 5, np.int32)
    assert lib.pg_build_slot_tables(5, arr, len(kmers), tt.ctypes.data, tu.ctypes.data) == 0
    code = lambda s: int("".join(str("ACGT".index(c.replace("U", "T"))) for c in s), 4)
    assert tt[code("ACGTA")] == 0 and tu[code("ACGTA")] == 2   # T-spelled vs U-spelled windows are different k-mers
    assert tt[code("AAAAA")] == 1 and tu[code("AAAAA")] == 1   # no T/U: the same string in both alphabets
    assert (tt >= 0).sum() == 2 and (tu >= 0).sum() == 2        # N-containing and mixed T/U k-mers can never match
    dup = (C.c_char_p * 2)(b"ACGTA", b"ACGTA")
    assert lib.pg_build_slot_tables(5, dup, 2, tt.ctypes.data, tu.ctypes.data) == _abi.PG_ERR_INVALID_ARG
    assert b"duplicate" in lib.pg_last_error(None)


def _has_gpu():
    import torch
    return torch.cuda.is_available()


@pytest.mark.skipif(_has_gpu(), reason="only meaningful on a box without a GPU")
def test_no_device_fails_loudly():
    from poregen_amd.engine import GmoveEngine, GmoveParams, PgError, generate_kmers
    with pytest.raises(PgError) as ei:
        GmoveEngine(GmoveParams(kmers=generate_kmers(3), kmer_size=3))
    assert ei.value.status == _abi.PG_ERR_NO_DEVICE and "no CPU fallback" in ei.value.text


def test_create_validates_arguments():
    lib = _abi.load()
    p = _abi.PgParams(); lib.pg_default_params(C.byref(p))
    h = C.c_void_p()
    assert lib.pg_create(C.byref(p), C.byref(h)) == _abi.PG_ERR_INVALID_ARG   # no tables / n_slots
    p.kmer_size = 20
    assert lib.pg_create(C.byref(p), C.byref(h)) == _abi.PG_ERR_INVALID_ARG
    assert b"kmer_size" in lib.pg_last_error(None)
