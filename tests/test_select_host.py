"""The shared host/device selection arithmetic (poregen_amd/csrc/pg_select.h, compiled for the host) against
the oracle's order statistics on the doubles the reference would build (src/gmove.cpp:754-771, 142-184)."""
import ctypes
import os

import numpy as np
import pytest

import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def shim():
    h = ctypes.CDLL((os.environ.get("PG_HOSTTEST_SO") or os.path.join(ROOT, "poregen_amd", "_pg_hosttest.so")))
    h.pgt_medmad.argtypes = [ctypes.c_void_p, ctypes.c_uint64] + [ctypes.c_double] * 5 + [ctypes.POINTER(ctypes.c_double)] * 3
    h.pgt_medmad_sym.argtypes = h.pgt_medmad.argtypes
    h.pgt_plan.argtypes = [ctypes.c_double] * 5 + [ctypes.c_void_p]
    return h


def ref_medmad(raw, dig, off, rg, pmin, pmax):
    L = orc.lib()
    pa = (raw.astype(np.float64) + off) * (rg / dig)
    x = np.where((pa < pmin) | (pa > pmax), 0.0, pa)
    med = L.orc_median(x.ctypes.data, x.size)
    return med, L.orc_madf(x.ctypes.data, x.size, med)


def shim_medmad(h, raw, dig, off, rg, pmin, pmax):
    med, mad, mr = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    rc = h.pgt_medmad(raw.ctypes.data, raw.size, dig, off, rg, pmin, pmax, ctypes.byref(med), ctypes.byref(mad), ctypes.byref(mr))
    return rc, med.value, mad.value, mr.value


@pytest.mark.parametrize("mode", range(8))
def test_fuzz_bit_exact(shim, mode):
    rng = np.random.default_rng(100 + mode)
    for trial in range(250):
        n = int(rng.choice([1, 2, 3, 4, 5, 7, 16, 100, 1000, 4000]))
        dig = float(rng.choice([2048.0, 8192.0])); rg = float(rng.uniform(200, 1500))
        off = float(rng.integers(-300, 300)) if mode != 5 else float(rng.uniform(-300, 300))
        raw = np.clip(np.rint(rng.normal(rng.uniform(300, 1500), rng.choice([1, 5, 50, 300, 3000]), n)), -32768, 32767).astype(np.int16)
        pmin, pmax = 40.0, 180.0
        if mode == 1: raw[rng.random(n) < 0.3] = rng.integers(-32768, 32767)
        if mode == 2: raw[:] = raw[0]
        if mode == 3: pmin = 100.0
        if mode == 4: pmin, pmax = -50.0, 60.0          # zero-filled samples sort into the middle of the values
        if mode == 6: pmin, pmax, off = float(rng.uniform(-100, 100)), float(rng.uniform(100, 400)), float(rng.integers(-3000, -500))
        if mode == 7: pmin, pmax = 0.0, 0.0
        rc, med, mad, mad_raw = shim_medmad(shim, raw, dig, off, rg, pmin, pmax)
        rm, rd = ref_medmad(raw, dig, off, rg, pmin, pmax)
        assert rc == 0
        assert np.float64(med).tobytes() == np.float64(rm).tobytes(), (mode, trial, n)
        assert np.float64(mad_raw * 1.4826).tobytes() == np.float64(rd).tobytes(), (mode, trial, n)
        assert mad == (rd if rd > 1.0 else 1.0)


def shim_medmad_sym(h, raw, dig, off, rg, pmin, pmax):
    med, mad, mr = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    rc = h.pgt_medmad_sym(raw.ctypes.data, raw.size, dig, off, rg, pmin, pmax, ctypes.byref(med), ctypes.byref(mad), ctypes.byref(mr))
    return rc, med.value, mad.value, mr.value


@pytest.mark.parametrize("mode", range(10))
def test_symmetric_fast_path_is_exact_whenever_it_applies(shim, mode):
    """The ring argument of pg_select.h (PgSym): wherever the fast path says "applicable" its doubles are the oracle's, bit
    for bit; and it does apply to ordinary reads (otherwise the kernel would always pay for the general search)."""
    rng = np.random.default_rng(900 + mode)
    applied = ordinary = ordinary_applied = 0
    trials = 400
    for trial in range(trials):
        n = int(rng.choice([2, 3, 4, 5, 7, 16, 100, 1000, 4000]))
        dig = float(rng.choice([2048.0, 8192.0])); rg = float(rng.uniform(200, 1500))
        off = float(rng.integers(-300, 300))
        sd = rng.choice([1, 5, 50, 300])
        raw = np.clip(np.rint(rng.normal(rng.uniform(300, 1500), sd, n)), -32768, 32767).astype(np.int16)
        pmin, pmax = 40.0, 180.0
        if mode == 1: raw[rng.random(n) < 0.05] = rng.integers(-32768, 32767)      # spikes: a zero-filled class far above the MAD
        if mode == 2: raw[rng.random(n) < 0.45] = rng.integers(-32768, 32767)      # ... of nearly half of the samples
        if mode == 3: off = float(rng.uniform(-300, 300))                           # fractional offsets: code + offset rounds
        if mode == 4: pmin, pmax = -50.0, 60.0
        if mode == 5: rg = float(10 ** rng.uniform(-9, 9))                          # tiny / huge scales: the guard decides
        if mode == 6: raw = (raw[0] + rng.integers(-1, 2, n)).astype(np.int16)     # three adjacent codes: ties in every ring
        if mode == 7: pmin, pmax = float(rng.uniform(60, 100)), float(rng.uniform(100, 140))   # narrow window: rings clipped on one side
        if mode == 8: off = float(rng.uniform(-1e13, 1e13)); rg = 1e14                # |offset| beyond the guard
        if mode == 9: raw = np.sort(raw)[:: int(rng.choice([1, -1]))].copy()
        rc, med, mad, mad_raw = shim_medmad_sym(shim, raw, dig, off, rg, pmin, pmax)
        pa = (raw.astype(np.float64) + off) * (rg / dig)
        inr = (pa >= pmin) & (pa <= pmax)
        # an ordinary read: nearly all samples in range, their spread small against the distance of the values from zero
        is_ordinary = n >= 16 and inr.mean() >= 0.9 and np.ptp(pa[inr]) < 0.5 * np.abs(pa[inr]).min()
        ordinary += is_ordinary
        if rc != 1:
            assert rc == 0
            continue
        applied += 1
        ordinary_applied += is_ordinary
        rm, rd = ref_medmad(raw, dig, off, rg, pmin, pmax)
        assert np.float64(med).tobytes() == np.float64(rm).tobytes(), (mode, trial, n)
        assert np.float64(mad_raw * 1.4826).tobytes() == np.float64(rd).tobytes(), (mode, trial, n, mad_raw * 1.4826, rd)
        assert mad == (rd if rd > 1.0 else 1.0)
    if mode in (0, 1, 3, 9):
        assert ordinary > 20 and ordinary_applied == ordinary, (ordinary, ordinary_applied, applied)


def test_fixture_read_known_answers(shim):
    G = os.path.join(ROOT, "tests", "golden", "single_read")
    raw = np.array([int(x) for x in [l for l in open(os.path.join(G, "reads.slow5")) if not l.startswith(("#", "@"))][0].split("\t")[7].split(",")], np.int16)
    rc, med, mad, _ = shim_medmad(shim, raw, 2048.0, -101.0, 281.345551, 40.0, 180.0)
    assert rc == 0 and abs(med - 111.5491149473) < 5e-11 and abs(mad - 20.1636564831) < 5e-11   # KA-2
    rc, med, mad, _ = shim_medmad(shim, raw, 2048.0, -101.0, 281.345551, 100.0, 180.0)
    assert rc == 0 and abs(med - 111.5491149473) < 5e-11 and abs(mad - 21.1820229721) < 5e-11   # KA-3


def test_plan_rejects_non_positive_scale(shim):
    out = (ctypes.c_int32 * 4)()
    assert shim.pgt_plan(2048.0, 0.0, -5.0, 40.0, 180.0, out) == -1
    assert shim.pgt_plan(0.0, 0.0, 5.0, 40.0, 180.0, out) == -1
    assert shim.pgt_plan(2048.0, -101.0, 281.345551, 40.0, 180.0, out) == 0
    c_lo, span = out[0], out[1]
    scale = 281.345551 / 2048.0
    assert (c_lo - 101.0) * scale >= 40.0 > (c_lo - 1 - 101.0) * scale
    assert (c_lo + span - 1 - 101.0) * scale <= 180.0 < (c_lo + span - 101.0) * scale


def test_plan_boundaries_match_brute_force(shim):
    """pg_make_plan's estimate-guided code search against an exhaustive scan of all 65536 codes, including bounds at or
    beyond the int16 range, infinite bounds and huge / tiny scales."""
    rng = np.random.default_rng(7)
    codes = np.arange(-32768, 32768, dtype=np.float64)
    cases = [(2048.0, -101.0, 281.345551, 40.0, 180.0), (8192.0, 5.0, 1400.0, -1e300, 1e300), (8192.0, 5.0, 1400.0, float("-inf"), float("inf")),
             (2048.0, 0.0, 1e-9, 40.0, 180.0), (2048.0, 0.0, 1e12, 40.0, 180.0), (2048.0, -40000.0, 300.0, 0.0, 10.0),
             (2048.0, 40000.0, 300.0, 0.0, 10.0), (2048.0, 0.0, 300.0, 180.0, 40.0), (8192.0, 0.5, 1000.0, 0.0, 0.0)]
    for _ in range(300):
        cases.append((float(rng.choice([2048.0, 8192.0])), float(rng.uniform(-40000, 40000)), float(10 ** rng.uniform(-3, 6)),
                      float(rng.uniform(-500, 500)), float(rng.uniform(-500, 5000))))
    for dig, off, rg, pmin, pmax in cases:
        out = (ctypes.c_int32 * 4)()
        assert shim.pgt_plan(dig, off, rg, pmin, pmax, out) == 0
        pa = (codes + off) * (rg / dig)
        ge_min = np.flatnonzero(~(pa < pmin)); gt_max = np.flatnonzero(pa > pmax); ge0 = np.flatnonzero(pa >= 0.0)
        c_lo = int(codes[ge_min[0]]) if ge_min.size else 32768
        c_gt = int(codes[gt_max[0]]) if gt_max.size else 32768
        c_z = int(codes[ge0[0]]) if ge0.size else 32768
        span = max(c_gt - c_lo, 0)
        assert (out[0], out[1], out[2]) == (c_lo, span, min(max(c_z - c_lo, 0), span)), (dig, off, rg, pmin, pmax)


# ---- the dense gathers' division (pg_select.h: pg_div_by_recip) -------------------------------------------------
def _near_midpoint_operands(rng, n_divisors):
    """Quotients of two doubles as close to a rounding midpoint as they can come: for an odd 53-bit B and a small odd t,
    M = t / B (mod 2^54) is the odd numerator of a midpoint M / 2^54 of (1/2, 1) and A = (M B - t) / 2^54 gives
    A / B = M / 2^54 - t / (2^54 B), i.e. |t| * 2^-107 .. 2^-106 off the midpoint (the same construction one binade up for A >= B)."""
    out = []
    for i in range(n_divisors):
        B = int(rng.integers(0, 1 << 52)) | (1 << 52) | 1
        if i % 4 == 0:
            B = (1 << 53) - 1 - 2 * int(rng.integers(0, 1 << 12))   # significands next to 2
        if i % 4 == 1:
            B = (1 << 52) + 1 + 2 * int(rng.integers(0, 1 << 12))   # significands next to 1
        for bits in (54, 53):
            inv = pow(B, -1, 1 << bits)
            for t in range(-9, 10, 2):
                M = (t * inv) % (1 << bits)
                if bits == 53:
                    M += 1 << 53
                if not ((1 << 53) < M < (1 << 54)):
                    continue
                P = M * B - t
                assert P % (1 << bits) == 0
                A = P >> bits
                if A >= (1 << 53) and (A & 1 or A >= (1 << 54)):
                    continue  # not a double
                e = int(rng.integers(0, 9))
                out.append((float(A) * 2.0 ** (e - 52 + int(rng.integers(-30, 31))) * (-1.0 if rng.integers(0, 2) else 1.0), float(B) * 2.0 ** (e - 52)))
    return out


def test_div_by_recip_is_the_correctly_rounded_quotient(shim):
    shim.pgt_div_by_recip.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    rng = np.random.default_rng(20251005)
    cases = _near_midpoint_operands(rng, 1500)
    assert len(cases) > 10000
    # the significand pairs DESIGN.md section 6's bound leaves over: a in {2 - 4u, 2 - 6u}, b above it (u = 2^-53), at many exponent offsets
    u2 = 2.0 ** -52
    for a, b in ((2 - 2 * u2, 2 - u2), (2 - 3 * u2, 2 - u2), (2 - 3 * u2, 2 - 2 * u2)):
        for eb in range(0, 10):
            for ea in range(-60, 61, 4):
                cases.append((a * 2.0 ** (eb + ea), b * 2.0 ** eb))
                cases.append((-a * 2.0 ** (eb + ea), b * 2.0 ** eb))
    # the gather's own operands: x - median over pA-like values, MAD in [1, 60]
    for _ in range(20000):
        cases.append((float(rng.uniform(40, 180) - rng.uniform(60, 140)), float(rng.uniform(1, 60))))
    cases += [(0.0, 3.0), (5.0, 1.0), (1.0, 3.0), (-1.0, 3.0)]
    qr, qd = ctypes.c_double(), ctypes.c_double()
    for a, b in cases:
        same = shim.pgt_div_by_recip(a, b, ctypes.byref(qr), ctypes.byref(qd))
        assert same == 1 and qd.value == a / b, (a.hex(), b.hex(), qr.value.hex(), qd.value.hex())
    # the one operand the sequence gets wrong, and why it cannot occur: a = -0.0 gives +0.0 (the residual fma(-b, -0, -0) is +0), but
    # x - median is -0.0 only for x = -0.0, and inside pg_div_domain_ok no pA underflows (the zero fill writes +0.0; raw + offset == 0 is +0.0)
    assert shim.pgt_div_by_recip(-0.0, 1.0, ctypes.byref(qr), ctypes.byref(qd)) == 0 and qr.value == 0.0


def test_div_domain_guard(shim):
    shim.pgt_div_domain_ok.argtypes = [ctypes.c_double, ctypes.c_double]
    ok = shim.pgt_div_domain_ok
    assert ok(-243.0, 0.1373) == 1 and ok(0.0, 1.0) == 1 and ok(10.0, 2.0 ** -200) == 1
    assert ok(1e-300, 0.1373) == 0 and ok(-243.0, 1e-70) == 0 and ok(-243.0, 1e70) == 0 and ok(1e80, 1.0) == 0
    assert ok(float("nan"), 1.0) == 0 and ok(1.0, float("inf")) == 0
