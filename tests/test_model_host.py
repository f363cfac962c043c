"""The k-mer model step (scripts/poregen.sh:54-85, 33-52) on the host: the fixed-point view of "%.8f", the library's
finishing arithmetic, and the CPU oracle (oracle/model_oracle.c) against independent exact arithmetic (Python
Decimal / Fraction). datamash itself is not available here: parity with the real tool is unpinned (see the oracle's header);
what these tests pin is that oracle, library arithmetic and exact mathematics agree."""
import ctypes as C
import os
import random
import subprocess
from decimal import Decimal, getcontext
from fractions import Fraction

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(os.environ.get("PG_ORACLE_DIR") or os.path.join(ROOT, "oracle"), "model_oracle")


@pytest.fixture(scope="module")
def host():
    L = C.CDLL((os.environ.get("PG_HOSTTEST_SO") or os.path.join(ROOT, "poregen_amd", "_pg_hosttest.so")))
    L.pgt_fixed8.argtypes = [C.c_double, C.POINTER(C.c_int)]; L.pgt_fixed8.restype = C.c_longlong
    L.pgt_model_texts.argtypes = [C.POINTER(C.c_longlong), C.c_size_t, C.c_char_p, C.c_char_p, C.c_size_t]
    L.pgt_sstdev_text.argtypes = [C.c_ulonglong, C.c_ulonglong, C.c_ulonglong, C.c_char_p, C.c_char_p, C.c_size_t]
    return L


@pytest.fixture(scope="module", autouse=True)
def _oracle_built():
    if not os.path.exists(ORACLE):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])


def g14(x: Fraction) -> str:
    """'%.14g' of an exact rational (what "%.14Lg" prints when the long double carries no visible error)."""
    getcontext().prec = 60
    d = Decimal(x.numerator) / Decimal(x.denominator)
    return "%.14g" % float(d) if d == 0 else format(d, ".14g").replace("E", "e")


def norm_g(s: str) -> Decimal:
    return Decimal(s)


def exact_stats(units):
    """median and sample variance of integers (1e-8 units) as exact rationals of the value scale"""
    v = sorted(units); n = len(v)
    med = Fraction(v[n // 2], 10**8) if n & 1 else Fraction(v[n // 2 - 1] + v[n // 2], 2 * 10**8)
    if n < 2:
        return med, None
    s1 = sum(units); s2 = sum(u * u for u in units)
    var = Fraction(n * s2 - s1 * s1, n * (n - 1)) / 10**16
    return med, var


def close14(text: str, exact: Decimal, ulps: int = 1) -> bool:
    """text == exact rounded to 14 significant digits, give or take `ulps` units of the 14th digit"""
    getcontext().prec = 60
    if exact == 0:
        return Decimal(text) == 0
    unit = Decimal(1).scaleb(exact.adjusted() - 13)
    return abs(Decimal(text) - exact) <= unit * (Decimal("0.5") + ulps)


def sqrt_text14(var: Fraction) -> Decimal:
    getcontext().prec = 60
    return (Decimal(var.numerator) / Decimal(var.denominator)).sqrt()


def test_fixed8_is_printf_8f(host):
    rng = random.Random(7)
    xs = [0.0, -0.0, 0.001953125, -0.001953125, 100.001953125, 100.005859375, 0.5e-8, 1.5e-8, 2.5e-8, 1e-9, -1e-9, 179.99999999499,
          126.79782401, 54.675551, 3.9e7, -3.9e7, 1.0 / 3, 2.0 / 3, 123456.7890123456]
    xs += [k / 512.0 for k in range(-2000, 2000, 7)]                      # exact ties: odd multiples of 2^-9 sit on xxx.5 units
    xs += [rng.uniform(-200, 200) for _ in range(20000)] + [rng.uniform(-3, 3) for _ in range(20000)]
    xs += [rng.uniform(-1e7, 1e7) for _ in range(5000)]
    bad = C.c_int(0)
    for x in xs:
        got = host.pgt_fixed8(x, C.byref(bad))
        assert bad.value == 0
        want = int(Decimal("%.8f" % x).scaleb(8))
        assert got == want, (x, got, want)
    for x in (float("nan"), float("inf"), -float("inf"), 4.0e7, -1e300):
        host.pgt_fixed8(x, C.byref(bad))
        assert bad.value == 1


def test_library_arithmetic_is_exact(host):
    rng = random.Random(11)
    for trial in range(300):
        n = rng.choice([1, 2, 3, 4, 5, 10, 99, 100, 1001])
        centre = rng.choice([0, 75_00000000, -1_50000000, 111_54911495])
        spread = rng.choice([1, 1000, 3_00000000, 60_00000000, 10000_00000000])
        units = [centre + rng.randrange(-spread, spread + 1) for _ in range(n)]
        arr = (C.c_longlong * n)(*units)
        med, sd = C.create_string_buffer(64), C.create_string_buffer(64)
        host.pgt_model_texts(arr, n, med, sd, 64)
        emed, evar = exact_stats(units)
        assert Decimal(med.value.decode()) == Decimal(g14(emed)), (units[:5], med.value, g14(emed))
        if n < 2:
            assert sd.value == b"nan"
        elif evar == 0:
            assert Decimal(sd.value.decode()) == 0
        else:
            want = sqrt_text14(evar)
            assert close14(sd.value.decode(), want, ulps=0), (sd.value, want)   # exact moments: correctly rounded text


def test_sstdev_text_next_to_a_rounding_boundary(host):
    """The 14 digits of the sstdev text are decided in exact integer arithmetic (pg_model_sstdev_text): moments whose standard deviation
    sits one part in 10^29 below / above / exactly on the boundary between two 14-digit decimals -- far inside the long double's own
    rounding error, where the plain "%.14Lg" of the square root gets about half of them wrong (the fuzzer met one such file in ~1600)."""
    rng = random.Random(5)
    plain_wrong = 0
    for trial in range(400):
        n = rng.choice([2, 3, 10, 1000])
        den = n * (n - 1)
        D = rng.randrange(10**13, 10**14 - 1)
        e = rng.choice([-9, -8, -7])                       # sd = D * 10^e: 1e4 .. 1e7 sample units
        # boundary b = (2D + 1) / 2 * 10^e; sd = b  <=>  num = b^2 * den * 10^16 = (2D + 1)^2 * den * 10^(2e + 16) / 4
        tie_num = Fraction((2 * D + 1) ** 2 * den * 10 ** (2 * e + 18), 400)
        base = tie_num.numerator // tie_num.denominator
        for num in (base - 1, base, base + 1, base + 2):
            if num >= 1 << 128:
                continue
            exact, plain = C.create_string_buffer(64), C.create_string_buffer(64)
            host.pgt_sstdev_text(n, num >> 64, num & ((1 << 64) - 1), exact, plain, 64)
            var = Fraction(num, den) / 10**16
            want = sqrt_text14(var)
            on_tie = Fraction(num) == tie_num
            if on_tie:                                     # exactly half way: ties to even on the 14th digit
                assert Decimal(exact.value.decode()) == Decimal(D + (D & 1)).scaleb(e), (n, D, e, exact.value)
            else:
                assert close14(exact.value.decode(), want, ulps=0), (n, D, e, num - base, exact.value, want)
                plain_wrong += not close14(plain.value.decode(), want, ulps=0)
    assert plain_wrong > 50                                # the test does exercise what the long double cannot decide


def write_dir(tmp_path, files):
    d = tmp_path / "dump"
    d.mkdir()
    for name, text in files.items():
        (d / name).write_text(text)
    return d


def run_oracle(mode, d, *args):
    return subprocess.run([ORACLE, mode, str(d)] + list(args), capture_output=True, check=True).stdout.decode()


def test_oracle_on_the_fixture_known_answer(tmp_path):
    """dump/TGTGTG of the reference's single-read fixture: ten values, the first is dropped."""
    text = "126.79782401,126.79782401,125.56144219,128.44633310,126.79782401;149.60219973,147.95369064,149.05269670,148.77794518,147.81631488;"
    d = write_dir(tmp_path, {"TGTGTG": text, "ATGTTG": "", "AAAAAA": "101.5;", "CCCCCC": "1.25,2.50;"})
    out = run_oracle("stats", d, "3.1").splitlines()
    assert [l.split("\t")[0] for l in out] == ["AAAAAA", "ATGTTG", "CCCCCC", "TGTGTG"]           # glob order
    assert out[0] == "AAAAAA\t\t" and out[1] == "ATGTTG\t\t"                                        # nothing reaches datamash
    assert out[2] == "CCCCCC\t2.5\tnan"                                                             # one value
    vals = [Decimal(x) for x in text.replace(";", ",").strip(",").split(",")][1:]
    units = [int(v.scaleb(8)) for v in vals]
    emed, evar = exact_stats(units)
    name, med, sd = out[3].split("\t")
    assert med == "147.81631488" and Decimal(med) == Decimal(g14(emed))
    assert sd == "3.1" and sqrt_text14(evar) > Decimal("3.1")                                       # capped at the limit text
    sd_uncapped = run_oracle("stats", d, "1000").splitlines()[3].split("\t")[2]
    assert close14(sd_uncapped, sqrt_text14(evar))
    dw = run_oracle("dwell", d).splitlines()
    assert dw == ["AAAAAA\t0", "ATGTTG\t", "CCCCCC\t0.5", "TGTGTG\t4"]                              # fields: 4,4,0 -> median 4


def test_oracle_matches_exact_arithmetic_on_random_files(tmp_path, host):
    rng = random.Random(3)
    files, expect = {}, {}
    for i in range(40):
        name = "".join(rng.choice("ACGT") for _ in range(6)) + str(i)
        n_ev = rng.choice([1, 2, 3, 50, 400])
        evs = [[rng.uniform(-3, 3) if i % 2 else rng.uniform(50, 170) for _ in range(rng.randint(1, 40))] for _ in range(n_ev)]
        text = "".join(",".join("%.8f" % x for x in e) + ";" for e in evs)
        files[name] = text
        flat = [x for e in evs for x in e]
        bad = C.c_int(0)
        units = [host.pgt_fixed8(x, C.byref(bad)) for x in flat][1:]
        lens = sorted([len(e) - 1 for e in evs] + [0])
        nd = len(lens)
        dmed = Fraction(lens[nd // 2]) if nd & 1 else Fraction(lens[nd // 2 - 1] + lens[nd // 2], 2)
        expect[name] = (units, dmed)
    d = write_dir(tmp_path, files)
    lines = {l.split("\t")[0]: l.split("\t") for l in run_oracle("stats", d, "1e9").splitlines()}
    dlines = {l.split("\t")[0]: l.split("\t") for l in run_oracle("dwell", d).splitlines()}
    assert list(lines) == sorted(files)
    for name, (units, dmed) in expect.items():
        _, med, sd = lines[name]
        assert Decimal(dlines[name][1]) == Decimal(g14(dmed))
        if not units:
            assert med == "" and sd == ""
            continue
        emed, evar = exact_stats(units)
        assert Decimal(med) == Decimal(g14(emed)), name
        if evar is None:
            assert sd == "nan"
        else:
            want = sqrt_text14(evar)
            assert close14(sd, want), (name, sd, want)
        # the library's arithmetic prints the same text as the oracle's (median always; sstdev up to the last digit)
        arr = (C.c_longlong * len(units))(*units)
        m2, s2 = C.create_string_buffer(64), C.create_string_buffer(64)
        host.pgt_model_texts(arr, len(units), m2, s2, 64)
        assert m2.value.decode() == med
        if evar is not None and evar != 0:
            assert close14(s2.value.decode(), sqrt_text14(evar), ulps=0)
