"""The seeded read vectors behind tests/golden/ksmall_vectors.json (median / MAD selection, gmove.cpp:142-184, 751-771).

Integer-only generation (splitmix64 in numpy uint64): the same int16 samples on every numpy. The fixture stores a CRC of each
vector's bytes, so drift of this module is detected instead of silently changing what the expected values mean."""
import zlib

import numpy as np

M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed: int, n: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (np.uint64(seed) + (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)))
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


CALS = {  # digitisation, offset, range
    "r9": (8192.0, 12.0, 1437.976685),
    "r10": (2048.0, -243.0, 281.345551),
    "odd": (2048.0, -101.5, 283.1),
}
LENGTHS = [1, 2, 3, 4, 5, 7, 8, 16, 17, 63, 64, 65, 127, 128, 1000, 1001, 4000, 4001, 16384, 32768, 32769, 49153, 100000, 100001]
KINDS = ["normal", "ties", "const", "allzero", "halfzero_lo", "halfzero_eq", "halfzero_hi", "wide", "huge"]


def specs():
    out = []
    for li, n in enumerate(LENGTHS):
        for ki, kind in enumerate(KINDS):
            cal = ["r10", "r9", "odd"][(li + ki) % 3]
            pa = (40.0, 180.0) if (li + 2 * ki) % 4 else (-1e9, 1e9)   # a quarter of the vectors keep every sample
            if kind.startswith("halfzero") or kind == "allzero":
                pa = (40.0, 180.0)
            out.append(dict(id=f"{kind}_n{n}", kind=kind, n=n, seed=1000 * li + ki + 1, cal=cal, pa_min=pa[0], pa_max=pa[1]))
    return out


def make_raw(spec) -> np.ndarray:
    n, kind = spec["n"], spec["kind"]
    dig, off, rng = CALS[spec["cal"]]
    scale = rng / dig
    centre = int(round(95.0 / scale - off))      # ~95 pA
    u = splitmix64(spec["seed"], n)
    if kind in ("normal", "halfzero_lo", "halfzero_eq", "halfzero_hi"):
        s4 = ((u & np.uint64(0xFFFF)) + ((u >> np.uint64(16)) & np.uint64(0xFFFF)) + ((u >> np.uint64(32)) & np.uint64(0xFFFF)) + (u >> np.uint64(48))).astype(np.int64)
        sd = max(2, int(12.0 / scale))          # ~12 pA
        raw = centre + (((s4 - 131070) * sd) >> 16)
        if kind != "normal":                      # zero-filled samples: exactly n/2 - 1, n/2, n/2 + 1 of them -> the median sits next to the 0.0 class
            z = {"halfzero_lo": n // 2 - 1, "halfzero_eq": n // 2, "halfzero_hi": n // 2 + 1}[kind]
            z = max(0, min(n, z))
            idx = np.argsort(splitmix64(spec["seed"] + 77, n), kind="stable")[:z]
            lowraw = int(round(10.0 / scale - off))   # 10 pA: below pa_min 40
            raw[idx] = lowraw
    elif kind == "ties":
        levels = centre + np.array([-3, 0, 0, 2, 2, 2, 9], np.int64)
        raw = levels[(u % np.uint64(7)).astype(np.int64)]
    elif kind == "const":
        raw = np.full(n, centre + 5, np.int64)
    elif kind == "allzero":
        raw = np.full(n, int(round(300.0 / scale - off)), np.int64) + (u % np.uint64(3)).astype(np.int64)
    elif kind == "wide":                          # in-range interval of ~3000 codes: beyond the 1024-bin LDS histogram
        raw = centre - 1500 + (u % np.uint64(3000)).astype(np.int64)
    elif kind == "huge":                          # the whole int16 range
        raw = (u % np.uint64(65536)).astype(np.int64) - 32768
    else:
        raise ValueError(kind)
    return np.clip(raw, -32768, 32767).astype(np.int16)


def pa_zero_filled(raw, spec) -> np.ndarray:
    """gmove.cpp:751-760 in numpy: TO_PICOAMPS with the reference's operation order, zero where out of [pa_min, pa_max]."""
    dig, off, rng = CALS[spec["cal"]]
    pa = (raw.astype(np.float64) + off) * (rng / dig)
    x = np.zeros(raw.size, np.float64)
    keep = ~((pa < spec["pa_min"]) | (pa > spec["pa_max"]))
    x[keep] = pa[keep]
    return x


def crc(raw) -> int:
    return zlib.crc32(np.ascontiguousarray(raw).tobytes()) & 0xFFFFFFFF


def bits(v: float) -> str:
    return "%016x" % np.float64(v).view(np.uint64)
