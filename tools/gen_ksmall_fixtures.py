#!/usr/bin/env python3
"""Writes tests/golden/ksmall_vectors.json: median / MAD of ~200 seeded reads computed by the REFERENCE's own quickselect
(/root/reference/src/ksort.h:233-259 compiled into oracle/_ref/libref_selection.so by `make -C oracle ref`; build container only).
The vectors come from tests/ksmall_vectors.py; the fixture holds the recipe, a CRC of each vector and the expected doubles as bit patterns."""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ksmall_vectors as kv

lib = C.CDLL(os.path.join(ROOT, "oracle/_ref/libref_selection.so"))
lib.ref_read_medmad.argtypes = [C.c_void_p, C.c_size_t] + [C.c_double] * 5 + [C.POINTER(C.c_double)]
lib.ref_read_medmad.restype = None
vecs = []
for sp in kv.specs():
    raw = kv.make_raw(sp)
    dig, off, rng = kv.CALS[sp["cal"]]
    out = (C.c_double * 3)()
    lib.ref_read_medmad(raw.ctypes.data, raw.size, dig, off, rng, sp["pa_min"], sp["pa_max"], out)
    e = dict(sp); e.update(crc=kv.crc(raw), med=kv.bits(out[0]), madf=kv.bits(out[1]), mad=kv.bits(out[2]))
    if raw.size <= 64: e["raw"] = raw.tolist()
    vecs.append(e)
doc = {"what": "upper median, calc_madf and clamped MAD per read, from the reference's ks_ksmall_double (src/ksort.h:233-259) via oracle/ref_selection.c",
       "generator": "tools/gen_ksmall_fixtures.py + tests/ksmall_vectors.py", "calibrations": {k: list(v) for k, v in kv.CALS.items()}, "vectors": vecs}
path = os.path.join(ROOT, "tests/golden/ksmall_vectors.json")
with open(path, "w") as f:
    json.dump(doc, f, indent=0, separators=(",", ":"))
print(len(vecs), "vectors ->", path, os.path.getsize(path), "bytes")
