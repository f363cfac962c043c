"""ctypes bridge to the CPU oracle (oracle/libgmove_oracle.so) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; nothing under
poregen_amd/ does. See oracle/gmove_oracle.h for what the oracle is and how it is pinned.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.environ.get("PG_ORACLE_DIR") or os.path.dirname(os.path.abspath(__file__))  # PG_ORACLE_DIR: the sanitizer builds (make asan)
LIB = os.path.join(_HERE, "libgmove_oracle.so")
CLI = os.path.join(_HERE, "gmove_oracle")

ORC_OK, ORC_SKIPPED, ORC_STOPPED = 0, 1, 2
ORC_ERR_RNA_FLAG, ORC_ERR_BAD_SS, ORC_ERR_INTERNAL, ORC_ERR_ASSERT, ORC_ERR_UNDEFINED = -1, -2, -3, -4, -5


class OrcOpt(C.Structure):
    _fields_ = [
        ("kmer_size", C.c_uint32), ("sig_move_offset", C.c_uint32), ("kmer_start_offset", C.c_uint32),
        ("scaling", C.c_int32), ("signal_print_margin", C.c_uint32), ("sample_limit", C.c_uint32),
        ("index_start", C.c_uint32), ("index_end", C.c_uint32), ("delimit_files", C.c_int32),
        ("max_dur", C.c_uint32), ("min_dur", C.c_uint32), ("pa_min", C.c_double), ("pa_max", C.c_double),
        ("kmer_pick_margin", C.c_int32), ("flag_rna", C.c_int32),
    ]


_lib = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        L.orc_default_opt.argtypes = [C.POINTER(OrcOpt)]
        L.orc_create.argtypes = [C.POINTER(OrcOpt), C.POINTER(C.c_char_p), C.c_size_t]; L.orc_create.restype = C.c_void_p
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_paf_read.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_double, C.c_double, C.c_double, C.c_int32, C.c_int32,
                                   C.c_int32, C.c_char_p, C.c_int64, C.c_char_p]
        L.orc_paf_read.restype = C.c_int
        L.orc_table_read.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_double, C.c_double, C.c_double, C.c_int32, C.c_char_p,
                                     C.c_int32, C.c_char_p, C.c_uint64, C.c_int32]
        L.orc_table_read.restype = C.c_int
        L.orc_n_slots.argtypes = [C.c_void_p]; L.orc_n_slots.restype = C.c_size_t
        L.orc_slot_kmer.argtypes = [C.c_void_p, C.c_size_t]; L.orc_slot_kmer.restype = C.c_char_p
        L.orc_slot_count.argtypes = [C.c_void_p, C.c_size_t]; L.orc_slot_count.restype = C.c_uint64
        L.orc_slot_text.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]; L.orc_slot_text.restype = C.c_void_p
        L.orc_slot_n_values.argtypes = [C.c_void_p, C.c_size_t]; L.orc_slot_n_values.restype = C.c_size_t
        L.orc_slot_values.argtypes = [C.c_void_p, C.c_size_t]; L.orc_slot_values.restype = C.c_void_p
        L.orc_slot_event_lens.argtypes = [C.c_void_p, C.c_size_t]; L.orc_slot_event_lens.restype = C.c_void_p
        L.orc_total_samples.argtypes = [C.c_void_p]; L.orc_total_samples.restype = C.c_uint64
        L.orc_reads_seen.argtypes = [C.c_void_p]; L.orc_reads_seen.restype = C.c_uint64
        L.orc_last_medmad.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_median.argtypes = [C.c_void_p, C.c_size_t]; L.orc_median.restype = C.c_double
        L.orc_madf.argtypes = [C.c_void_p, C.c_size_t, C.c_double]; L.orc_madf.restype = C.c_double
        L.orc_write_outputs.argtypes = [C.c_void_p, C.c_char_p]; L.orc_write_outputs.restype = C.c_int
        _lib = L
    return _lib


class Oracle:
    """One sequential gmove run on the CPU oracle. kmers = the FULL list; the slice is 1-based closed."""

    def __init__(self, kmers, index_start=1, index_end=None, **opt):
        L = lib()
        self.L = L
        o = OrcOpt()
        L.orc_default_opt(C.byref(o))
        o.index_start = index_start
        o.index_end = len(kmers) if index_end is None else index_end
        for k, v in opt.items():
            assert hasattr(o, k), k
            setattr(o, k, v)
        self.opt = o
        arr = (C.c_char_p * len(kmers))(*[k.encode() for k in kmers])
        self.h = L.orc_create(C.byref(o), arr, len(kmers))
        if not self.h:
            raise ValueError("oracle rejected the k-mer list / slice")
        self.n_slots = L.orc_n_slots(self.h)
        self.medmad = []

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_destroy(self.h)
            self.h = None

    def paf_read(self, raw, dig, off, rng, query_start, target_start, target_end, target_seq, ss):
        raw = np.ascontiguousarray(raw, dtype=np.int16)
        ts = None if target_seq is None else target_seq.encode()
        rc = self.L.orc_paf_read(self.h, raw.ctypes.data, raw.size, dig, off, rng, query_start, target_start, target_end,
                                 ts, 0 if ts is None else len(ts), ss.encode())
        return rc

    def run_batch(self, b, full_target=True, record_medmad=False):
        """Feed a poregen_amd Batch (host) read by read. The batch holds the FETCHED range of each target; the
        oracle re-applies faidx_fetch_seq's clamping to the full sequence, so rebuild a full target by padding
        the bases before min(ts,te) with 'N'."""
        from poregen_amd import synth  # generator helpers only (ss/seq strings); no engine code
        rcs = []
        for r in range(b.n_reads):
            ts, te = int(b.target_start[r]), int(b.target_end[r])
            seq = synth.seq_string(b, r)
            full = "N" * min(ts, te) + seq
            raw = b.sig[int(b.sig_off[r]):int(b.sig_off[r + 1])]
            rc = self.paf_read(raw, float(b.digitisation[r]), float(b.offset[r]), float(b.range[r]), int(b.query_start[r]), ts, te,
                               full, synth.ss_string(b, r))
            if record_medmad:
                m, d = C.c_double(), C.c_double()
                self.L.orc_last_medmad(self.h, C.byref(m), C.byref(d))
                self.medmad.append((m.value, d.value))
            rcs.append(rc)
            if rc == ORC_STOPPED:
                break
        return rcs

    def count(self, s):
        return int(self.L.orc_slot_count(self.h, s))

    def counts(self):
        return np.array([self.count(s) for s in range(self.n_slots)], dtype=np.uint64)

    def text(self, s):
        n = C.c_size_t()
        p = self.L.orc_slot_text(self.h, s, C.byref(n))
        return C.string_at(p, n.value).decode()

    def values(self, s):
        n = self.L.orc_slot_n_values(self.h, s)
        if n == 0:
            return np.zeros(0)
        p = self.L.orc_slot_values(self.h, s)
        return np.frombuffer((C.c_char * (8 * n)).from_address(p), dtype=np.float64).copy()

    def all_values(self):
        """Every slot's kept samples back to back in slot order (the k-mer-major stream the product returns), one memmove per slot."""
        ns = [self.L.orc_slot_n_values(self.h, s) for s in range(self.n_slots)]
        out = np.empty(int(sum(ns)), dtype=np.float64)
        pos = 0
        for s, n in enumerate(ns):
            if n:
                C.memmove(out.ctypes.data + 8 * pos, self.L.orc_slot_values(self.h, s), 8 * n)
                pos += n
        return out

    def all_event_lens(self):
        cnt = self.counts()
        out = np.empty(int(cnt.sum()), dtype=np.uint32)
        pos = 0
        for s in np.flatnonzero(cnt):
            n = int(cnt[s])
            C.memmove(out.ctypes.data + 4 * pos, self.L.orc_slot_event_lens(self.h, int(s)), 4 * n)
            pos += n
        return out

    def event_lens(self, s):
        n = self.count(s)
        if n == 0:
            return np.zeros(0, np.uint32)
        p = self.L.orc_slot_event_lens(self.h, s)
        return np.frombuffer((C.c_char * (4 * n)).from_address(p), dtype=np.uint32).copy()

    def kmer(self, s):
        return self.L.orc_slot_kmer(self.h, s).decode()

    def total_samples(self):
        return int(self.L.orc_total_samples(self.h))
