"""Host-side mirror of the gmove seam over libpgmove's C ABI (include/pgmove.h).

Names follow the reference's vocabulary (reads, ss ops, k-mer slice, slots, dump events); see
src/gmove.cpp:707-975 of hiruna72/poregen for the loop this replaces. All computation happens in
libpgmove.so on the GPU; this module only marshals pointers (numpy arrays for host batches, torch
tensors for device-resident batches).
"""
import ctypes as C
import itertools
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import _abi


class PgError(RuntimeError):
    def __init__(self, status, text):
        super().__init__(f"libpgmove status {status}: {text}")
        self.status = status
        self.text = text


def generate_kmers(k: int, rna: bool = False) -> List[str]:
    """Lexicographic 4^k k-mers over ACGT / ACGU (generate_kmers, src/poregen.cpp:248-267)."""
    return ["".join(t) for t in itertools.product("ACGU" if rna else "ACGT", repeat=k)]


def reconcile_slice(n_kmers: int, index_start: int = 1, index_end: int = 500, file_limit: int = 500):
    """The slice reconciliation of gmove() (src/gmove.cpp:428-440) -> final (index_start, index_end)."""
    if file_limit < n_kmers:
        pass
    elif file_limit > n_kmers - index_start + 1:
        if index_end > n_kmers:
            file_limit = n_kmers - index_start + 1
            index_end = index_start + file_limit - 1
    return index_start, index_end


@dataclass
class GmoveParams:
    kmers: Sequence[str]                 # the k-mer slice, in list order: slot i = kmers[i]
    kmer_size: int = 9
    sig_move_offset: int = 0
    margin: int = 0
    sample_limit: int = 100
    max_dur: int = 70
    min_dur: int = 5
    kmer_pick_margin: int = 2
    scaling: int = 0
    rna: bool = False
    pa_min: float = 40.0
    pa_max: float = 180.0
    device: int = 0
    lazy_stats: bool = False
    profile: bool = False
    overlap: Optional[bool] = None       # None: the library's default (two streams when it computes eager statistics); True / False: PG_FLAG_OVERLAP / PG_FLAG_ONE_STREAM
    debug_narrow: bool = False
    overlap_tail: bool = False           # statistics on a second stream next to pg_collect's small launches (PG_FLAG_OVERLAP_TAIL)
    split_walk: bool = False             # measurement/tests: ss walk and event filter as two launches (PG_FLAG_DEBUG_SPLIT_WALK)
    defer_stats: bool = False            # multi-GPU step: the statistics are queued by stats() (behind the issue of the collective) or by collect()
    stop_when_full: bool = False         # kmers is the WHOLE list: errors behind the read that completes the last k-mer do not count


_BATCH_FIELDS = [
    ("sig", np.int16), ("sig_off", np.uint64), ("digitisation", np.float64), ("offset", np.float64),
    ("range", np.float64), ("query_start", np.int32), ("target_start", np.int32), ("target_end", np.int32),
    ("seq", np.uint8), ("seq_off", np.uint64), ("op_n", np.uint32), ("op_t", np.uint8), ("op_off", np.uint64),
]


@dataclass
class Batch:
    """One batch of reads in PAF-line order (layout: pg_batch in include/pgmove.h)."""
    n_reads: int
    sig: object
    sig_off: object
    digitisation: object
    offset: object
    range: object
    query_start: object
    target_start: object
    target_end: object
    seq: object
    seq_off: object
    op_n: object
    op_t: object
    op_off: object
    on_device: bool = False
    n_ops: int = 0   # device batches: op_off[n_reads] when known (pg_batch.n_ops); 0 = the library reads it back (one sync per call)
    all_matches: bool = False  # the batch holds match ops only and the caller vouches for it (PG_BATCH_ALL_MATCHES, verified on the device)
    resident: bool = False     # device batch, engine on the caller's stream: nothing queued on that stream produces the arrays (PG_BATCH_RESIDENT)

    def validate_host(self):
        for name, dt in _BATCH_FIELDS:
            a = getattr(self, name)
            assert isinstance(a, np.ndarray) and a.dtype == dt and a.flags["C_CONTIGUOUS"], (name, getattr(a, "dtype", None))
        return self

    def to_device(self, device):
        import torch
        kw = {}
        for name, dt in _BATCH_FIELDS:
            a = getattr(self, name)
            # torch has no uint64/uint32: ship the bytes as int64/int32 of the same width
            view = {np.uint64: np.int64, np.uint32: np.int32}.get(dt, dt)
            t = torch.from_numpy(np.ascontiguousarray(a).view(view))
            if name == "sig":  # 16-byte slack so the tail vector of the last read stays inside the allocation
                t = torch.cat([t, torch.zeros(8, dtype=t.dtype)])
            kw[name] = t.to(device)
        return Batch(n_reads=self.n_reads, on_device=True, n_ops=int(self.op_off[-1]), all_matches=not bool(np.any(self.op_t)), **kw)

    def slice_reads(self, lo: int, hi: int) -> "Batch":
        """Host batch holding reads [lo, hi) (used to shard a PAF-ordered batch across ranks)."""
        assert not self.on_device
        so, qo, oo = self.sig_off, self.seq_off, self.op_off
        return Batch(
            n_reads=hi - lo,
            sig=np.ascontiguousarray(self.sig[int(so[lo]):int(so[hi])]), sig_off=(so[lo:hi + 1] - so[lo]).astype(np.uint64),
            digitisation=np.ascontiguousarray(self.digitisation[lo:hi]), offset=np.ascontiguousarray(self.offset[lo:hi]),
            range=np.ascontiguousarray(self.range[lo:hi]), query_start=np.ascontiguousarray(self.query_start[lo:hi]),
            target_start=np.ascontiguousarray(self.target_start[lo:hi]), target_end=np.ascontiguousarray(self.target_end[lo:hi]),
            seq=np.ascontiguousarray(self.seq[int(qo[lo]):int(qo[hi])]), seq_off=(qo[lo:hi + 1] - qo[lo]).astype(np.uint64),
            op_n=np.ascontiguousarray(self.op_n[int(oo[lo]):int(oo[hi])]), op_t=np.ascontiguousarray(self.op_t[int(oo[lo]):int(oo[hi])]),
            op_off=(oo[lo:hi + 1] - oo[lo]).astype(np.uint64))

    @property
    def n_samples(self) -> int:
        return int(self.sig_off[-1]) if not self.on_device else int(self.sig_off[-1].item())


@dataclass
class Result:
    counts: np.ndarray
    ev_off: np.ndarray
    ev_len: np.ndarray
    ev_read: np.ndarray
    samp_off: np.ndarray
    samples: np.ndarray
    read_skipped: np.ndarray
    n_reads: int

    def slot_values(self, s: int) -> np.ndarray:
        a, b = int(self.ev_off[s]), int(self.ev_off[s + 1])
        return self.samples[int(self.samp_off[a]):int(self.samp_off[b])]

    def slot_text(self, s: int, delimit: bool = False, sample_limit: Optional[int] = None) -> str:
        """dump/<KMER> content (src/gmove.cpp:941-944, 196-203) -- test helper; the CLI formats in C++."""
        out = []
        a, b = int(self.ev_off[s]), int(self.ev_off[s + 1])
        closed_at = None
        if sample_limit is not None and b - a == sample_limit and sample_limit > 0:
            closed_at = int(self.ev_read[b - 1])
        e = a
        for r in range(self.n_reads if delimit else 0):
            while e < b and self.ev_read[e] == r:
                out.append(self._ev_text(e)); e += 1
            if not self.read_skipped[r] and (closed_at is None or r < closed_at):
                out.append(":")
        if not delimit:
            out = [self._ev_text(i) for i in range(a, b)]
        return "".join(out)

    def _ev_text(self, e: int) -> str:
        v = self.samples[int(self.samp_off[e]):int(self.samp_off[e + 1])]
        return ",".join("%.8f" % x for x in v) + ";"


@dataclass
class Model:
    """pg_model_result as numpy arrays (one entry per slot) plus datamash-formatted texts."""
    n_values: np.ndarray
    median: np.ndarray
    sstdev: np.ndarray
    mid_lo: np.ndarray
    mid_hi: np.ndarray
    origin: np.ndarray
    sum1: np.ndarray
    sum2_lo: np.ndarray
    sum2_hi: np.ndarray
    dwell_n: np.ndarray
    dwell_median: np.ndarray
    median_text: List[str]
    sstdev_text: List[str]
    dwell_text: List[str]

    def raw_model_lines(self, kmers: List[str], limit: str = "3.1") -> str:
        """The file calculate_mean_stddev_all writes (scripts/poregen.sh:54-85): sorted by name, stddev capped."""
        out = []
        for i in sorted(range(len(kmers)), key=lambda j: kmers[j]):
            sd = self.sstdev_text[i]
            if sd not in ("", "nan") and float(sd) > float(limit):
                sd = limit
            out.append(f"{kmers[i]}\t{self.median_text[i]}\t{sd}\n")
        return "".join(out)

    def dwell_lines(self, kmers: List[str]) -> str:
        return "".join(f"{kmers[i]}\t{self.dwell_text[i]}\n" for i in sorted(range(len(kmers)), key=lambda j: kmers[j]))


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    return a.data_ptr()  # torch tensor


def _fill_params(owner, lib, params: "GmoveParams"):
    """pg_params for `params`; the code -> slot tables are kept alive on `owner`."""
    n_slots = len(params.kmers)
    k = params.kmer_size
    n_codes = 4 ** k
    owner._table_t = np.empty(n_codes, dtype=np.int32)
    owner._table_u = np.empty(n_codes, dtype=np.int32)
    arr = (C.c_char_p * n_slots)(*[s.encode() for s in params.kmers])
    st = lib.pg_build_slot_tables(k, arr, n_slots, owner._table_t.ctypes.data, owner._table_u.ctypes.data)
    if st != 0:
        raise PgError(st, lib.pg_last_error(None).decode())
    p = _abi.PgParams()
    lib.pg_default_params(C.byref(p))
    p.kmer_size = k; p.sig_move_offset = params.sig_move_offset; p.signal_print_margin = params.margin
    p.sample_limit = params.sample_limit; p.max_dur = params.max_dur; p.min_dur = params.min_dur
    p.kmer_pick_margin = params.kmer_pick_margin; p.scaling = params.scaling; p.allow_rna = int(params.rna)
    p.pa_min = params.pa_min; p.pa_max = params.pa_max; p.n_slots = n_slots
    p.flags = ((_abi.PG_FLAG_LAZY_STATS if params.lazy_stats else 0) | (_abi.PG_FLAG_PROFILE if params.profile else 0)
               | (_abi.PG_FLAG_OVERLAP if params.overlap else (_abi.PG_FLAG_ONE_STREAM if params.overlap is False else 0)) | (_abi.PG_FLAG_DEBUG_NARROW if params.debug_narrow else 0)
               | (_abi.PG_FLAG_STOP_WHEN_FULL if params.stop_when_full else 0) | (_abi.PG_FLAG_DEFER_STATS if params.defer_stats else 0)
               | (_abi.PG_FLAG_DEBUG_SPLIT_WALK if params.split_walk else 0)
               | (_abi.PG_FLAG_OVERLAP_TAIL if params.overlap_tail else 0))
    p.device = params.device
    p.table_t = owner._table_t.ctypes.data; p.table_u = owner._table_u.ctypes.data
    return p


def _result_from(r) -> "Result":
    def arr(ptr, n, dt):
        if n == 0 or not ptr:
            return np.zeros(0, dtype=dt)
        buf = (C.c_char * (n * np.dtype(dt).itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dt).copy()
    ns, ne, nsmp, nr = r.n_slots, r.n_events, r.n_samples, r.n_reads
    return Result(counts=arr(r.counts, ns, np.uint64), ev_off=arr(r.ev_off, ns + 1, np.uint64),
                  ev_len=arr(r.ev_len, ne, np.uint32), ev_read=arr(r.ev_read, ne, np.uint32),
                  samp_off=arr(r.samp_off, ne + 1, np.uint64), samples=arr(r.samples, nsmp, np.float64),
                  read_skipped=arr(r.read_skipped, nr, np.uint8), n_reads=int(nr))


def _text_slots(owner, t, fetch):
    off = [int(t.slot_off[i]) for i in range(t.n_slots + 1)]
    buf = C.create_string_buffer(int(t.n_bytes) + 1)
    owner._check(fetch(owner._h, 0, int(t.n_bytes), buf))
    raw = buf.raw
    return [raw[off[i]:off[i + 1]] for i in range(t.n_slots)]


def _deferred_result(owner, r, fetch, piece) -> "Result":
    res = _result_from(r)
    if r.n_samples and not r.samples:  # still on the device: range by range
        smp = np.empty(int(r.n_samples), dtype=np.float64)
        for a in range(0, int(r.n_samples), piece):
            n = min(piece, int(r.n_samples) - a)
            owner._check(fetch(owner._h, a, n, smp[a:a + n].ctypes.data_as(C.c_void_p)))
        res.samples = smp
    return res


def _c_batch(b: "Batch"):
    cb = _abi.PgBatch()
    cb.struct_size = C.sizeof(_abi.PgBatch)
    cb.location = _abi.PG_LOC_DEVICE if b.on_device else _abi.PG_LOC_HOST
    cb.n_reads = b.n_reads
    cb.n_ops = b.n_ops if b.on_device else 0
    cb.flags = (_abi.PG_BATCH_ALL_MATCHES if b.all_matches else 0) | (_abi.PG_BATCH_RESIDENT if (b.resident and b.on_device) else 0)
    for name, _ in _BATCH_FIELDS:
        setattr(cb, name, _ptr(getattr(b, name)))
    return cb


class GmoveEngine:
    """One gmove run on one GPU: the state the reference keeps in gmove() + process_move_table_paf()."""

    def __init__(self, params: GmoveParams):
        self._lib = _abi.load()
        self.params = params
        self.n_slots = len(params.kmers)
        p = _fill_params(self, self._lib, params)
        h = C.c_void_p()
        st = self._lib.pg_create(C.byref(p), C.byref(h))
        if st != 0:
            raise PgError(st, self._lib.pg_last_error(None).decode())
        self._h = h
        self._keep = None  # keeps the arrays of the batch between count and collect alive

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st != 0:
            raise PgError(st, self._lib.pg_last_error(self._h).decode())

    def _c_batch(self, b: Batch):
        return _c_batch(b)

    def submit(self, b: Batch):
        self._keep = b
        self._check(self._lib.pg_submit(self._h, C.byref(self._c_batch(b))))

    def count(self, b: Batch, out=None):
        """Phase 1. Returns the per-slot accepted-event counts of this batch (uncapped): a numpy uint64
        array, or fills `out` (a torch int64 CUDA tensor) in place when given."""
        self._keep = b
        if out is None:
            res = np.empty(self.n_slots, dtype=np.uint64)
            self._check(self._lib.pg_count(self._h, C.byref(self._c_batch(b)), res.ctypes.data, _abi.PG_LOC_HOST))
            return res
        self._check(self._lib.pg_count(self._h, C.byref(self._c_batch(b)), out.data_ptr(), _abi.PG_LOC_DEVICE))
        return out

    def stats(self):
        """Between count() and collect() of an engine made with defer_stats: queue the per-read statistics now (pg_stats)."""
        self._check(self._lib.pg_stats(self._h))

    def collect(self, base=None):
        """Phase 2. base: per-slot count of accepted events that precede this batch (numpy uint64 or a
        torch int64 CUDA tensor); None = this engine's own running count."""
        if base is None:
            self._check(self._lib.pg_collect(self._h, None, _abi.PG_LOC_HOST))
        elif isinstance(base, np.ndarray):
            base = np.ascontiguousarray(base, dtype=np.uint64)
            self._check(self._lib.pg_collect(self._h, base.ctypes.data, _abi.PG_LOC_HOST))
        else:
            self._check(self._lib.pg_collect(self._h, base.data_ptr(), _abi.PG_LOC_DEVICE))

    def collect_gathered(self, all_counts, world: int, rank: int):
        """Phase 2 of a multi-GPU job: all_counts = the all_gather's receive buffer (torch int64 CUDA tensor,
        world x n_slots, row g = rank g's count()); the library sums the rows below `rank` on its own stream."""
        if all_counts.numel() != world * self.n_slots or not all_counts.is_contiguous():
            raise ValueError("all_counts must be a contiguous world x n_slots tensor")
        self._check(self._lib.pg_collect_gathered(self._h, all_counts.data_ptr(), world, rank))

    def job_totals(self, device=None):
        """After collect_gathered: (accepted events of all ranks per slot, freq.txt values of the job) as int64 CUDA tensors that
        alias the library's buffers (pg_job_totals_device); rewritten by every collect_gathered on the engine's stream."""
        import torch
        if getattr(self, "_job_totals", None) is None:
            a, b = C.c_void_p(), C.c_void_p()
            self._check(self._lib.pg_job_totals_device(self._h, C.byref(a), C.byref(b)))
            dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device

            class _Alias:
                def __init__(self, ptr, n):
                    self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<i8", "data": (int(ptr), False), "version": 2}
            self._job_totals = tuple(torch.as_tensor(_Alias(x.value, self.n_slots), device=dev) for x in (a, b))
        return self._job_totals

    def sync(self):
        self._check(self._lib.pg_sync(self._h))

    def use_torch_stream(self, stream=None):
        """Run the main chain on a torch CUDA stream (default: the current one) so that torch collectives
        between count() and collect() are ordered on the device, without host synchronisation."""
        import torch
        s = torch.cuda.current_stream() if stream is None else stream
        if not s.cuda_stream:
            raise ValueError("use a non-default torch stream (torch.cuda.Stream() + torch.cuda.set_stream): handle 0 means "
                             "'the context's own stream' in pg_set_stream")
        self._check(self._lib.pg_set_stream(self._h, C.c_void_p(s.cuda_stream)))

    def reset(self):
        self._check(self._lib.pg_reset(self._h))

    def all_slots_full(self) -> bool:
        return bool(self._lib.pg_all_slots_full(self._h))

    def finish(self) -> Result:
        r = _abi.PgResult()
        self._check(self._lib.pg_finish(self._h, C.byref(r)))
        return _result_from(r)

    def model(self, keep_first: bool = False) -> "Model":
        """Per-k-mer median / sample stddev / dwell median of everything collected so far, reduced on the device from
        the kept samples (pg_model): the values `scripts/poregen.sh:54-85,33-52` derive from the dump files with
        tr | tail | datamash. `text(slot, which)` gives the number exactly as datamash prints it."""
        m = _abi.PgModelResult()
        self._check(self._lib.pg_model(self._h, _abi.PG_MODEL_KEEP_FIRST if keep_first else 0, C.byref(m)))
        return self._model_from(m)

    def model_device(self, counts, ev_len, samples, keep_first: bool = False) -> "Model":
        """The same reduction over torch CUDA tensors in the layout `dist.gather_kept` returns on the writing rank: counts
        int64[n_slots] kept events per k-mer, ev_len int32/uint32[n_events] (k-mer-major), samples float64[] back to back
        (pg_model_device)."""
        import torch
        ev_off = torch.zeros(counts.numel() + 1, dtype=torch.int64, device=counts.device)
        ev_off[1:] = torch.cumsum(counts.to(torch.int64), 0)
        samp_off = torch.zeros(ev_len.numel() + 1, dtype=torch.int64, device=counts.device)
        samp_off[1:] = torch.cumsum(ev_len.to(torch.int64), 0)
        ev_len = ev_len.contiguous(); samples = samples.contiguous()
        assert ev_len.element_size() == 4 and samples.dtype == torch.float64 and int(samp_off[-1]) == samples.numel()
        torch.cuda.current_stream(counts.device).synchronize()   # the library launches on its own stream
        m = _abi.PgModelResult()
        self._check(self._lib.pg_model_device(self._h, counts.numel(), ev_off.data_ptr(), samp_off.data_ptr(), ev_len.data_ptr() or None,
                                              samples.data_ptr() or None, _abi.PG_MODEL_KEEP_FIRST if keep_first else 0, C.byref(m)))
        return self._model_from(m)

    def _model_from(self, m) -> "Model":
        ns = m.n_slots

        def arr(ptr, dt):
            if ns == 0 or not ptr:
                return np.zeros(0, dtype=dt)
            return np.frombuffer((C.c_char * (ns * np.dtype(dt).itemsize)).from_address(ptr), dtype=dt).copy()
        buf = C.create_string_buffer(64)
        texts = []
        for which in (_abi.PG_MODEL_TEXT_MEDIAN, _abi.PG_MODEL_TEXT_SSTDEV, _abi.PG_MODEL_TEXT_DWELL):
            col = []
            for s in range(ns):
                n = self._lib.pg_model_format(C.byref(m), s, which, buf, 64)
                col.append(buf.raw[:n].decode())
            texts.append(col)
        return Model(n_values=arr(m.n_values, np.uint64), median=arr(m.median, np.float64), sstdev=arr(m.sstdev, np.float64),
                     mid_lo=arr(m.mid_lo, np.int64), mid_hi=arr(m.mid_hi, np.int64), origin=arr(m.origin, np.int64),
                     sum1=arr(m.sum1, np.int64), sum2_lo=arr(m.sum2_lo, np.uint64), sum2_hi=arr(m.sum2_hi, np.uint64),
                     dwell_n=arr(m.dwell_n, np.uint64), dwell_median=arr(m.dwell_median, np.float64),
                     median_text=texts[0], sstdev_text=texts[1], dwell_text=texts[2])

    def text(self):
        """The dump files' text produced on the device (pg_text): list of bytes objects, one per slot -- what the reference's fprintf calls
        write into dump/<KMER> without -d."""
        t = _abi.PgTextResult()
        self._check(self._lib.pg_text(self._h, C.byref(t)))
        return _text_slots(self, t, self._lib.pg_fetch_text)

    def finish_deferred(self, piece: int = 1 << 20) -> Result:
        """pg_finish_deferred + pg_fetch_samples: the same Result as finish(), the samples fetched from the device `piece` at a time."""
        r = _abi.PgResult()
        self._check(self._lib.pg_finish_deferred(self._h, C.byref(r)))
        return _deferred_result(self, r, self._lib.pg_fetch_samples, piece)

    def device_view(self):
        v = _abi.PgDeviceView()
        self._check(self._lib.pg_last_batch_device(self._h, C.byref(v)))
        return v

    def kept_tensors(self, device=None, with_reads=False):
        """The last collected batch as torch CUDA tensors that alias the library's device buffers (valid until the next
        submit/collect/reset): (kept events per slot int64[n_slots], window lengths int32[n_events] in slot-major
        order, samples float64[n_samples]; with_reads: also the kept events' read indices inside the batch, int32[n_events]). What
        dist.gather_kept sends to the writing rank."""
        import torch
        v = self.device_view()

        class _Alias:  # zero-copy hand-over through the CUDA array interface
            def __init__(self, ptr, n, typestr):
                self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr or 0), False), "version": 2}

        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device

        def wrap(ptr, n, typestr, dtype):
            if n == 0 or not ptr:
                return torch.empty(0, dtype=dtype, device=dev)
            return torch.as_tensor(_Alias(ptr, n, typestr), device=dev)

        counts = wrap(v.d_keep, self.n_slots, "<i8", torch.int64)
        ev_len = wrap(v.d_ev_len, v.n_events, "<i4", torch.int32)
        samples = wrap(v.d_samples, v.n_samples, "<f8", torch.float64)
        if with_reads:
            return counts, ev_len, samples, wrap(v.d_ev_read, v.n_events, "<i4", torch.int32)
        return counts, ev_len, samples

    def kernel_stats(self):
        n = C.c_uint32(0)
        buf = (_abi.PgKernelStat * 64)()
        self._check(self._lib.pg_kernel_stats(self._h, buf, 64, C.byref(n)))
        return {buf[i].name.decode(): (int(buf[i].launches), float(buf[i].total_ms)) for i in range(min(n.value, 64))}

    def kernel_stats_reset(self):
        self._check(self._lib.pg_kernel_stats_reset(self._h))


class GmoveJob:
    """One gmove job over several GPUs from THIS process (pg_job_*): the batch is cut into contiguous shards, one per listed
    device, with one exchange of per-k-mer counts per batch (RCCL all-gather when the devices are distinct, host memory
    otherwise). Same results as a GmoveEngine fed the same batches."""

    def __init__(self, params: GmoveParams, devices: Sequence[int], exchange: int = _abi.PG_JOB_EXCHANGE_AUTO):
        self._lib = _abi.load()
        self.params = params
        self.n_slots = len(params.kmers)
        p = _fill_params(self, self._lib, params)
        devs = (C.c_int32 * len(devices))(*devices)
        h = C.c_void_p()
        st = self._lib.pg_job_create(C.byref(p), devs, len(devices), exchange, C.byref(h))
        if st != 0:
            raise PgError(st, self._lib.pg_job_last_error(None).decode())
        self._h = h
        self._keep = None

    def _check(self, st):
        if st != 0:
            raise PgError(st, self._lib.pg_job_last_error(self._h).decode())

    @property
    def uses_rccl(self) -> bool:
        return bool(self._lib.pg_job_uses_rccl(self._h))

    def submit(self, b: Batch):
        self._keep = b
        self._check(self._lib.pg_job_submit(self._h, C.byref(_c_batch(b))))

    def submit_shards(self, shards: Sequence[Batch]):
        """pg_job_submit_shards: one batch per device of the job, in PAF order, each where it will be worked (a device batch resident
        on that device, or a host batch): nothing is cut or copied on the host."""
        self._keep = list(shards)
        arr = (_abi.PgBatch * len(shards))(*[_c_batch(b) for b in shards])
        self._check(self._lib.pg_job_submit_shards(self._h, arr, len(shards)))

    def reset(self):
        self._check(self._lib.pg_job_reset(self._h))

    def sync(self):
        self._check(self._lib.pg_job_sync(self._h))

    def all_slots_full(self) -> bool:
        return bool(self._lib.pg_job_all_slots_full(self._h))

    def finish(self) -> Result:
        r = _abi.PgResult()
        self._check(self._lib.pg_job_finish(self._h, C.byref(r)))
        return _result_from(r)

    def finish_deferred(self, piece: int = 1 << 20) -> Result:
        """pg_job_finish_deferred + pg_job_fetch_samples: the shards' samples concatenated on the job's first device, fetched in pieces."""
        r = _abi.PgResult()
        self._check(self._lib.pg_job_finish_deferred(self._h, C.byref(r)))
        return _deferred_result(self, r, self._lib.pg_job_fetch_samples, piece)

    def text(self):
        """pg_job_text: the dump files' text of the whole job, produced on its first device; one bytes object per slot."""
        t = _abi.PgTextResult()
        self._check(self._lib.pg_job_text(self._h, C.byref(t)))
        return _text_slots(self, t, self._lib.pg_job_fetch_text)

    def model(self, keep_first: bool = False) -> "Model":
        m = _abi.PgModelResult()
        self._check(self._lib.pg_job_model(self._h, _abi.PG_MODEL_KEEP_FIRST if keep_first else 0, C.byref(m)))
        return GmoveEngine._model_from(self, m)

    def kernel_stats(self, shard: int):
        """pg_kernel_stats of one shard's context (params.profile): {kernel: (launches, total ms)}."""
        n = C.c_uint32(0)
        buf = (_abi.PgKernelStat * 64)()
        self._check(self._lib.pg_job_kernel_stats(self._h, shard, buf, 64, C.byref(n)))
        return {buf[i].name.decode(): (int(buf[i].launches), float(buf[i].total_ms)) for i in range(min(n.value, 64))}

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pg_job_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
