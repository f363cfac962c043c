"""Per-wave phase times of k_read_stats and k_walk on bench.py's workload, from the measurement build
(make variant NAME=phase EXTRA=-DPG_PHASE_PROBE). usage (GPU box): python3 tools/phase_probe.py [reads] [sample_limit]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from poregen_amd import _abi
_abi.LIB_PATH = os.path.join(ROOT, "build", "phase", "libpgmove.so")
import torch
from poregen_amd import synth
from poregen_amd.engine import GmoveEngine, GmoveParams, generate_kmers

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 100
kind = sys.argv[3] if len(sys.argv) > 3 else "rna004"   # e.g. `50000 1000 dna_r10 9` = BASELINE configs[3]
k = int(sys.argv[4]) if len(sys.argv) > 4 else 5
rna = kind == "rna004"
host = synth.make_batch_fast(reads, read_len=4000, kind=kind, seed=20251003 + 1, homopolymer_frac=0.0 if rna else 0.1)
p = dict(kmer_size=k, rna=rna, scaling=1, sample_limit=limit)
if rna: p.update(min_dur=20, max_dur=40)
eng = GmoveEngine(GmoveParams(kmers=generate_kmers(k, rna=rna), **p))
lib = eng._lib
import numpy as np
W = 65536
buf = np.zeros((W, 8), np.uint64)
lib.pg_debug_phases.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]; lib.pg_debug_phases.restype = None
shard = host.to_device(torch.device("cuda:0"))
for _ in range(3):
    eng.reset(); eng.submit(shard)
eng.sync()
lib.pg_debug_phases(buf.ctypes.data, 0, 1)
eng.reset(); eng.submit(shard); eng.sync()
names = {0: ("k_read_stats", ["record load", "samples arrive", "binning", "prefix scan", "selection + stores"]),
         1: ("k_events<true> (per wave, 4 tiles)", ["first reads of the tiles + op_n", "table entry + block sums", "the two barriers", "reads of the group + base codes",
                                                   "stores (straight-line form) / counts + stores (partitioned) / the whole of the events (other forms)", "masks (straight-line) / tests + table indices (partitioned)", "look-ups + counts (straight-line) / look-ups (partitioned)"])}
# k_events: the last barrier and the histogram rows are the rest of the lifetime
names[2] = ("k_rank_emit (waves of the tiles that place events)", ["last useful tile known", "keys + column + keep + offsets arrive", "any-room test, counts, wave bases",
                                                                    "ordered rows (+ read records)", "windows", "stores"])
for k, (name, ph) in names.items():
    lib.pg_debug_phases(buf.ctypes.data, k, 0)
    live = buf[:, 7] > 0
    a = buf[live].astype(np.float64)
    if not a.size: continue
    print(f"{name}: {int(live.sum())} waves recorded; wave lifetime mean {a[:, 7].mean():.0f} ticks (median {np.median(a[:, 7]):.0f}, p90 {np.percentile(a[:, 7], 90):.0f}) of s_memtime")
    for i, nm in enumerate(ph):
        print(f"   {nm:24s} mean {a[:, i].mean():9.0f} ticks  {100.0 * a[:, i].sum() / a[:, 7].sum():5.1f} % of the lifetime   median {np.median(a[:, i]):7.0f}")
