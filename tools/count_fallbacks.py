"""How many reads of bench.py's workload leave the statistics kernel's fast selection (stats_select_fast) for the general one?
Needs the measurement build of the library (-DPG_COUNT_FALLBACKS, `make fallback_probe` -> build/fb/libpgmove_fb.so).
usage: python3 tools/count_fallbacks.py [reads] [kind]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from poregen_amd import _abi
_abi.LIB_PATH = os.path.join(ROOT, "build", "fb", "libpgmove_fb.so")
import torch
from poregen_amd import synth
from poregen_amd.engine import GmoveEngine, GmoveParams, generate_kmers

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
kind = sys.argv[2] if len(sys.argv) > 2 else "rna004"
rna = kind == "rna004"
host = synth.make_batch_fast(reads, read_len=4000, kind=kind, seed=20251003 + 1)
p = dict(kmer_size=5, rna=rna, scaling=1, sample_limit=100)
if rna: p.update(min_dur=20, max_dur=40)
eng = GmoveEngine(GmoveParams(kmers=generate_kmers(5, rna=rna), **p))
lib = eng._lib
lib.pg_debug_fallbacks.restype = ctypes.c_ulonglong; lib.pg_debug_fallbacks.argtypes = [ctypes.c_int]
shard = host.to_device(torch.device("cuda:0"))
lib.pg_debug_fallbacks(1)
eng.submit(shard); eng.sync()
n = lib.pg_debug_fallbacks(1)
print(f"{kind}: {n} of {reads} reads took the general selection ({100.0 * n / reads:.2f} %)")
