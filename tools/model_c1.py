#!/usr/bin/env python3
"""k_slot_model on BASELINE configs[1] (k = 5, 1024 k-mers, sample_limit 100 or argv[1]): ms per pg_model launch sequence (profile-mode bracket).
usage: python3 tools/model_c1.py [sample_limit] [--lib build/x/libpgmove.so]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "--lib" in sys.argv:
    from poregen_amd import _abi
    _abi.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
LIMIT = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 100
import torch
from poregen_amd import synth
from poregen_amd.engine import GmoveEngine, GmoveParams, generate_kmers
dev = torch.device("cuda", 0)
host = synth.make_batch_fast(50000, read_len=4000, kind="rna004", seed=20251003 + 1)
shard = host.to_device(dev)
kmers = generate_kmers(5, rna=True)
e = GmoveEngine(GmoveParams(kmers=kmers, kmer_size=5, rna=True, scaling=1, sample_limit=LIMIT, min_dur=20, max_dur=40, profile=True))
e.submit(shard); e.sync()
for _ in range(3):
    m = e.model()
e.kernel_stats_reset()
for _ in range(10):
    m = e.model()
ks = e.kernel_stats()["k_slot_model"]
print("k_slot_model %.1f us per call (%d calls); values per file: median %d max %d" % (ks[1] / ks[0] * 1e3, ks[0], int(sorted(m.n_values)[len(m.n_values) // 2]), int(max(m.n_values))))
e.close()
