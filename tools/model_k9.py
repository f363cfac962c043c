#!/usr/bin/env python3
"""k_slot_model on BASELINE configs[3] (k = 9, 262 144 k-mers, sample_limit 1000): ms per launch and its share of the HBM roofline.
usage: python tools/model_k9.py [--lib build/x/libpgmove.so]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "--lib" in sys.argv:
    from poregen_amd import _abi
    _abi.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
import torch
from poregen_amd import synth
from poregen_amd.engine import GmoveEngine, GmoveParams, generate_kmers
dev = torch.device("cuda", 0)
host = synth.make_batch_fast(50000, read_len=4000, kind="dna_r10", seed=20251003 + 3, homopolymer_frac=0.1)
shard = host.to_device(dev)
kmers = generate_kmers(9, rna=False)
e = GmoveEngine(GmoveParams(kmers=kmers, kmer_size=9, rna=False, scaling=1, sample_limit=1000, profile=True))
e.submit(shard); e.sync()
for _ in range(2):
    m = e.model()
e.kernel_stats_reset()
for _ in range(5):
    m = e.model()
ks = e.kernel_stats()["k_slot_model"]
v = e.device_view()
nbytes = 8 * int(v.n_samples) + 4 * int(v.n_events)
ms = ks[1] / ks[0]
print("k_slot_model %.3f ms per call (%d launches), %.2f GB -> %.0f GB/s = %.3f of peak" % (ms, ks[0], nbytes / 1e9, nbytes / (ms * 1e-3) / 1e9, nbytes / (ms * 1e-3) / 1e9 / 8000.0))
e.close()
