"""How long does the host take to enqueue one step (reset + submit) vs how long the GPU takes to run it?"""
import sys, time; sys.path.insert(0, '.')
import torch
from poregen_amd import synth
from poregen_amd.engine import GmoveEngine, GmoveParams, generate_kmers
b = synth.make_batch_fast(50000, kind="rna004", seed=20251004)
d = b.to_device(torch.device("cuda:0"))
for kw in ({}, {"overlap": True}, {"scaling": 0}):
    p = dict(kmers=generate_kmers(5, True), kmer_size=5, rna=True, scaling=1, min_dur=20, max_dur=40, sample_limit=100); p.update(kw)
    e = GmoveEngine(GmoveParams(**p))
    for _ in range(5): e.reset(); e.submit(d)
    e.sync()
    K = 50
    t0 = time.perf_counter()
    for _ in range(K): e.reset(); e.submit(d)
    t1 = time.perf_counter(); e.sync(); t2 = time.perf_counter()
    print(kw, "enqueue us/step", (t1 - t0) / K * 1e6, "total us/step", (t2 - t0) / K * 1e6)
    e.close()
