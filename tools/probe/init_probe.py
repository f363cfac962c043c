import sys; sys.path.insert(0, '.')
import torch
from poregen_amd import synth
from poregen_amd.engine import GmoveEngine, GmoveParams, generate_kmers
b = synth.make_batch_fast(50000, kind="rna004", seed=20251004)
d = b.to_device(torch.device("cuda:0"))
for kw in ({}, {"lazy_stats": True}, {"scaling": 0}):
    p = dict(kmers=generate_kmers(5, True), kmer_size=5, rna=True, scaling=1, min_dur=20, max_dur=40, sample_limit=100, profile=True); p.update(kw)
    e = GmoveEngine(GmoveParams(**p))
    for _ in range(5): e.reset(); e.submit(d)
    e.sync(); e.kernel_stats_reset()
    for _ in range(20): e.reset(); e.submit(d)
    e.sync()
    print(kw, {k: round(v[1] / v[0] * 1e3, 1) for k, v in e.kernel_stats().items()})
    e.close()
