"""Does bench.py's step gain from resubmitting the SAME device batch (its op / sequence arrays may still sit in the Infinity Cache
from the previous step)? One batch against N different batches of the same shape in rotation. usage (GPU box): python3 tools/probe/rotate_batches.py"""
import sys, time; sys.path.insert(0, '.')
import torch
from poregen_amd import synth
from poregen_amd.engine import GmoveEngine, GmoveParams, generate_kmers
dev = torch.device("cuda:0")
shards = [synth.make_batch_fast(50000, kind="rna004", seed=20251004 + i).to_device(dev) for i in range(6)]
p = dict(kmers=generate_kmers(5, True), kmer_size=5, rna=True, scaling=1, min_dur=20, max_dur=40, sample_limit=100)
for two in (False, True):
    e = GmoveEngine(GmoveParams(overlap=two, **p))
    for n in (1, 2, 3, 6, 1):
        for i in range(10): e.reset(); e.submit(shards[i % n])
        e.sync(); torch.cuda.synchronize()
        K = 60
        t0 = time.perf_counter()
        for i in range(K): e.reset(); e.submit(shards[i % n])
        e.sync(); torch.cuda.synchronize()
        print("two-stream" if two else "one stream", "batches in rotation:", n, " ms per step %.4f" % ((time.perf_counter() - t0) / K * 1e3))
    e.close()
