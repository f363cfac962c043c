"""How much does the chip gain when two INDEPENDENT jobs (two contexts, own streams) run side by side? An upper bound for what deeper
pipelining of ONE job's batches (gather of batch i next to the chain of batch i + 1) could win. usage: python3 tools/probe/two_jobs.py [k9|l5000|c1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from poregen_amd import synth
from poregen_amd.engine import GmoveEngine, GmoveParams, generate_kmers
mode = sys.argv[1] if len(sys.argv) > 1 else "k9"
dev = torch.device("cuda", 0)
if mode == "k9":
    kind, p = "dna_r10", dict(kmer_size=9, rna=False, scaling=1, sample_limit=1000)
else:
    kind, p = "rna004", dict(kmer_size=5, rna=True, scaling=1, min_dur=20, max_dur=40, sample_limit=5000 if mode == "l5000" else 100)
kmers = generate_kmers(p["kmer_size"], rna=p["rna"])
shards = [synth.make_batch_fast(50000, kind=kind, seed=77 + i).to_device(dev) for i in range(2)]
def run(engs, steps):
    for e, s in zip(engs, shards): e.reset(); e.submit(s)
    for e in engs: e.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        for e, s in zip(engs, shards): e.reset(); e.submit(s)
    for e in engs: e.sync()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
for ov in (None, False):
    one = [GmoveEngine(GmoveParams(kmers=kmers, overlap=ov, **p))]
    t1 = run(one, 10)
    two = one + [GmoveEngine(GmoveParams(kmers=kmers, overlap=ov, **p))]
    t2 = run(two, 10)
    print(f"{mode} overlap={ov}: one job {t1:.4f} ms per batch; two jobs side by side {t2:.4f} ms per pair = {t2 / 2:.4f} ms per batch ({t1 / (t2 / 2):.2f}x)")
    for e in two: e.close()
