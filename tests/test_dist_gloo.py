"""The N>1 path on CPU: two `gloo` ranks run poregen_amd.dist.sharded_step on contiguous PAF-order shards.
The GPU engine is replaced by a stand-in with the same count/collect contract built on the CPU oracle (this
is a test of the exchange logic, not of the kernels), and the rank-ordered concatenation must equal a single
sequential oracle run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as tdist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleEngine:
    """count(): accepted events per slot of the shard, uncapped. collect(base): keep the events whose global
    rank base+local is < sample_limit. Same contract as GmoveEngine.count/collect (include/pgmove.h)."""

    def __init__(self, kmers, p):
        from helpers import oracle_for
        self.kmers, self.p = kmers, p
        self.n_slots = len(kmers)
        big = dict(p); big["sample_limit"] = 2 ** 31 - 1
        self.o = oracle_for(kmers, **big)

    def count(self, shard):
        self.o.run_batch(shard)
        self.calls = ["count"]
        return self.o.counts().astype(np.uint64)

    def stats(self):  # GmoveEngine.stats (pg_stats): placed by sharded_step behind the exchange's issue, in front of collect
        self.calls.append("stats")

    def collect(self, base):
        assert self.calls == ["count", "stats"], self.calls
        self.calls.append("collect")
        lim = self.p["sample_limit"]
        self.kept, self.kept_lens = [], []
        for s in range(self.n_slots):
            room = max(0, lim - int(base[s]))
            lens = self.o.event_lens(s)[:room]
            self.kept_lens.append(lens.astype(np.int32))
            self.kept.append(self.o.values(s)[: int(lens.sum())])

    def sync(self):
        pass


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
    from poregen_amd import dist as pgdist
    from poregen_amd import synth
    from poregen_amd.engine import generate_kmers
    tdist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    b = synth.make_batch(120, kind="rna004", seed=21, indel_rate=0.02)
    p = dict(kmer_size=5, rna=True, scaling=1, min_dur=20, max_dur=40, sample_limit=9)
    kmers = generate_kmers(5, rna=True)
    lo, hi = pgdist.shard_bounds(b.n_reads, world, rank)
    eng = OracleEngine(kmers, p)
    total = pgdist.sharded_step(eng, b.slice_reads(lo, hi))
    freq = pgdist.merged_freq(total, p["sample_limit"]).numpy()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), freq=freq, sizes=np.array([v.size for v in eng.kept]),
             flat=np.concatenate(eng.kept) if eng.kept else np.zeros(0))
    # the single-writer end: every rank's kept events to rank 0, slot-major / rank order inside a slot
    g = pgdist.gather_kept(torch.tensor([len(x) for x in eng.kept_lens], dtype=torch.int64), torch.from_numpy(np.concatenate(eng.kept_lens)),
                           torch.from_numpy(np.concatenate(eng.kept)))
    assert (g is None) == (rank != 0)
    if rank == 0:
        np.savez(os.path.join(out_dir, "gathered.npz"), counts=g[0].numpy(), ev_len=g[1].numpy(), samples=g[2].numpy())
    tdist.barrier()
    tdist.destroy_process_group()


def test_two_rank_gloo_equals_sequential_oracle(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    from helpers import oracle_for
    from poregen_amd import synth
    from poregen_amd.engine import generate_kmers
    b = synth.make_batch(120, kind="rna004", seed=21, indel_rate=0.02)
    p = dict(kmer_size=5, rna=True, scaling=1, min_dur=20, max_dur=40, sample_limit=9)
    kmers = generate_kmers(5, rna=True)
    o = oracle_for(kmers, **p)
    o.run_batch(b)
    r = [np.load(tmp_path / f"rank{i}.npz") for i in range(2)]
    assert np.array_equal(r[0]["freq"], r[1]["freq"])
    assert np.array_equal(r[0]["freq"].astype(np.uint64), o.counts())
    offs = [np.concatenate([[0], np.cumsum(x["sizes"])]) for x in r]
    for s in range(len(kmers)):
        cat = np.concatenate([r[i]["flat"][offs[i][s]:offs[i][s + 1]] for i in range(2)])
        assert np.array_equal(cat.view(np.uint64), o.values(s).view(np.uint64)), s
    # dist.gather_kept on rank 0: the whole job as one writer would dump it
    g = np.load(tmp_path / "gathered.npz")
    assert np.array_equal(g["counts"].astype(np.uint64), o.counts())
    assert np.array_equal(g["ev_len"], np.concatenate([o.event_lens(s) for s in range(len(kmers))]).astype(np.int32))
    assert np.array_equal(g["samples"].view(np.uint64), np.concatenate([o.values(s) for s in range(len(kmers))]).view(np.uint64))


# ---- the same with the REAL engine: two gloo ranks share cuda:0 (what tools/rehearse_two_ranks.sh did by hand in round 3) -------------
def _gpu_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
    from poregen_amd import dist as pgdist
    from poregen_amd import synth
    from poregen_amd.engine import GmoveEngine, GmoveParams, generate_kmers
    tdist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    b = synth.make_batch(240, kind="rna004", seed=23, indel_rate=0.02)
    p = dict(kmer_size=5, rna=True, scaling=1, min_dur=20, max_dur=40, sample_limit=9)
    kmers = generate_kmers(5, rna=True)
    lo, hi = pgdist.shard_bounds(b.n_reads, world, rank)
    eng = GmoveEngine(GmoveParams(kmers=kmers, defer_stats=True, **p))  # the statistics placed by sharded_step, as bench.py --gpus N does
    shard = b.slice_reads(lo, hi).to_device(dev)
    total = pgdist.sharded_step(eng, shard)
    eng.sync()
    freq = pgdist.merged_freq(total, p["sample_limit"], engine=eng).cpu().numpy()
    counts, ev_len, samples = eng.kept_tensors(device=dev)
    sizes = torch.zeros(len(kmers), dtype=torch.int64)
    starts = torch.cumsum(counts, 0) - counts
    cs = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(ev_len.to(torch.int64), 0)])
    sizes = (cs[(starts + counts)] - cs[starts]).cpu().numpy()          # kept samples per slot of this rank
    np.savez(os.path.join(out_dir, f"grank{rank}.npz"), freq=freq, sizes=sizes, flat=samples.cpu().numpy())
    # the single-writer end on CPU tensors (gloo has no device point-to-point)
    g = pgdist.gather_kept(counts.cpu(), ev_len.cpu(), samples.cpu())
    assert (g is None) == (rank != 0)
    if rank == 0:
        np.savez(os.path.join(out_dir, "ggathered.npz"), counts=g[0].numpy(), ev_len=g[1].numpy(), samples=g[2].numpy())
    tdist.barrier()
    eng.close()
    tdist.destroy_process_group()


@pytest.mark.gpu
def test_two_process_gloo_real_engine(tmp_path):
    """Two PROCESSES, one rank each, both on cuda:0, through dist.sharded_step with the real GmoveEngine (libpgmove): pg_count ->
    exchange of u64[n_slots] (gloo) -> pg_stats -> pg_collect with the lower rank's base. The rank-ordered concatenation of the ranks'
    streams equals ONE sequential oracle run bit for bit, and so does dist.gather_kept on the writing rank."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_gpu_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    from helpers import oracle_for
    from poregen_amd import synth
    from poregen_amd.engine import generate_kmers
    b = synth.make_batch(240, kind="rna004", seed=23, indel_rate=0.02)
    p = dict(kmer_size=5, rna=True, scaling=1, min_dur=20, max_dur=40, sample_limit=9)
    kmers = generate_kmers(5, rna=True)
    o = oracle_for(kmers, **p)
    o.run_batch(b)
    r = [np.load(tmp_path / f"grank{i}.npz") for i in range(2)]
    assert np.array_equal(r[0]["freq"], r[1]["freq"])
    assert np.array_equal(r[0]["freq"].astype(np.uint64), o.counts())
    offs = [np.concatenate([[0], np.cumsum(x["sizes"])]) for x in r]
    for sl in range(len(kmers)):
        cat = np.concatenate([r[i]["flat"][offs[i][sl]:offs[i][sl + 1]] for i in range(2)])
        assert np.array_equal(cat.view(np.uint64), o.values(sl).view(np.uint64)), sl
    g = np.load(tmp_path / "ggathered.npz")
    assert np.array_equal(g["counts"].astype(np.uint64), o.counts())
    assert np.array_equal(g["ev_len"], np.concatenate([o.event_lens(sl) for sl in range(len(kmers))]).astype(np.int32))
    assert np.array_equal(g["samples"].view(np.uint64), np.concatenate([o.values(sl) for sl in range(len(kmers))]).view(np.uint64))
