"""Randomised parity run on the GPU box (not part of the test suite: a few minutes of random small jobs against the CPU
oracle). usage: python3 tools/fuzz_gpu.py [n_cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
from helpers import assert_result_equals_oracle, oracle_for
from poregen_amd import synth
from poregen_amd.engine import GmoveEngine, GmoveParams, generate_kmers
import pathlib, shutil, tempfile
from test_gpu_model import compare_raw_model, dump_from_oracle, oracle_lines

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
skipped = 0
n_model = 0
for case in range(n_cases):
    rna = bool(rng.integers(0, 2))
    k = int(rng.choice([3, 5, 5, 6, 9]))
    p = dict(kmer_size=k, rna=rna, scaling=int(rng.integers(0, 2)), sample_limit=int(rng.choice([1, 3, 20, 100, 1000])),
             kmer_pick_margin=int(rng.integers(0, 4)), sig_move_offset=int(rng.integers(0, k)),
             min_dur=int(rng.choice([1, 5, 20])), max_dur=int(rng.choice([30, 40, 70, 200])))
    if rng.random() < 0.25:
        p.update(margin=int(rng.integers(1, 4)))   # --margin: windows widened on both sides (start < margin is undefined in the reference)
    if rng.random() < 0.3:
        lo = float(rng.uniform(-60, 100)); p.update(pa_min=lo, pa_max=lo + float(rng.choice([30, 120, 250, 600])))
    n_reads = int(rng.choice([1, 7, 60, 300]))
    read_len = int(rng.choice([150, 1000, 4000, 9001]))
    if n_reads <= 60 and rng.random() < 0.15:
        read_len = int(rng.choice([33000, 40001, 70000]))   # above the split threshold of the statistics (PgLongState): several waves per read
    b = synth.make_batch(n_reads, read_len=read_len, kind="rna004" if rna else "dna_r10", seed=int(rng.integers(1 << 30)),
                         indel_rate=float(rng.choice([0.0, 0.02, 0.1])), spike_rate=float(rng.choice([0.0, 0.005, 0.2])))
    if rng.random() < 0.4:  # PAF column 3 (query_start) > 0: the walk starts inside the signal; the last match gives the samples back
        from poregen_amd.engine import Batch
        qs = b.query_start.copy(); opn = b.op_n.copy()
        for r in range(b.n_reads):
            last = int(b.op_off[r + 1]) - 1
            q = int(rng.integers(0, 60))
            if last >= int(b.op_off[r]) and b.op_t[last] == 0 and opn[last] > q:
                opn[last] -= q; qs[r] = q
        b = Batch(**{**b.__dict__, "query_start": qs, "op_n": opn})
    kmers = generate_kmers(k, rna=rna)
    sl = {}
    if rng.random() < 0.3 and k <= 6:  # a shuffled whitelist and a slice of it (--kmer_file, --index_start/--index_end)
        kmers = [kmers[i] for i in rng.permutation(len(kmers))[:int(rng.integers(1, len(kmers) + 1))]]
        a = int(rng.integers(1, len(kmers) + 1)); b2 = int(rng.integers(a, len(kmers) + 1))
        sl = dict(index_start=a, index_end=b2)
    o = oracle_for(kmers, **sl, **p)
    rcs = o.run_batch(b)
    if min(rcs) < 0:  # the oracle flags an input on which the reference has undefined behaviour
        skipped += 1
        continue
    slice_kmers = kmers[sl["index_start"] - 1:sl["index_end"]] if sl else kmers  # the engine takes the slice itself
    # the op-parallel event kernel or (forced) the wave-per-read walk for every read; host batches or device-resident ones (which carry
    # pg_batch.n_ops and, without indels, PG_BATCH_ALL_MATCHES: no k_walk launch at all)
    eng = GmoveEngine(GmoveParams(kmers=slice_kmers, lazy_stats=bool(rng.integers(0, 2)), debug_narrow=bool(rng.random() < 0.15),
                                  split_walk=bool(rng.random() < 0.2), **p))
    on_device = rng.random() < 0.4
    def put(x):
        if on_device:
            import torch
            return x.to_device(torch.device("cuda:0"))
        return x
    try:
        cut = int(rng.integers(0, n_reads + 1))
        if cut and cut < n_reads:
            eng.submit(put(b.slice_reads(0, cut))); eng.submit(put(b.slice_reads(cut, n_reads)))
        else:
            eng.submit(put(b))
        res = eng.finish()
        assert_result_equals_oracle(res, o, check_text_slots=2, sample_limit=p["sample_limit"])
        if len(slice_kmers) <= 1024:  # the k-mer model (pg_model) against tr | tail | datamash restated on the oracle's dump files
            tmp = pathlib.Path(tempfile.mkdtemp(prefix="pgfuzz"))
            try:
                d = dump_from_oracle(tmp, o, slice_kmers)
                m = eng.model()
                limit = str(rng.choice(["3.1", "0.5", "1e9"]))
                compare_raw_model(m.raw_model_lines(slice_kmers, limit), oracle_lines(d, "stats", limit), d, limit)
                assert m.dwell_lines(slice_kmers) == "".join(oracle_lines(d, "dwell"))
                n_model += 1
            finally:
                shutil.rmtree(tmp, ignore_errors=True)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("CASE", case, "FAILED:", p, n_reads, read_len, repr(e)[:300], flush=True)
    finally:
        eng.close()
    if case % 20 == 19:
        print("case", case + 1, "ok so far" if not bad else f"{bad} failures", flush=True)
print("fuzz done:", n_cases, "cases,", skipped, "outside the reference's defined behaviour (skipped),", n_model, "with the model step,", bad, "failures")
sys.exit(1 if bad else 0)
