"""Does the back-to-back long-read test catch round 5's helper-counter race? Runs tests/test_gpu_edges.py's scenario against a given library build.
usage (GPU box): python3 tools/probe/r06_ring_race_check.py build/oldring/libpgmove.so [repeats]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
from poregen_amd import _abi
_abi.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np, torch
from helpers import assert_result_equals_oracle, oracle_for
from poregen_amd import synth
from poregen_amd.engine import GmoveEngine, GmoveParams, generate_kmers
L = np.array([3000, 70_000, 5000, 300_000, 40_001, 4000, 120_000, 33_000, 2500, 65_537] * 3, np.int64)
b = synth.make_ragged_fast(L, kind="dna_r10", seed=77)
kmers = generate_kmers(5); p = dict(kmer_size=5, scaling=1, sample_limit=2000)
o = oracle_for(kmers, **p); o.run_batch(b)
d = b.to_device(torch.device("cuda", 0))
bad = 0
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10):
    eng = GmoveEngine(GmoveParams(kmers=kmers, **p))
    for _ in range(40):
        eng.reset(); eng.submit(d)
    res = eng.finish(); st = eng.kernel_stats(); eng.close()
    try:
        assert_result_equals_oracle(res, o, sample_limit=2000)
    except AssertionError as e:
        bad += 1; print("rep", rep, "DIFFERS:", str(e)[:120])
print(sys.argv[1], ":", bad, "of the repetitions differ from the oracle")
