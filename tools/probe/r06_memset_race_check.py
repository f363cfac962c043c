"""Does the overlap-mode long-histogram test catch round 5's unordered zero-fill (advisor r05, high)? The scenario of
tests/test_gpu_edges.py::test_long_histograms_are_zero_before_the_statistics_stream_uses_them against a given library build.
usage (GPU box): python3 tools/probe/r06_memset_race_check.py build/r05/libpgmove.so [repeats]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
from poregen_amd import _abi
_abi.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np
from helpers import assert_result_equals_oracle, oracle_for
from poregen_amd import synth
from poregen_amd.engine import GmoveEngine, GmoveParams, generate_kmers
L = np.full(400, 200_000, np.int64); L[::7] = 150_001; L[3::11] = 40_000
b1 = synth.make_ragged_fast(L, kind="dna_r10", seed=61)
kmers = generate_kmers(5); p = dict(kmer_size=5, scaling=1, sample_limit=4000)
o = oracle_for(kmers, **p); o.run_batch(b1)
bad = 0
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
    eng = GmoveEngine(GmoveParams(kmers=kmers, **p))   # a fresh context: the histogram buffer is grown (and filled) by this batch
    eng.submit(b1)
    res = eng.finish(); eng.close()
    try:
        assert_result_equals_oracle(res, o, sample_limit=4000)
    except AssertionError as e:
        bad += 1; print("rep", rep, "DIFFERS:", str(e)[:100])
print(sys.argv[1], ":", bad, "of the repetitions differ from the oracle")
