"""bench.py's end_to_end leg alone (the CLI as a child process on configs[1] written as files, sample_limit 100 and 5000)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import argparse
import bench
from poregen_amd import synth
host = synth.make_batch_fast(50000, kind="rna004", seed=20251003 + 1)
args = argparse.Namespace(sample_limit=100, k=5, kind="rna004")
r = bench.end_to_end(host, args)
for k, v in r["runs"].items():
    print(k, "wall %.3f s" % v["wall_s"], v["stages"][-3:])
