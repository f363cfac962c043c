#!/usr/bin/env python3
"""bench.py's ragged_mode alone (one JSON object): python tools/ragged_quick.py [--lib build/x/libpgmove.so]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "--lib" in sys.argv:
    from poregen_amd import _abi
    _abi.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
import torch
import bench
torch.cuda.set_device(0)
r = bench.ragged_mode(torch.device("cuda", 0))
for k in ("k9", "k5"):
    x = r[k]
    print(k, "step %.4f ms frac %.3f | k_read_stats %.1f us frac %.3f | one wave per read: step %.4f ms, k_read_stats %.1f us frac %.3f | split reads %s" % (
        x["ms_per_step"], x["whole_step_frac"], x["k_read_stats"]["avg_launch_ms"] * 1e3, x["k_read_stats"]["frac"], x["one_wave_per_read"]["ms_per_step"],
        x["one_wave_per_read"]["k_read_stats_ms"] * 1e3, x["one_wave_per_read"]["k_read_stats_frac"], x.get("counters")), file=sys.stderr)
print(json.dumps(r))
