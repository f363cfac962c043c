#!/bin/bash
# the driver's command, timed on the wall clock: bash tools/bench_default_timed.sh <tag>
tag=${1:-bd}; out=gpurun_out/$tag; mkdir -p $out
t0=$(date +%s.%N)
python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
t1=$(date +%s.%N)
python3 - $out $t0 $t1 <<'PY'
import json, sys
d = json.loads(open(sys.argv[1] + "/bench.json").read().strip().splitlines()[-1]); r = d["roofline"]
print("wall %.1f s" % (float(sys.argv[3]) - float(sys.argv[2])))
print("traffic", r["traffic"], "x%.3f" % (r.get("traffic_over_algorithmic") or 0), "|", r["traffic_source"][:160], "|", r.get("traffic_live_error"))
print("value %.4g ms %.4f median %.4f frac %.3f" % (d["value"], d["ms_per_step"], d["ms_per_step_blocks"]["median"], r["frac"]))
PY
