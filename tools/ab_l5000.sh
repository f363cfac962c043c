#!/bin/bash
# sample_limit 5000 bench lines (one stream, per-kernel times) for several library builds on one box, each twice: bash tools/ab_l5000.sh <tag> <default|build dir name> ...
set -o pipefail
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
common="--no-cpu-baseline --no-lazy-extra --no-extras --steps 20 --warmup 3 --one-stream --sample-limit 5000"
for rep in 1 2; do for v in "$@"; do
  lib=""; [ $v != default ] && lib="--lib build/$v/libpgmove.so"
  timeout -k 10 300 python3 bench.py $common $lib > $out/${v}_$rep.json 2> $out/${v}_$rep.err || { tail -5 $out/${v}_$rep.err; exit 1; }
done; done
python3 - $out "$@" > $out/summary.txt <<'PY'
import json, sys
for v in sys.argv[2:]:
    for rep in (1, 2):
        d = json.loads(open(f"{sys.argv[1]}/{v}_{rep}.json").read().strip().splitlines()[-1])
        print(v.ljust(10), "%.4f ms " % d["ms_per_step"], " ".join("%s %.1f" % (k, x * 1e3) for k, x in d["kernels_ms_per_step"].items()))
PY
cat $out/summary.txt
