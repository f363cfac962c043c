#!/bin/bash
# k = 9 bench lines (one stream, per-kernel times) for several library builds on one box: bash tools/ab_k9.sh <tag> <default|build dir name> ...
set -o pipefail
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
common="--no-cpu-baseline --no-lazy-extra --no-extras --steps 10 --warmup 3 --one-stream --kind dna_r10 --k 9 --sample-limit 1000"
for v in "$@"; do
  lib=""; [ $v != default ] && lib="--lib build/$v/libpgmove.so"
  timeout -k 10 300 python3 bench.py $common $lib > $out/${v}.json 2> $out/${v}.err || { tail -5 $out/${v}.err; exit 1; }
done
python3 - $out "$@" > $out/summary.txt <<'PY'
import json, sys
for v in sys.argv[2:]:
    d = json.loads(open(f"{sys.argv[1]}/{v}.json").read().strip().splitlines()[-1])
    print(v.ljust(10), "%.4f ms " % d["ms_per_step"], " ".join("%s %.1f" % (k, x * 1e3) for k, x in d["kernels_ms_per_step"].items()))
PY
cat $out/summary.txt
