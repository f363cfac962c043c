#!/bin/bash
# k = 9 and limit-5000 bench lines for several library builds on one box: bash tools/ab_modes.sh <tag> <default|build dir name> ...
set -o pipefail
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
common="--no-cpu-baseline --no-lazy-extra --no-extras --steps 20 --warmup 3"
for v in "$@"; do
  lib=""; [ $v != default ] && lib="--lib build/$v/libpgmove.so"
  timeout -k 10 300 python3 bench.py $common --kind dna_r10 --k 9 --sample-limit 1000 $lib > $out/${v}_k9.json 2> $out/${v}_k9.err || { tail -5 $out/${v}_k9.err; exit 1; }
  timeout -k 10 300 python3 bench.py $common --sample-limit 5000 $lib > $out/${v}_l5000.json 2> $out/${v}_l5000.err || { tail -5 $out/${v}_l5000.err; exit 1; }
done
python3 - $out "$@" <<'PY'
import json, sys
for v in sys.argv[2:]:
    for n in ("k9", "l5000"):
        d = json.loads(open(f"{sys.argv[1]}/{v}_{n}.json").read().strip().splitlines()[-1])
        print(v.ljust(10), n.ljust(6), "%.4f ms " % d["ms_per_step"], " ".join("%s %.1f" % (k, x * 1e3) for k, x in d["kernels_ms_per_step"].items()))
PY
