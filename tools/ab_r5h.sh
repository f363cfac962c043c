#!/bin/bash
# one-stream per-kernel lines of C1, sample_limit 5000 and k = 9 for several builds, interleaved: bash tools/ab_r5h.sh <tag> <reps> <default|build dir name> ...
set -o pipefail
tag=$1; reps=$2; shift 2
out=gpurun_out/$tag; mkdir -p $out
common="--no-cpu-baseline --no-lazy-extra --no-extras --steps 20 --warmup 3 --one-stream"
for rep in $(seq 1 $reps); do for v in "$@"; do
  lib=""; [ $v != default ] && lib="--lib build/$v/libpgmove.so"
  timeout -k 10 300 python3 bench.py $common --sample-limit 5000 $lib > $out/l5000_${v}_$rep.json 2> $out/l5000_${v}_$rep.err || { tail -5 $out/l5000_${v}_$rep.err; exit 1; }
  timeout -k 10 300 python3 bench.py $common $lib > $out/c1_${v}_$rep.json 2> $out/c1_${v}_$rep.err || { tail -5 $out/c1_${v}_$rep.err; exit 1; }
  timeout -k 10 300 python3 bench.py $common --kind dna_r10 --k 9 --sample-limit 1000 $lib > $out/k9_${v}_$rep.json 2> $out/k9_${v}_$rep.err || { tail -5 $out/k9_${v}_$rep.err; exit 1; }
done; done
python3 - $out $reps "$@" > $out/summary.txt <<'PY'
import json, sys
reps = int(sys.argv[2])
for w in ("l5000", "c1", "k9"):
    for v in sys.argv[3:]:
        for rep in range(1, reps + 1):
            d = json.loads(open(f"{sys.argv[1]}/{w}_{v}_{rep}.json").read().strip().splitlines()[-1])
            print(w.ljust(6), v.ljust(10), "%.4f ms " % d["ms_per_step"], " ".join("%s %.1f" % (k, x * 1e3) for k, x in d["kernels_ms_per_step"].items()))
PY
cat $out/summary.txt
