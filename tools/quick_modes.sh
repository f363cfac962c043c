#!/bin/bash
# three bench lines on one box: the headline workload, BASELINE configs[3] (k = 9) and configs[2]'s limit (5000); per-kernel times only
# usage: bash tools/quick_modes.sh <tag> [--lib build/x/libpgmove.so]
set -o pipefail
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
common="--no-cpu-baseline --no-lazy-extra --no-extras --steps 20 --warmup 3"
timeout -k 10 300 python3 bench.py $common "$@" > $out/c1.json 2> $out/c1.err || { tail -5 $out/c1.err; exit 1; }
timeout -k 10 300 python3 bench.py $common --kind dna_r10 --k 9 --sample-limit 1000 "$@" > $out/k9.json 2> $out/k9.err || { tail -5 $out/k9.err; exit 1; }
timeout -k 10 300 python3 bench.py $common --sample-limit 5000 "$@" > $out/l5000.json 2> $out/l5000.err || { tail -5 $out/l5000.err; exit 1; }
python3 - $out <<'PY'
import json, sys
for n in ("c1", "k9", "l5000"):
    d = json.loads(open(f"{sys.argv[1]}/{n}.json").read().strip().splitlines()[-1])
    print(n.ljust(6), "%.4f ms  frac %.3f " % (d["ms_per_step"], d["whole_step_frac"]), " ".join("%s %.1f" % (k, v * 1e3) for k, v in d["kernels_ms_per_step"].items()))
PY
