#!/bin/bash
# A/B of library builds on one box: bash tools/ab_lib.sh <tag> <build dir names under build/ ...>
set -o pipefail
out=gpurun_out/$1; shift; mkdir -p $out
for rep in 1 2 3; do for v in "$@"; do
  timeout -k 10 200 python3 bench.py --lib build/$v/libpgmove.so --no-cpu-baseline --no-lazy-extra --no-extras > $out/${v}_$rep.json 2> $out/${v}_$rep.err || { tail -5 $out/${v}_$rep.err; exit 1; }
done; done
python3 - $out <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); k = d["kernels_ms_per_step"]
    print(f.split("/")[-1].ljust(18), "%.4f" % d["ms_per_step"], " ".join("%s %.1f" % (n, v * 1e3) for n, v in k.items()))
PY
