#!/bin/bash
# full GPU suite + three plain bench runs. usage: bash tools/ab_quick.sh <tag>
set -o pipefail
tag=${1:-abq}; out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1 || { tail -30 $out/pytest_gpu.txt; exit 1; }
tail -1 $out/pytest_gpu.txt
for rep in 1 2 3; do
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-lazy-extra --no-extras > $out/plain_$rep.json 2> $out/plain_$rep.err || { tail -5 $out/plain_$rep.err; exit 1; }
done
timeout -k 10 200 python3 bench.py --force-dist --no-cpu-baseline --no-lazy-extra --no-extras > $out/dist_1.json 2> $out/dist_1.err || { tail -5 $out/dist_1.err; exit 1; }
python3 - $out <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/*_[123].json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], "%.4f" % d["ms_per_step"], {k: round(v * 1e3, 1) for k, v in d["kernels_ms_per_step"].items()})
PY
