#!/bin/bash
# quick A/B of the fused walk kernel (parity file + two bench pairs). usage: bash tools/ab_walk_quick.sh <tag>
set -o pipefail
tag=${1:-abq}; out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $out/pytest_parity.txt 2>&1 || { tail -30 $out/pytest_parity.txt; exit 1; }
tail -1 $out/pytest_parity.txt
for rep in 1 2; do
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-lazy-extra --no-extras > $out/fused_$rep.json 2> $out/fused_$rep.err || { tail -5 $out/fused_$rep.err; exit 1; }
  timeout -k 10 200 python3 bench.py --split-walk --no-cpu-baseline --no-lazy-extra --no-extras > $out/split_$rep.json 2> $out/split_$rep.err || { tail -5 $out/split_$rep.err; exit 1; }
done
python3 - $out <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/*_[12].json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], "%.4f" % d["ms_per_step"], {k: round(v * 1e3, 1) for k, v in d["kernels_ms_per_step"].items()})
PY
