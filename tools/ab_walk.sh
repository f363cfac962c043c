#!/bin/bash
# fused walk+events kernel: parity suite, a fuzz run, and the A/B against the two-launch form. usage: bash tools/ab_walk.sh <tag> [fuzz cases]
set -o pipefail
tag=${1:-abw}; out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1 || { tail -30 $out/pytest_gpu.txt; exit 1; }
tail -2 $out/pytest_gpu.txt
timeout -k 10 500 python3 tools/fuzz_gpu.py ${2:-150} 777 > $out/fuzz.txt 2>&1 || { tail -20 $out/fuzz.txt; exit 1; }
tail -3 $out/fuzz.txt
for rep in 1 2; do
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-lazy-extra --no-extras > $out/fused_$rep.json 2> $out/fused_$rep.err || { tail -5 $out/fused_$rep.err; exit 1; }
  timeout -k 10 200 python3 bench.py --split-walk --no-cpu-baseline --no-lazy-extra --no-extras > $out/split_$rep.json 2> $out/split_$rep.err || { tail -5 $out/split_$rep.err; exit 1; }
done
python3 - $out <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/*_[12].json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], "%.4f" % d["ms_per_step"], {k: round(v * 1e3, 1) for k, v in d["kernels_ms_per_step"].items()})
PY
