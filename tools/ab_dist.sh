#!/bin/bash
# A/B of the multi-GPU step on a one-rank RCCL group: statistics behind the issue of the all_gather (default) or inside pg_count.
# usage (GPU box, repo root): bash tools/ab_dist.sh <tag> [nopytest]  -> gpurun_out/<tag>/
set -o pipefail
tag=${1:-ab}; out=gpurun_out/$tag; mkdir -p $out
if [ "$2" != "nopytest" ]; then
  timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $out/pytest_parity.txt 2>&1 || { tail -30 $out/pytest_parity.txt; exit 1; }
  tail -2 $out/pytest_parity.txt
fi
for rep in 1 2; do
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-lazy-extra --no-extras > $out/plain_$rep.json 2> $out/plain_$rep.err || { tail -5 $out/plain_$rep.err; exit 1; }
  timeout -k 10 200 python3 bench.py --force-dist --no-cpu-baseline --no-lazy-extra --no-extras > $out/defer_$rep.json 2> $out/defer_$rep.err || { tail -5 $out/defer_$rep.err; exit 1; }
  timeout -k 10 200 python3 bench.py --force-dist --no-defer --no-cpu-baseline --no-lazy-extra --no-extras > $out/nodefer_$rep.json 2> $out/nodefer_$rep.err || { tail -5 $out/nodefer_$rep.err; exit 1; }
done
python3 - $out <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["ms_per_step"], d["roofline"]["frac"])
PY
