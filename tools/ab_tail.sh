#!/bin/bash
# A/B of PG_FLAG_OVERLAP_TAIL on one box. usage: bash tools/ab_tail.sh <tag>
set -o pipefail
tag=${1:-abt}; out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $out/pytest_parity.txt 2>&1 || { tail -30 $out/pytest_parity.txt; exit 1; }
tail -1 $out/pytest_parity.txt
for rep in 1 2 3; do
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-lazy-extra --no-extras > $out/plain_$rep.json 2> $out/plain_$rep.err || { tail -5 $out/plain_$rep.err; exit 1; }
  timeout -k 10 200 python3 bench.py --overlap-tail --no-cpu-baseline --no-lazy-extra --no-extras > $out/tail_$rep.json 2> $out/tail_$rep.err || { tail -5 $out/tail_$rep.err; exit 1; }
done
timeout -k 10 200 python3 bench.py --sample-limit 5000 --no-cpu-baseline --no-lazy-extra --no-extras > $out/plain5000.json 2> $out/plain5000.err || { tail -5 $out/plain5000.err; exit 1; }
timeout -k 10 200 python3 bench.py --sample-limit 5000 --overlap-tail --no-cpu-baseline --no-lazy-extra --no-extras > $out/tail5000.json 2> $out/tail5000.err || { tail -5 $out/tail5000.err; exit 1; }
python3 - $out <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], "%.4f" % d["ms_per_step"], "%.3f" % d["roofline"]["frac"])
PY
