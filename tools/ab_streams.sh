#!/bin/bash
# one stream against the default two streams, per workload, alternating on one box. usage: bash tools/ab_streams.sh <tag>
set -o pipefail
tag=$1; out=gpurun_out/$tag; mkdir -p $out
common="--no-cpu-baseline --no-lazy-extra --no-extras --steps 20 --warmup 3"
for rep in 1 2; do
  for m in "c1:" "k9:--kind dna_r10 --k 9 --sample-limit 1000" "l5000:--sample-limit 5000"; do
    name=${m%%:*}; flags=${m#*:}
    timeout -k 10 300 python3 bench.py $common $flags > $out/${name}_two_$rep.json 2> $out/${name}_two_$rep.err || { tail -5 $out/${name}_two_$rep.err; exit 1; }
    timeout -k 10 300 python3 bench.py $common $flags --one-stream > $out/${name}_one_$rep.json 2> $out/${name}_one_$rep.err || { tail -5 $out/${name}_one_$rep.err; exit 1; }
  done
done
python3 - $out <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f).ljust(22), "%.4f ms  frac %.3f  kernels_sum %.4f" % (d["ms_per_step"], d["whole_step_frac"], d["whole_step"]["kernels_sum_ms"]))
PY
