#!/bin/bash
# k = 9, one stream, per-kernel lines for several builds, interleaved: bash tools/ab_k9b.sh <tag> <reps> <default|build dir name> ...
set -o pipefail
tag=$1; reps=$2; shift 2
out=gpurun_out/$tag; mkdir -p $out
common="--no-cpu-baseline --no-lazy-extra --no-extras --steps 20 --warmup 3 --one-stream --kind dna_r10 --k 9 --sample-limit 1000"
for rep in $(seq 1 $reps); do for v in "$@"; do
  lib=""; [ $v != default ] && lib="--lib build/$v/libpgmove.so"
  timeout -k 10 300 python3 bench.py $common $lib > $out/k9_${v}_$rep.json 2> $out/k9_${v}_$rep.err || { tail -5 $out/k9_${v}_$rep.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('$out/k9_${v}_$rep.json').read().strip().splitlines()[-1]); print('$v'.ljust(8), '%.4f ms ' % d['ms_per_step'], ' '.join('%s %.1f' % (k, x * 1e3) for k, x in d['kernels_ms_per_step'].items()))"
done; done
