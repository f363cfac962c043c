#!/bin/bash
# library variants at BASELINE configs[3] (k = 9), alternating with the default build on one box: bash tools/ab_k9.sh <tag> <variant names under build/ ...>
out=gpurun_out/$1; shift; mkdir -p $out
common="--no-cpu-baseline --no-lazy-extra --no-extras --steps 20 --warmup 3 --kind dna_r10 --k 9 --sample-limit 1000"
for v in default "$@" default "$@"; do
  lib=""; [ $v != default ] && lib="--lib build/$v/libpgmove.so"
  timeout -k 10 250 python3 bench.py $common $lib > $out/$v.json 2> $out/$v.err || { tail -3 $out/$v.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('$out/$v.json').read().strip().splitlines()[-1]); k=d['kernels_ms_per_step']
print('$v'.ljust(10), '%.4f ms' % d['ms_per_step'], ' '.join('%s %.1f' % (a, b*1e3) for a, b in k.items()))"
done
