#!/bin/bash
# library variants at K9 and L5000, alternating with the default build on one box: bash tools/ab_modes2.sh <tag> <variant names under build/ ...>
out=gpurun_out/$1; shift; mkdir -p $out
common="--no-cpu-baseline --no-lazy-extra --no-extras --steps 20 --warmup 3"
for m in "k9:--kind dna_r10 --k 9 --sample-limit 1000" "l5000:--sample-limit 5000"; do
  name=${m%%:*}; flags=${m#*:}
  for v in default "$@" default "$@"; do
    lib=""; [ $v != default ] && lib="--lib build/$v/libpgmove.so"
    timeout -k 10 250 python3 bench.py $common $flags $lib > $out/${name}_$v.json 2> $out/${name}_$v.err || { tail -3 $out/${name}_$v.err; exit 1; }
    python3 -c "
import json
d=json.loads(open('$out/${name}_$v.json').read().strip().splitlines()[-1]); k=d['kernels_ms_per_step']
print('$name', '$v'.ljust(10), '%.4f ms' % d['ms_per_step'], 'k_gather %.1f' % (k['k_gather']*1e3))"
  done
done
