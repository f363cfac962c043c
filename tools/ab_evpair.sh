#!/bin/bash
# the event-pair gather (PGMOVE_GATHER_LANES=1) for several library builds on one box, both dense workloads: bash tools/ab_evpair.sh <tag> <default|build dir> ...
set -o pipefail
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
common="--no-cpu-baseline --no-lazy-extra --no-extras --steps 10 --warmup 3 --one-stream"
for v in "$@"; do
  lib=""; [ $v != default ] && lib="--lib build/$v/libpgmove.so"
  for wl in "k9:--kind dna_r10 --k 9 --sample-limit 1000" "l5000:--sample-limit 5000"; do
    n=${wl%%:*}; a=${wl#*:}
    PGMOVE_GATHER_LANES=1 timeout -k 10 300 python3 bench.py $common $a $lib > $out/${v}_$n.json 2> $out/${v}_$n.err || { tail -5 $out/${v}_$n.err; exit 1; }
    python3 -c "import json,sys; d=json.loads(open('$out/${v}_$n.json').read().strip().splitlines()[-1]); print('$v'.ljust(10), '$n'.ljust(6), '%.4f ms  k_gather %.1f us' % (d['ms_per_step'], d['kernels_ms_per_step']['k_gather']*1e3))"
  done
done
