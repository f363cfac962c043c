#!/bin/bash
# round 6 A/B on one box: bash tools/ab_r6.sh <tag> <reps> <case> ...   case = name[:lib][:ENV=VAL,ENV=VAL][:extra bench flags with _ for spaces]
#   e.g.  c1  l5k:emit6  c1::PGMOVE_STATS_CU_WITHHELD=32   k9:noxcd
# name prefix picks the workload: c1* (headline, two streams), l5k* (sample_limit 5000), k9* (dna_r10 k = 9 limit 1000); suffix "1s" = --one-stream
set -o pipefail
tag=$1; reps=$2; shift 2
out=gpurun_out/$tag; mkdir -p $out
for rep in $(seq 1 $reps); do for c in "$@"; do
  IFS=: read -r name lib envs extra <<< "$c"
  flags="--no-cpu-baseline --no-lazy-extra --no-extras --steps 40 --warmup 10"
  case $name in
    l5k*) flags="$flags --sample-limit 5000";;
    k9*)  flags="$flags --steps 20 --kind dna_r10 --k 9 --sample-limit 1000";;
  esac
  case $name in *1s) flags="$flags --one-stream";; esac
  [ -n "$lib" ] && flags="$flags --lib build/$lib/libpgmove.so"
  [ -n "$extra" ] && flags="$flags ${extra//_/ }"
  e="PG_X=1"; [ -n "$envs" ] && e="${envs//,/ }"
  id="${name}_${lib:-default}_$(echo "$envs$extra" | tr -c 'A-Za-z0-9\n' '_')_$rep"
  env $e timeout -k 10 300 python3 bench.py $flags > $out/$id.json 2> $out/$id.err || { tail -5 $out/$id.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('$out/$id.json').read().strip().splitlines()[-1]); b=d['ms_per_step_blocks']
print('$c'.ljust(44), 'first %.4f min %.4f med %.4f |' % (d['ms_per_step'], b['min'], b['median']), ' '.join('%s %.1f' % (k.replace('k_',''), x * 1e3) for k, x in d['kernels_ms_per_step'].items()))"
done; done
