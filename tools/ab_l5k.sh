#!/bin/bash
# sample_limit 5000, one stream, wall per step only (no profile differences): several builds / environments interleaved, 3 reps
set -o pipefail
out=gpurun_out/$1; shift; mkdir -p $out
common="--no-cpu-baseline --no-lazy-extra --no-extras --steps 40 --warmup 3 --one-stream --sample-limit 5000"
for rep in 1 2 3 4; do for v in "$@"; do
  lib=""; envs="PG_X=1"
  case $v in
    default) ;;
    nolong) envs="PGMOVE_NO_LONG_SPLIT=1";;
    *) lib="--lib build/$v/libpgmove.so";;
  esac
  env $envs timeout -k 10 300 python3 bench.py $common $lib > $out/${v}_$rep.json 2> $out/${v}_$rep.err || { tail -5 $out/${v}_$rep.err; exit 1; }
  python3 -c "
import json,sys; d=json.loads(open('$out/${v}_$rep.json').read().strip().splitlines()[-1]); print('$v'.ljust(8), '%.4f' % d['ms_per_step'], d['ms_per_step_blocks']['min'], d['ms_per_step_blocks']['median'])"
done; done
