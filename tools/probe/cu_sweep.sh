#!/bin/bash
# how many CUs to withhold from the statistics stream in the default two-stream mode (PGMOVE_STATS_CU_WITHHELD), per workload, one box
out=gpurun_out/${1:-cusweep}; mkdir -p $out
common="--no-cpu-baseline --no-lazy-extra --no-extras --steps 30 --warmup 3"
for w in 64 32 48 80 96 128 64; do
  for m in "c1:" "l5000:--sample-limit 5000"; do
    name=${m%%:*}; flags=${m#*:}
    PGMOVE_STATS_CU_WITHHELD=$w timeout -k 10 200 python3 bench.py $common $flags > $out/${name}_$w.json 2> $out/${name}_$w.err || { tail -3 $out/${name}_$w.err; exit 1; }
    python3 -c "
import json,sys
d=json.loads(open('$out/${name}_$w.json').read().strip().splitlines()[-1]); print('$name withheld $w: %.4f ms  frac %.3f' % (d['ms_per_step'], d['whole_step_frac']))"
  done
done
