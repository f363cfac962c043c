#!/bin/bash
# the gather forms (k_gather_wave, round 3's k_gather_chunks<8>, k_gather_evpair) on one box, three workloads: bash tools/ab_gather.sh <tag> [build dir names...]
set -o pipefail
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
common="--no-cpu-baseline --no-lazy-extra --no-extras --steps 10 --warmup 3 --one-stream"
run() { # name, env, lib
  for wl in "k9:--kind dna_r10 --k 9 --sample-limit 1000" "l5000:--sample-limit 5000"; do
    n=${wl%%:*}; a=${wl#*:}
    env $2 timeout -k 10 300 python3 bench.py $common $a $3 > $out/$1_$n.json 2> $out/$1_$n.err || { tail -5 $out/$1_$n.err; exit 1; }
  done
}
run wave "X=1" ""
run chunks8 "PGMOVE_GATHER_LANES=8" ""
run evpair "PGMOVE_GATHER_LANES=1" ""
for v in "$@"; do run $v "X=1" "--lib build/$v/libpgmove.so"; done
python3 - $out wave chunks8 evpair "$@" > $out/summary.txt <<'PY'
import json, sys
for v in sys.argv[2:]:
    for n in ("k9", "l5000"):
        d = json.loads(open(f"{sys.argv[1]}/{v}_{n}.json").read().strip().splitlines()[-1])
        print(v.ljust(10), n.ljust(6), "%.4f ms  k_gather %.1f us" % (d["ms_per_step"], d["kernels_ms_per_step"]["k_gather"] * 1e3))
PY
cat $out/summary.txt
