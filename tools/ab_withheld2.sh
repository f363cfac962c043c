#!/bin/bash
# repeated A/B of the statistics stream's CU mask (process-to-process spread is a few per cent): bash tools/ab_withheld2.sh "64 128 160" 3
C="--no-cpu-baseline --no-lazy-extra --no-extras --steps 20 --warmup 3"
vals=${1:-"64 128 160"}; reps=${2:-3}
for r in $(seq 1 $reps); do
  for w in $vals; do
    for wl in "k9:--kind dna_r10 --k 9 --sample-limit 1000" "l5000:--sample-limit 5000"; do
      n=${wl%%:*}; a=${wl#*:}
      PGMOVE_STATS_CU_WITHHELD=$w timeout -k 10 200 python3 bench.py $C $a > gpurun_out/wh2_${w}_${n}_$r.json 2>/dev/null || { echo "failed $w $n"; exit 1; }
    done
  done
done
python3 - $reps $vals <<'PY'
import json, sys
reps = int(sys.argv[1]); vals = sys.argv[2:]
for w in vals:
    for n in ("k9", "l5000"):
        v = []
        for r in range(1, reps + 1):
            d = json.loads(open(f"gpurun_out/wh2_{w}_{n}_{r}.json").read().strip().splitlines()[-1]); v.append(d["ms_per_step_blocks"]["median"])
        print("withheld %3s %-6s " % (w, n) + " ".join("%.4f" % x for x in v))
PY
