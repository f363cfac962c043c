#!/bin/bash
# the statistics stream's CU mask (PGMOVE_STATS_CU_WITHHELD: CUs it may NOT use; default a quarter = 64) on the three workloads, one box
C="--no-cpu-baseline --no-lazy-extra --no-extras --steps 20 --warmup 3"
for w in 64 32 96 128 160 0; do
  for wl in "c1:" "k9:--kind dna_r10 --k 9 --sample-limit 1000" "l5000:--sample-limit 5000"; do
    n=${wl%%:*}; a=${wl#*:}
    PGMOVE_STATS_CU_WITHHELD=$w timeout -k 10 200 python3 bench.py $C $a > gpurun_out/wh_${w}_$n.json 2>/dev/null || { echo "failed $w $n"; exit 1; }
  done
done
python3 - <<'PY'
import json
for w in (64, 32, 96, 128, 160, 0):
    r = []
    for n in ("c1", "k9", "l5000"):
        d = json.loads(open(f"gpurun_out/wh_{w}_{n}.json").read().strip().splitlines()[-1]); b = d["ms_per_step_blocks"]
        r.append("%s %.4f (med %.4f)" % (n, d["ms_per_step"], b["median"]))
    print("withheld %3d: " % w + "  ".join(r))
PY
