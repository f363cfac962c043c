#!/bin/bash
# k_read_stats capped in waves per CU by unused LDS (PGMOVE_STATS_LDS_PAD bytes per wave; 4.5 KB are its own) x the CUs withheld from its stream
C="--no-cpu-baseline --no-lazy-extra --no-extras --steps 20 --warmup 3"
for cfg in "0:64" "0:0" "2200:0" "3500:0" "5500:0" "8800:0" "3500:32" "5500:32" "0:64"; do
  pad=${cfg%%:*}; wh=${cfg#*:}
  PGMOVE_STATS_LDS_PAD=$pad PGMOVE_STATS_CU_WITHHELD=$wh timeout -k 10 200 python3 bench.py $C > gpurun_out/lp_${pad}_${wh}.json 2>/dev/null || { echo failed $cfg; exit 1; }
  python3 - $pad $wh <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/lp_{sys.argv[1]}_{sys.argv[2]}.json").read().strip().splitlines()[-1]); b = d["ms_per_step_blocks"]
print("pad %5s withheld %3s: %.4f ms (blocks %.4f %.4f %.4f)" % (sys.argv[1], sys.argv[2], d["ms_per_step"], b["min"], b["median"], b["max"]))
PY
done
