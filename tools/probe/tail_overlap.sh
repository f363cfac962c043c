#!/bin/bash
# A/B of the second-stream modes (PG_FLAG_OVERLAP, PG_FLAG_OVERLAP_TAIL) with N compute units withheld from the statistics
# stream (PGMOVE_STATS_CU_WITHHELD). usage (GPU box): bash tools/probe/tail_overlap.sh OUTDIR "--overlap" "0 8 16 32"
out=${1:-gpurun_out/tail}; mkdir -p "$out"
flag=${2:---overlap-tail}
run() { # name, env N, extra flags
    PGMOVE_STATS_CU_WITHHELD=$2 timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras $3 > "$out/$1.json" 2> "$out/$1.err" || { echo "$1 failed"; tail -3 "$out/$1.err"; return 1; }
    python - "$out/$1.json" "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("%-24s %.4f ms  %s" % (sys.argv[2], d["ms_per_step"], {k: round(v * 1e3, 1) for k, v in d["kernels_ms_per_step"].items()}))
PY
}
run serial 0 "" || exit 1
for n in ${3:-0 8 16 32}; do run "${flag#--}_$n" $n "$flag" || exit 1; done
run serial_again 0 ""
