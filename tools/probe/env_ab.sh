#!/bin/bash
# A/B of one library build under two environments on ONE box, alternating: bash tools/probe/env_ab.sh OUTDIR "VAR=1" [reps]
out=${1:-gpurun_out/envab}; mkdir -p "$out"; var=$2; reps=${3:-3}
run() { # name, env assignment
    env $2 timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras > "$out/$1.json" 2> "$out/$1.err" || { echo "$1 failed"; tail -3 "$out/$1.err"; return 1; }
    python - "$out/$1.json" "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); t = d.get("two_stream_mode") or {}
print("%-14s %.4f ms  two-stream %.4f ms  %s" % (sys.argv[2], d["ms_per_step"], t.get("ms_per_step", 0), {k: round(v * 1e3, 1) for k, v in d["kernels_ms_per_step"].items()}))
PY
}
for rep in $(seq 1 $reps); do run default_$rep "PG_AB_UNUSED=1" || exit 1; run with_$rep "$var" || exit 1; done
