#!/bin/bash
# kernel trace of bench.py in two-stream mode (PG_FLAG_OVERLAP): per-dispatch start / end and queue, for a timeline of the two streams
out=${1:-gpurun_out/tl}; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $out/trace -o tl -- python3 bench.py --overlap --steps 12 --warmup 3 --no-cpu-baseline --no-lazy-extra --no-extras > $out/bench.json 2> $out/trace.err || { tail -5 $out/trace.err; exit 1; }
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $out/timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the timed loop: the last 12 steps = the last 12 k_gather dispatches
g = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_gather")]
lo = g[6] + 1; hi = g[11] + 1   # five steps of the timed loop (3 warm-up + 12 timed steps come first; the profiled passes behind them synchronise)
t0 = int(rows[lo]["Start_Timestamp"])
print("cols:", list(rows[0].keys()))
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%-28s q=%-4s start %9.1f us  end %9.1f us  dur %7.1f" % (r["Kernel_Name"].split("(")[0][:28], r.get("Queue_Id", "?"), s / 1e3, e / 1e3, (e - s) / 1e3))
PY
head -80 $out/timeline.txt
