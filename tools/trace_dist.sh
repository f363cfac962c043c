#!/bin/bash
# kernel trace of the multi-GPU step on a one-rank RCCL group. usage: bash tools/trace_dist.sh <tag>
set -o pipefail
out=gpurun_out/${1:-trace_dist}; mkdir -p $out
R=$PWD; cd /tmp && export TMPDIR=/tmp && cd $R
rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- python3 bench.py --force-dist --steps 10 --warmup 2 --no-cpu-baseline --no-lazy-extra --no-extras > $out/bench.json 2> $out/trace.err || { tail -5 $out/trace.err; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, sqlite3, sys
out = sys.argv[1]
db = sqlite3.connect(glob.glob(f"{out}/trace/**/*results.db", recursive=True)[0])
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
with open(f"{out}/kernel_stats.csv", "w", newline="") as f:
    w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows: w.writerow([r[0], r[1], r[2], round(r[3], 1), round(100 * r[2] / tot, 2), r[4], r[5]])
for r in rows[:16]: print(r[0].split("(")[0][:50].ljust(52), r[1], round(r[3], 1))
PY
