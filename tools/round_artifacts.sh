#!/bin/bash
# Everything the round's profile artefacts come from, in one GPU call: the GPU test suite, the bench line, the kernel
# trace of the same command and the HBM-traffic counters of the dominant kernel (separate --pmc passes).
# The traced / counted passes run on ONE stream (--one-stream): in the default two-stream mode the statistics kernel shares the chip with
# the ranking kernels by design, and a kernel's duration then says little about the kernel (bench.py's own per-kernel times are
# taken the same way, in its profile-mode pass).
# usage (from the repo root, on the GPU box): bash tools/round_artifacts.sh <tag>   -> gpurun_out/<tag>/
set -o pipefail
tag=${1:-r03}
out=gpurun_out/$tag; mkdir -p $out
R=$PWD; cd /tmp && export TMPDIR=/tmp && cd $R
if [ "$2" != "nopytest" ]; then
  timeout -k 10 900 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1 || { tail -20 $out/pytest_gpu.txt; exit 1; }
  tail -2 $out/pytest_gpu.txt
fi
python3 bench.py > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- python3 bench.py --steps 10 --warmup 2 --one-stream --no-cpu-baseline --no-lazy-extra --no-extras > $out/trace_bench.json 2> $out/trace.err || { tail -5 $out/trace.err; exit 1; }
# the same trace of the DEFAULT command (two streams): what a driver-side trace of `python bench.py` sees; kernels that share the chip take longer there
rocprofv3 --kernel-trace --stats -d $out/trace2 -o trace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-lazy-extra --no-extras > $out/trace2_bench.json 2> $out/trace2.err || { tail -5 $out/trace2.err; exit 1; }
for pass in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $pass --output-format csv -d $out/pmc_$pass -- python3 bench.py --steps 2 --warmup 1 --one-stream --no-cpu-baseline --no-lazy-extra --no-extras > /dev/null 2> $out/pmc_$pass.err || { tail -5 $out/pmc_$pass.err; exit 1; }
done
python3 - "$out" <<'PY'
import csv, glob, collections, json, sqlite3, sys
out = sys.argv[1]
res = {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{out}/pmc_{name}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == name: acc[row["Kernel_Name"].split("(")[0].strip()].append(float(row["Counter_Value"]))
        res[name] = {k: sum(v) / len(v) for k, v in acc.items()}
json.dump(res, open(f"{out}/pmc_fetch_write_kb.json", "w"), indent=1)
db = sqlite3.connect(glob.glob(f"{out}/trace/**/*results.db", recursive=True)[0])
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
with open(f"{out}/kernel_stats.csv", "w", newline="") as f:
    w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows: w.writerow([r[0], r[1], r[2], round(r[3], 1), round(100 * r[2] / tot, 2), r[4], r[5]])
db2 = sqlite3.connect(glob.glob(f"{out}/trace2/**/*results.db", recursive=True)[0])
rows2 = db2.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot2 = sum(r[2] for r in rows2)
with open(f"{out}/kernel_stats_two_streams.csv", "w", newline="") as f:
    w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows2: w.writerow([r[0], r[1], r[2], round(r[3], 1), round(100 * r[2] / tot2, 2), r[4], r[5]])
for kn in res["FETCH_SIZE"]:
    print(kn[:60].ljust(60), "FETCH KB %.0f (x2 on gfx950 for wide streaming reads)" % res["FETCH_SIZE"][kn], "WRITE KB %.0f" % res["WRITE_SIZE"].get(kn, 0))
PY
cat $out/bench.json
