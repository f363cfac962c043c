#!/bin/bash
# Kernel trace + HBM-traffic counters (separate --pmc passes) of ONE bench workload other than the headline one:
#   bash tools/mode_artifacts.sh <tag> [bench args...]     -> gpurun_out/<tag>/{bench.json,kernel_stats.csv,pmc_fetch_write_kb.json,traffic.json}
# e.g. BASELINE configs[3]:  bash tools/mode_artifacts.sh r03_k9 --kind dna_r10 --k 9 --sample-limit 1000
#      configs[2]'s limit:   bash tools/mode_artifacts.sh r03_l5000 --sample-limit 5000
# traffic.json: per kernel, FETCH_SIZE x 2 (the gfx950 correction of MI355X_MICROARCH.md for wide streaming reads; an upper bound for
# narrow / scattered ones) + WRITE_SIZE, in bytes per launch, next to the trace's average duration.
set -o pipefail
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
R=$PWD; cd /tmp && export TMPDIR=/tmp && cd $R
common="--no-cpu-baseline --no-lazy-extra --no-extras"
python3 bench.py --steps 20 --warmup 3 $common "$@" > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- python3 bench.py --steps 10 --warmup 2 --one-stream $common "$@" > $out/trace_bench.json 2> $out/trace.err || { tail -5 $out/trace.err; exit 1; }
for pass in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $pass --output-format csv -d $out/pmc_$pass -- python3 bench.py --steps 2 --warmup 1 --one-stream $common "$@" > /dev/null 2> $out/pmc_$pass.err || { tail -5 $out/pmc_$pass.err; exit 1; }
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
avg = {}
with open(f"{out}/kernel_stats.csv", "w", newline="") as f:
    w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], round(r[3], 1), round(100 * r[2] / tot, 2), r[4], r[5]])
        avg[r[0].split("(")[0].strip()] = r[3]
traffic = {}
for kn in sorted(res["FETCH_SIZE"], key=lambda k: -avg.get(k, 0)):
    f, wr = res["FETCH_SIZE"][kn] * 1024, res["WRITE_SIZE"].get(kn, 0) * 1024
    us = avg.get(kn, 0) / 1e3
    traffic[kn] = {"fetch_bytes_x2": 2 * f, "write_bytes": wr, "avg_us": us, "GBs_on_counter_traffic": ((2 * f + wr) / (us * 1e-6) / 1e9) if us else None}
    print(kn[:48].ljust(48), "%8.1f us  FETCHx2 %8.1f MB  WRITE %8.1f MB" % (us, 2 * f / 1e6, wr / 1e6))
json.dump(traffic, open(f"{out}/traffic.json", "w"), indent=1)
PY
python3 - $out/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms_per_step %.4f  whole_step_frac %.4f" % (d["ms_per_step"], d["whole_step_frac"]), {k: round(v * 1e3, 1) for k, v in d["kernels_ms_per_step"].items()})
PY
