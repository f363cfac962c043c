#!/usr/bin/env python3
"""Kernel-to-kernel gaps on the chain's queue and the batch period from a rocprofv3 --kernel-trace database (what a batch boundary, an
event record or a cross-stream wait costs). usage: python3 tools/trace_gaps.py <dir or results.db> [--timeline N]"""
import collections, glob, os, re, sqlite3, sys
p = sys.argv[1]
if os.path.isdir(p):
    p = sorted(glob.glob(os.path.join(p, "**", "*results.db"), recursive=True))[-1]
c = sqlite3.connect(p).cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = list(c.execute(f"select s.kernel_name, d.start, d.end, d.queue_id from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
def short(n):
    m = re.search(r"\d+(k_[a-z_0-9]+?)(I|P|1|j|v|E|\d)", n)
    return m.group(1) if m else n[:26]
names = [short(r[0]) for r in rows]
mainq = collections.Counter(r[3] for r, n in zip(rows, names) if n == "k_batch_init").most_common(1)[0][0]
seq = [(n, r[1], r[2]) for r, n in zip(rows, names) if r[3] == mainq]
gaps = collections.defaultdict(list)
for a, b in zip(seq, seq[1:]):
    gaps[(a[0], b[0])].append((b[1] - a[2]) / 1e3)
for k, v in gaps.items():
    if len(v) > 50:
        v = sorted(v); print("  %-18s -> %-18s n=%d median gap %.2f us (p10 %.2f, p90 %.2f)" % (k[0], k[1], len(v), v[len(v) // 2], v[len(v) // 10], v[len(v) * 9 // 10]))
st = [s[1] for s in seq if s[0] == "k_batch_init"]
d = sorted((b - a) / 1e3 for a, b in zip(st, st[1:]))
print("  period median %.1f us (%d batches)" % (d[len(d) // 2], len(st)))
if "--timeline" in sys.argv:
    n = int(sys.argv[sys.argv.index("--timeline") + 1])
    idx = [i for i, x in enumerate(names) if x == "k_batch_init"]
    mid = idx[len(idx) * 2 // 3]; t0 = rows[mid][1]
    for r, x in zip(rows[mid - 3:mid + n], names[mid - 3:mid + n]):
        print(f"{(r[1]-t0)/1e3:9.1f} -> {(r[2]-t0)/1e3:9.1f}  dur {(r[2]-r[1])/1e3:6.1f}  q{r[3]}  {x}")
