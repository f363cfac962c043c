#!/bin/bash
# where does `poregen gmove` spend its wall time at sample_limit 5000 outside its own clock? loader baseline, page-fault counts,
# transparent huge pages for malloc'ed memory (GLIBC_TUNABLES), host text against device text. usage (GPU box): bash tools/probe/e2e_where.sh
python3 - <<'PY'
import os, sys, shutil
sys.path.insert(0, os.getcwd())
from poregen_amd import synth
d = "/tmp/pg_e2e"; shutil.rmtree(d, ignore_errors=True); os.makedirs(d)
b = synth.make_batch_fast(50000, kind="rna004", seed=20251004)
synth.write_blow5(b, d + "/r.blow5", compress=False); synth.write_paf_fastq(b, d + "/r")
PY
echo "THP enabled: $(cat /sys/kernel/mm/transparent_hugepage/enabled 2>/dev/null)  defrag: $(cat /sys/kernel/mm/transparent_hugepage/defrag 2>/dev/null)  shmem: $(cat /sys/kernel/mm/transparent_hugepage/shmem_enabled 2>/dev/null)"
df /tmp | tail -1
TIMEFORMAT="%R s wall %U user %S sys"
echo "== loader baseline: poregen with no arguments"
for i in 1 2 3; do { time ./bin/poregen > /dev/null 2>&1 ; } 2>&1 | tail -1; done
run() { # label, env...
  for i in 1 2 3; do
    rm -rf /tmp/pg_e2e/o
    echo "== $1, run $i"
    { time env "${@:2}" ./bin/poregen gmove -k 5 --rna --scaling 1 --min_dur 20 --max_dur 40 --file_limit 1024 --sample_limit 5000 /tmp/pg_e2e/r.blow5 /tmp/pg_e2e/r.paf --fastq /tmp/pg_e2e/r.fastq /tmp/pg_e2e/o ; } 2>&1 | grep -E "reading \+ parsing|from the start|Real time|wall|dump probe"
  done
}
run "default" A=1
run "malloc with transparent huge pages" GLIBC_TUNABLES=glibc.malloc.hugetlb=1
run "dump probe" POREGEN_DUMP_PROBE=1
