#!/bin/bash
# where does the time between "[main] Real time" and the process's end go? orderly teardown (POREGEN_CLEAN_EXIT=1) against _exit
python3 - <<'PY'
import os, sys, shutil
sys.path.insert(0, os.getcwd())
from poregen_amd import synth
d = "/tmp/pg_e2e"; shutil.rmtree(d, ignore_errors=True); os.makedirs(d)
b = synth.make_batch_fast(50000, kind="rna004", seed=20251004)
synth.write_blow5(b, d + "/r.blow5", compress=False); synth.write_paf_fastq(b, d + "/r")
PY
TIMEFORMAT="%R s wall %U user %S sys"
for mode in fast clean; do for lim in 100 5000; do
  rm -rf /tmp/pg_e2e/o
  echo "== $mode exit, sample_limit $lim"
  if [ $mode = clean ]; then export POREGEN_CLEAN_EXIT=1; else unset POREGEN_CLEAN_EXIT; fi
  { time ./bin/poregen gmove -k 5 --rna --scaling 1 --min_dur 20 --max_dur 40 --file_limit 1024 --sample_limit $lim /tmp/pg_e2e/r.blow5 /tmp/pg_e2e/r.paf --fastq /tmp/pg_e2e/r.fastq /tmp/pg_e2e/o ; } 2>&1 | grep -E "release the device|Real time|wall"
done; done
