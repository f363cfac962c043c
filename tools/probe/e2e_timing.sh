#!/bin/bash
# host-side timing marks of the library (PGMOVE_TIMING=1) inside an end-to-end CLI run on config 1 as files
out=${1:-gpurun_out/e2et}; mkdir -p "$out"
python3 - <<'PY'
import os, sys, shutil
sys.path.insert(0, os.getcwd())
from poregen_amd import synth
d = "/tmp/pg_e2e"; shutil.rmtree(d, ignore_errors=True); os.makedirs(d)
b = synth.make_batch_fast(50000, kind="rna004", seed=20251004)
synth.write_blow5(b, d + "/r.blow5", compress=False); synth.write_paf_fastq(b, d + "/r")
PY
TIMEFORMAT="%R s wall"
for lim in 100 5000; do for rep in 1 2; do
  rm -rf /tmp/pg_e2e/o
  { time PGMOVE_TIMING=1 ./bin/poregen gmove -k 5 --rna --scaling 1 --min_dur 20 --max_dur 40 --file_limit 1024 --sample_limit $lim /tmp/pg_e2e/r.blow5 /tmp/pg_e2e/r.paf --fastq /tmp/pg_e2e/r.fastq /tmp/pg_e2e/o --batch_reads 50000 ; } 2>&1 | grep -E "timing|time:|wall" > "$out/lim${lim}_$rep.txt"
  echo "== sample_limit $lim run $rep"; cat "$out/lim${lim}_$rep.txt"
done; done
