#!/bin/bash
# poregen gmove at sample_limit 5000 (three batches): batches' samples held on the device and merged there (default) against the
# per-batch download + host merge (PGMOVE_HOST_MERGE=1); files as tools/probe/exit_probe.sh leaves them in /tmp/pg_e2e
TIMEFORMAT="%R s wall"
for rep in 1 2 3; do for mode in device host; do
  rm -rf /tmp/pg_e2e/o
  if [ $mode = host ]; then export PGMOVE_HOST_MERGE=1; else unset PGMOVE_HOST_MERGE; fi
  echo "== $mode merge, run $rep"
  { time ./bin/poregen gmove -k 5 --rna --scaling 1 --min_dur 20 --max_dur 40 --file_limit 1024 --sample_limit 5000 /tmp/pg_e2e/r.blow5 /tmp/pg_e2e/r.paf --fastq /tmp/pg_e2e/r.fastq /tmp/pg_e2e/o ; } 2>&1 | grep -E "reading \+ parsing|Real time|wall"
done; done
