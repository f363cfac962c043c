#!/bin/bash
# the library's own marks (PGMOVE_TIMING=1) for the CLI at sample_limit 5000: what the staging copies of each batch cost. Files as
# tools/probe/e2e_where.sh leaves them in /tmp/pg_e2e. usage (GPU box): bash tools/probe/e2e_where.sh && bash tools/probe/e2e_staging.sh
for i in 1 2 3; do
  rm -rf /tmp/pg_e2e/o
  echo "== run $i $*"
  env PGMOVE_TIMING=1 "$@" ./bin/poregen gmove -k 5 --rna --scaling 1 --min_dur 20 --max_dur 40 --file_limit 1024 --sample_limit 5000 /tmp/pg_e2e/r.blow5 /tmp/pg_e2e/r.paf --fastq /tmp/pg_e2e/r.fastq /tmp/pg_e2e/o 2>&1 | grep -E "pgmove timing|gmove\] time"
done
