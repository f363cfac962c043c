#!/bin/bash
# where the end-to-end wall time of `poregen gmove` goes: process start-up (dynamic loading of the HIP runtime), the stages inside
# gmove, exit. usage (GPU box): bash tools/e2e_probe.sh <tag>
out=gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
G=tests/golden/single_read
TIMEFORMAT="%R s wall %U user %S sys"
{
echo "== poregen --version (links libpgmove + libamdhip64, no GPU call) x3"
for i in 1 2 3; do time ./bin/poregen --version > /dev/null; done
echo "== poregen gmove on the one-read fixture (context created, one tiny batch) x3"
for i in 1 2 3; do rm -rf /tmp/pg_probe_o; time ./bin/poregen gmove -k 6 $G/reads.slow5 $G/guppy_move.paf /tmp/pg_probe_o --fastq $G/read_0.fastq --kmer_file $G/kmer_file.txt 2>&1 | grep "gmove\] time: [0-9]"; done
echo "== the same with POREGEN_CLEAN_EXIT=1"
for i in 1 2; do rm -rf /tmp/pg_probe_o; time POREGEN_CLEAN_EXIT=1 ./bin/poregen gmove -k 6 $G/reads.slow5 $G/guppy_move.paf /tmp/pg_probe_o --fastq $G/read_0.fastq --kmer_file $G/kmer_file.txt 2>&1 | grep "gmove\] time: [0-9]"; done
} > $out/startup.txt 2>&1
python3 tools/cli_end_to_end.py 50000 > $out/e2e.txt 2>&1
cat $out/startup.txt; cat $out/e2e.txt
