#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d gpurun_out/pmc_probe -- tools/probe/stream_probe > /dev/null 2> gpurun_out/pmc_probe.err || { tail -5 gpurun_out/pmc_probe.err; exit 1; }
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/pmc_probe/**/*counter_collection.csv', recursive=True):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        k=row['Kernel_Name'].split('(')[0]
        acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
    for k,v in acc.items():
        if 'chunk_rec' in k or 'chunk_500' in k or 'chunk_x<7>' in k: print(k[:30], {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
