#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --output-format csv -d gpurun_out/pmc_$tag -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-lazy-extra --no-extras > gpurun_out/pmc_$tag.json 2> gpurun_out/pmc_$tag.err || exit 1
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmc_*/')):
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k=row['Kernel_Name'].split('(')[0]
            acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
        for k,v in acc.items():
            if 'k_' in k:
                print(d, k[:40], {c: (sum(x)/len(x)) for c,x in v.items()}, 'n=',len(next(iter(v.values()))))
PY
