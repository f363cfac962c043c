#!/bin/bash
# PMC counters of the statistics kernel only (separate passes; see MI355X_MICROARCH.md rocprofv3 section)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --output-format csv -d gpurun_out/pmcs_$tag -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-lazy-extra --no-extras > /dev/null 2> gpurun_out/pmcs_$tag.err || { tail -5 gpurun_out/pmcs_$tag.err; exit 1; }
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmcs_*/')):
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k=row['Kernel_Name'].split('(')[0]
            acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
        for k,v in acc.items():
            if k.strip() in ('k_read_stats','k_walk','k_events','k_rank_count','k_rank_emit','k_gather'):
                print(k[:34], {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
