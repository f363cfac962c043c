#!/bin/bash
# SQ counters of k_slot_model at k = 9 (tools/model_k9.py): what is the one-wave variant bound by?  -> gpurun_out/pm_*/, summary on stdout
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --output-format csv -d gpurun_out/pm_$tag -- python3 tools/model_k9.py > gpurun_out/pm_$tag.out 2> gpurun_out/pm_$tag.err || { echo "PMC pass $tag FAILED:" >&2; tail -5 gpurun_out/pm_$tag.err >&2; exit 1; }
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pm_*/')):
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k=row['Kernel_Name'].split('(')[0].strip()
            acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
        for k,v in acc.items():
            if 'slot_model' in k: print(k[:60], {c: round(sum(x)/len(x)) for c,x in v.items()}, "launches", len(next(iter(v.values()))))
PY
