#!/bin/bash
# PMC counters of chosen kernels under bench.py: bash tools/pmc_kernel.sh "<kernel name substrings, space separated>" [bench args...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
names="$1"; shift
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR GRBM_GUI_ACTIVE" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do  # every set fits ONE hardware pass. FETCH_SIZE + WRITE_SIZE together do not: rocprofv3 aborts with signal 6, "Request exceeds the capabilities of the hardware" (counter over-subscription, not a product fault); tools/round_artifacts.sh collects them in separate passes
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --output-format csv -d gpurun_out/pk_$tag -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-lazy-extra --no-extras "$@" > /dev/null 2> gpurun_out/pk_$tag.err || { echo "PMC pass $tag FAILED:" >&2; tail -5 gpurun_out/pk_$tag.err >&2; exit 1; }
done
python3 - "$names" <<'PY'
import csv, glob, collections, sys
want = sys.argv[1].split()
for d in sorted(glob.glob('gpurun_out/pk_*/')):
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k=row['Kernel_Name'].split('(')[0].strip()
            acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
        for k,v in acc.items():
            if any(w in k for w in want): print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
