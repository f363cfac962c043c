#!/bin/bash
# PMC counter sets of the k = 9 step's kernels (one rocprofv3 pass per set): bash tools/pmc_k9.sh <tag> "<kernel name substrings>" [--lib ...]
tag=$1; names="$2"; shift; shift
out=gpurun_out/$tag; mkdir -p $out
R=$PWD; cd /tmp && export TMPDIR=/tmp && cd $R
i=0
for pass in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum" "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_TAG_STALL_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $pass --output-format csv -d $out/pmc_$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-lazy-extra --no-extras --one-stream --kind dna_r10 --k 9 --sample-limit 1000 "$@" > /dev/null 2> $out/pmc_$i.err || { echo "PMC pass $i ($pass) FAILED"; tail -3 $out/pmc_$i.err; }
done
python3 - $out "$names" > $out/pmc_summary.txt <<'PY'
import csv, glob, collections, sys
want = sys.argv[2].split()
for d in sorted(glob.glob(sys.argv[1] + '/pmc_*/')):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[row['Kernel_Name'].split('(')[0].strip()][row['Counter_Name']].append(float(row['Counter_Value']))
        for k, v in acc.items():
            if any(x in k for x in want): print(k, {c: round(sum(x) / len(x)) for c, x in v.items()})
PY
cat $out/pmc_summary.txt
