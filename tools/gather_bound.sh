#!/bin/bash
# What bounds k_gather_chunks at k = 9: the product kernel built without window reads / without sample stores /
# without the FP64 division (tools: make variant NAME=gc_* EXTRA=-DPG_PROBE_GC_*), one box, plus one PMC pass of the default build.
# Round 6: the PG_PROBE_GC_* branches left the product sources; apply tools/probe/r05_timing_probes.patch (patch -p0 -R style: see its header) to a scratch copy first.
#   bash tools/gather_bound.sh <tag>    -> gpurun_out/<tag>/summary.txt
set -o pipefail
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
R=$PWD; cd /tmp && export TMPDIR=/tmp && cd $R
common="--no-cpu-baseline --no-lazy-extra --no-extras --steps 10 --warmup 3 --one-stream --kind dna_r10 --k 9 --sample-limit 1000"
for v in default gc_nr gc_ns gc_nrnd gc_nsnd "$@"; do
  lib=""; [ $v != default ] && lib="--lib build/$v/libpgmove.so"
  [ $v != default ] && [ ! -f build/$v/libpgmove.so ] && continue
  timeout -k 10 300 python3 bench.py $common $lib > $out/${v}.json 2> $out/${v}.err || { tail -5 $out/${v}.err; exit 1; }
done
python3 - $out default gc_nr gc_ns gc_nrnd gc_nsnd "$@" > $out/summary.txt <<'PY'
import json, sys, os
for v in sys.argv[2:]:
    f = f"{sys.argv[1]}/{v}.json"
    if not os.path.exists(f): continue
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(v.ljust(10), "%.4f ms " % d["ms_per_step"], " ".join("%s %.1f" % (k, x * 1e3) for k, x in d["kernels_ms_per_step"].items()))
PY
cat $out/summary.txt
for pass in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum" "TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  t=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $pass --output-format csv -d $out/pmc_$t -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-lazy-extra --no-extras --one-stream --kind dna_r10 --k 9 --sample-limit 1000 > /dev/null 2> $out/pmc_$t.err || { echo "PMC pass $t FAILED" >> $out/summary.txt; tail -3 $out/pmc_$t.err >> $out/summary.txt; }
done
python3 - $out >> $out/summary.txt <<'PY'
import csv, glob, collections, sys
for d in sorted(glob.glob(sys.argv[1] + '/pmc_*/')):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[row['Kernel_Name'].split('(')[0].strip()][row['Counter_Name']].append(float(row['Counter_Value']))
        for k, v in acc.items():
            if 'k_gather' in k or 'k_part_scatter' in k or 'k_region_place' in k or 'k_events' in k:
                print(k, {c: round(sum(x) / len(x)) for c, x in v.items()})
PY
tail -20 $out/summary.txt
