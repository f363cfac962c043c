#!/bin/bash
# the product k_read_stats inside the stand-alone harness, with the timing-probe macros. usage (GPU box): bash tools/probe/run_real.sh
cd $GRAFT_REPO_ROOT/tools/probe
for v in "" "-DPG_PROBE_NO_SELECT" "-DPG_PROBE_NO_SELECT -DPG_PROBE_NO_BIN" "-DPG_STATS_WPB=4"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DPROBE_REAL_KERNEL $v -I../../include -I../../poregen_amd/csrc -o /tmp/sp_real stream_probe.hip 2>/dev/null || { echo "build failed: $v"; continue; }
  echo "== product kernel in the harness, flags: [$v]"
  PROBE_ONLY_REAL=1 /tmp/sp_real 400 | grep -E "k_read_stats|^G|^H"
done
