#!/bin/bash
# round 5 A/B on one box: parity tests of the changed paths first, then the three bench lines for the baseline build and the tree's build
# usage: bash tools/ab_r5.sh <tag> [tests...]
set -o pipefail
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
if [ $# -gt 0 ]; then
  timeout -k 10 900 python -m pytest "$@" -x -q -m gpu > $out/pytest.txt 2>&1 || { tail -30 $out/pytest.txt; exit 1; }
  tail -2 $out/pytest.txt
fi
bash tools/quick_modes.sh $tag/new && bash tools/quick_modes.sh $tag/base --lib build/r4/libpgmove.so && bash tools/quick_modes.sh $tag/new2 && bash tools/quick_modes.sh $tag/base2 --lib build/r4/libpgmove.so
