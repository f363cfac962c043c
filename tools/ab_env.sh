#!/bin/bash
# quick_modes under several environments on one box: bash tools/ab_env.sh <tag> "NAME=VAL ..." "NAME=VAL ..." ...   (use X=1 for the default)
tag=$1; shift
i=0
for e in "$@"; do i=$((i+1)); echo "== $e"; env $e bash tools/quick_modes.sh ${tag}_$i; done
