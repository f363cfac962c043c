#!/bin/bash
# End-of-round validation in one GPU call: the round artefacts (GPU suite, bench, kernel trace, PMC passes), the randomised
# parity runs (library and CLI) and the one-rank RCCL step. usage: bash tools/final_check.sh <tag>
set -o pipefail
tag=${1:-final}; out=gpurun_out/$tag; mkdir -p $out
bash tools/round_artifacts.sh $tag || exit 1
timeout -k 10 500 python3 tools/fuzz_gpu.py ${2:-250} 4242 > $out/fuzz_gpu.txt 2>&1 || { tail -20 $out/fuzz_gpu.txt; exit 1; }
tail -1 $out/fuzz_gpu.txt
timeout -k 10 400 python3 tools/fuzz_cli.py ${3:-60} 4242 > $out/fuzz_cli.txt 2>&1 || { tail -20 $out/fuzz_cli.txt; exit 1; }
tail -1 $out/fuzz_cli.txt
bash tools/ab_dist.sh $tag/dist nopytest
