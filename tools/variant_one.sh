#!/bin/bash
# A/B build that differs in ONE source file: bash tools/variant_one.sh <name> <file stem, e.g. pg_place> "<-D... flags>"  -> build/<name>/libpgmove.so
# (the other objects are the default build's: run `make` first). Use with bench.py --lib / tools/quick_modes.sh <tag> --lib build/<name>/libpgmove.so
set -e
name=$1; stem=$2; extra=$3
mkdir -p build/$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-result $extra -c -o build/$name/$stem.o poregen_amd/csrc/$stem.hip
objs=""
for f in pg_kernels pg_place pg_api pg_model pg_job pg_text; do if [ $f = $stem ]; then objs="$objs build/$name/$f.o"; else objs="$objs build/$f.o"; fi; done
g++ -shared -o build/$name/libpgmove.so $objs -Wl,--allow-shlib-undefined -ldl -lpthread
