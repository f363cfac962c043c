#!/bin/bash
# parallel build of A/B variants: bash tools/variants.sh name1 "-DX=1" name2 "-DY=2 -DZ" ...   -> build/<name>/libpgmove.so
set -e
HIPCC=/opt/rocm/bin/hipcc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-result"
while [ $# -gt 0 ]; do
  name=$1; extra=$2; shift 2
  mkdir -p build/$name
  for f in pg_kernels pg_place pg_api pg_model pg_job pg_text; do
    $HIPCC $FLAGS $extra -c -o build/$name/$f.o poregen_amd/csrc/$f.hip &
  done
  wait
  g++ -shared -o build/$name/libpgmove.so build/$name/pg_kernels.o build/$name/pg_place.o build/$name/pg_api.o build/$name/pg_model.o build/$name/pg_job.o build/$name/pg_text.o -Wl,--allow-shlib-undefined -ldl -lpthread
  echo built build/$name/libpgmove.so
done
