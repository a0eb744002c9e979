#!/bin/bash
# Development aid: a copy of libcpt_amd.so whose render.hip was compiled with extra flags (e.g. -DFLOW_NUM=4), for A/B
# runs on the GPU box with CPT_AMD_LIBRARY=<that file>.  Needs the normal build to exist (its other objects are reused).
# usage: scripts/build_variant.sh <name> <extra compiler flags...>      -> cuda-path-tracer_amd/variants/libcpt_<name>.so
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/cuda-path-tracer_amd/csrc
out=$root/cuda-path-tracer_amd/variants
mkdir -p "$out" "$src/build/variant_$name"
hipcc -std=c++20 -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fPIC -Wall -Wno-unused-function "$@" -c "$src/render.hip" -o "$src/build/variant_$name/render.o"
objs=$(ls "$src"/build/*.o "$src"/build/host/*.o | grep -v "/render.o" | grep -v "/main.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libcpt_$name.so" "$src/build/variant_$name/render.o" $objs -lz -ldl -lpthread
echo "built $out/libcpt_$name.so"
