#!/bin/bash
# Development aid: path-tracing rates of the built library and of variant libraries on the measurement scenes.
# usage: scripts/ab_scenes.sh <variant names...>
root=${GRAFT_REPO_ROOT:-$(pwd)}
for v in base "$@"; do
  L=""; [ "$v" != base ] && L=$root/cuda-path-tracer_amd/variants/libcpt_$v.so
  echo "== $v heightfield 10M"; CPT_AMD_LIBRARY=$L timeout -k 10 200 python3 scripts/bench_synth.py heightfield 10000000 3840 2160 64 5 4 0 32 2>&1 | grep -E "path tracing"
  echo "== $v soup 1M"; CPT_AMD_LIBRARY=$L timeout -k 10 200 python3 scripts/bench_synth.py soup 1000000 1920 1080 64 5 4 0 32 2>&1 | grep -E "path tracing"
  echo "== $v cornell"; CPT_AMD_LIBRARY=$L timeout -k 10 200 python3 scripts/prof_render.py 256 1024 8 3 0 128 2>&1 | grep -E "integrator"
done
