#!/bin/bash
# Development aid: the 10M-triangle path-tracing rate with the built library and with each variant library named.
# usage: scripts/ab_flow.sh <spp> <variant names...>     (variants built by scripts/build_variant.sh; a name of the
# form base@N runs the built library with N samples per launch)
spp=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
run() { echo "== $1"; CPT_AMD_LIBRARY=$2 timeout -k 10 200 python3 scripts/bench_synth.py heightfield 10000000 3840 2160 $spp 5 4 0 ${3:-32} 2>&1 | grep -E "path tracing" ; }
run base ""
for v in "$@"; do
  case $v in
    base@*) run $v "" ${v#base@} ;;
    *) run $v $root/cuda-path-tracer_amd/variants/libcpt_$v.so ;;
  esac
done
