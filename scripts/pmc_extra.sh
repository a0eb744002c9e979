#!/bin/bash
# Two more counter passes for one program: instruction-cache behaviour and the average number of memory instructions in flight.
# usage: scripts/pmc_extra.sh <outdir> <program and args...>
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
while read -r counters; do
  [ -z "$counters" ] && continue
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d "$out/pass$i" -- "$@" > "$out/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$out/pass$i.log"; exit 1; }
done <<'LIST'
SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY
SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_WAVE_CYCLES
LIST
echo "passes done: $i"
