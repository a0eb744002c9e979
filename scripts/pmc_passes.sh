#!/bin/bash
# Counter passes for one program, one rocprofv3 run per pass (the tool does not split passes).
# usage: scripts/pmc_passes.sh <outdir> <program and args...>
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
while read -r counters; do
  [ -z "$counters" ] && continue
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d "$out/pass$i" -- "$@" > "$out/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$out/pass$i.log"; exit 1; }
done <<'LIST'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
FETCH_SIZE
WRITE_SIZE
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE
LIST
echo "passes done: $i"
