#!/bin/bash
# Three more counter passes for one program: how busy the vector memory pipeline of a CU is (address unit, L1 tags, data return)
# and how long an L1 miss waits for the L2.
# usage: scripts/pmc_l1.sh <outdir> <program and args...>      then: python3 scripts/pmc_summary.py <outdir> <kernel name part>
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
while read -r counters; do
  [ -z "$counters" ] && continue
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d "$out/pass$i" -- "$@" > "$out/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$out/pass$i.log"; exit 1; }
done <<'LIST'
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE
TA_FLAT_WAVEFRONTS_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum GRBM_GUI_ACTIVE
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_UTCL1_TRANSLATION_MISS_sum GRBM_GUI_ACTIVE
LIST
echo "passes done: $i"
