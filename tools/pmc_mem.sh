#!/bin/bash
# Memory-pipeline counters of the traversal kernels (texture addresser, L1 vector cache, data return), one rocprofv3
# pass per counter group. Usage: tools/pmc_mem.sh <tag>; summary in gpurun_out/<tag>_mem_pmc.txt
set -u
tag=${1:-x}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extra-workloads"
groups=(
  "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TD_TD_BUSY_sum TD_TC_STALL_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
  "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"
  "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_LATENCY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum"
  "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
)
dirs=""
i=0
for g in "${groups[@]}"; do
  out=gpurun_out/${tag}_mem_$i
  rm -rf "$out"
  timeout -s KILL 300 rocprofv3 --pmc $g -d "$out" --output-format csv -- $BENCH > gpurun_out/${tag}_mem_$i.log 2>&1 || { echo "pass $i failed"; tail -5 gpurun_out/${tag}_mem_$i.log; exit 1; }
  dirs="$dirs $out"
  i=$((i+1))
done
python3 tools/pmc_summary.py $dirs > gpurun_out/${tag}_mem_pmc.txt
rm -rf $dirs
cat gpurun_out/${tag}_mem_pmc.txt
