#!/bin/bash
# Memory-system counter passes (latencies, stalls) over one short bench run each.
set -u
tag=${1:-x}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 1 --warmup 0 --spp-step 8 --no-cpu-baseline --no-roofline"
groups=(
  "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum GRBM_GUI_ACTIVE"
  "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum"
  "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum"
  "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_READ_sum TCC_CYCLE_sum"
  "SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INST_LEVEL_VMEM SQ_WAVES SQ_WAVE_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL"
)
i=0
for g in "${groups[@]}"; do
  out=gpurun_out/pmc_${tag}_$i
  rm -rf "$out"
  timeout -s KILL 240 rocprofv3 --pmc $g -d "$out" --output-format csv -- $BENCH > gpurun_out/pmc_${tag}_$i.log 2>&1
  rc=$?
  echo "pass $i rc=$rc"
  if [ $rc -ne 0 ]; then tail -5 gpurun_out/pmc_${tag}_$i.log; exit $rc; fi
  i=$((i+1))
done
python3 tools/pmc_summary.py gpurun_out/pmc_${tag}_* > gpurun_out/pmc_${tag}_summary.txt
rm -rf gpurun_out/pmc_${tag}_[0-9]
echo done
