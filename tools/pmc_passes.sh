#!/bin/bash
# Hardware-counter passes over one short bench run each (one rocprofv3 --pmc run per counter group, never combined
# with tracing). Usage: tools/pmc_passes.sh <tag> ; results under gpurun_out/pmc_<tag>_<n>/ and a summary text file.
set -u
tag=${1:-x}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 1 --warmup 0 --spp-step 8 --no-cpu-baseline --no-roofline"
groups=(
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"
  "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR"
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
  "FETCH_SIZE"
  "WRITE_SIZE"
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
# the raw per-dispatch CSVs are large; keep only the summary
rm -rf gpurun_out/pmc_${tag}_[0-9]
echo done
