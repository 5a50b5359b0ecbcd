#!/bin/bash
# counters of the shading kernel on the textured workload next to the plain one (fabric bytes, vector ALU, waiting)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for w in ${WORKLOADS:-config3 config3tex:bilinear}; do
  tag=${TAGPREFIX:-}$(echo $w | tr ":" "_")
  dirs=""
  i=0
  for g in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
    out=gpurun_out/tex_${tag}_$i
    rm -rf "$out"
    timeout -s KILL 300 rocprofv3 --pmc $g -d "$out" --output-format csv -- python3 bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extra-workloads > gpurun_out/tex_${tag}_$i.log 2>&1 || { echo "pass $i failed"; tail -5 gpurun_out/tex_${tag}_$i.log; exit 1; }
    dirs="$dirs $out"
    i=$((i+1))
  done
  python3 tools/pmc_summary.py $dirs > gpurun_out/tex_${tag}_pmc.txt
  rm -rf $dirs
done
for f in gpurun_out/tex_${TAGPREFIX:-}*_pmc.txt; do echo $f; grep -A12 "^k_shade" $f | head -14; done
