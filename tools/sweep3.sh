#!/bin/bash
# Sweep over "<node stride> <tri stride>" lines (record strides in float4 units).
set -u
cd "$(dirname "$0")/.."
out=gpurun_out/sweep3_${1:-x}.txt
: > $out
while read -r ns ts; do
  [ -z "$ns" ] && continue
  r=$(MTSAMD_NODE_STRIDE=$ns MTSAMD_TRI_STRIDE=$ts timeout -k 10 120 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>gpurun_out/sweep_err.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); k = d['kernel_ms_rank0']
print('%.1f Mrays/s  ms/step %.1f  extend %.1f shade %.1f shadow %.1f' % (d['value'], d['ms_per_step'], k['extend']/d['steps'], k['shade']/d['steps'], k['shadow']/d['steps']))")
  rc=$?
  echo "node stride $ns tri stride $ts : $r" | tee -a $out
  if [ $rc -ne 0 ]; then tail -3 gpurun_out/sweep_err.txt; exit 1; fi
done
