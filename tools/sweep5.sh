#!/bin/bash
# Sweep "<lib> <refill below> <tri share>" lines.
set -u
cd "$(dirname "$0")/.."
out=gpurun_out/sweep5_${1:-x}.txt
: > $out
while read -r lib refill tri; do
  [ -z "$lib" ] && continue
  r=$(MTSAMD_LIB=$PWD/$lib MTSAMD_REFILL_BELOW=$refill MTSAMD_TRI_SHARE=$tri timeout -k 10 120 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>gpurun_out/sweep_err.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); k = d['kernel_ms_rank0']
print('%.1f Mrays/s  ms/step %.1f  extend %.1f shade %.1f shadow %.1f' % (d['value'], d['ms_per_step'], k['extend']/d['steps'], k['shade']/d['steps'], k['shadow']/d['steps']))")
  rc=$?
  echo "$lib refill=$refill tri=$tri : $r" | tee -a $out
  if [ $rc -ne 0 ]; then tail -3 gpurun_out/sweep_err.txt; exit 1; fi
done
