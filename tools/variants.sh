#!/bin/bash
# A/B run of library builds under variants/ (kernel tuning): prints one bench summary line per build.
set -u
cd "$(dirname "$0")/.."
out=gpurun_out/variants_${1:-x}.txt
: > $out
for lib in $(ls variants/*.so | sort); do
  [ -f "$lib" ] || continue
  r=$(MTSAMD_LIB=$PWD/$lib timeout -k 10 120 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extra-workloads 2>gpurun_out/variants_err.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); k = d['kernel_ms_rank0']
print('%.1f Mrays/s  ms/step %.1f  extend %.1f shade %.1f shadow %.1f film %.1f' % (d['value'], d['ms_per_step'], k['extend']/d['steps'], k['shade']/d['steps'], k['shadow']/d['steps'], k['film']/d['steps']))")
  rc=$?
  echo "$lib : $r" | tee -a $out
  if [ $rc -ne 0 ]; then tail -3 gpurun_out/variants_err.txt; exit 1; fi
done
