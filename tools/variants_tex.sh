#!/bin/bash
# A/B of library builds under variants/ on the textured workload
set -u
cd "$(dirname "$0")/.."
for lib in $(ls variants/*.so | sort); do
  r=$(MTSAMD_LIB=$PWD/$lib timeout -k 10 200 python3 bench.py --workload config3tex:bilinear --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extra-workloads 2>gpurun_out/variants_err.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); k = d['kernel_ms_rank0']
print('%.1f Mrays/s  ms/step %.1f  extend %.1f shade %.1f shadow %.1f' % (d['value'], d['ms_per_step'], k['extend']/d['steps'], k['shade']/d['steps'], k['shadow']/d['steps']))") || { tail -3 gpurun_out/variants_err.txt; exit 1; }
  echo "$lib : $r"
done
