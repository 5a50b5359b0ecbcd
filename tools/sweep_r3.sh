#!/bin/bash
# One bench summary line per environment setting (arguments: quoted "VAR=value VAR2=value" groups).
set -u
cd "$(dirname "$0")/.."
for setting in "$@"; do
  r=$(env $setting timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>gpurun_out/sweep_err.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); k = d['kernel_ms_rank0']; n = d['steps']
print('%.1f Mrays/s  ms/step %.1f  extend %.1f shade %.1f shadow %.1f film %.1f' % (d['value'], d['ms_per_step'], k['extend']/n, k['shade']/n, k['shadow']/n, k['film']/n))") || { tail -3 gpurun_out/sweep_err.txt; exit 1; }
  echo "$setting : $r"
done
