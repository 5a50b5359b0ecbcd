#!/bin/bash
# default bench (main workload + the short runs of the other configurations) for every build under variants/ and the tree's
set -u
cd "$(dirname "$0")/.."
for lib in variants/*.so mitsuba_amd/libmtsamd.so; do
  MTSAMD_LIB=$PWD/$lib timeout -k 10 400 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>gpurun_out/ab_err.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); w = d['extra']['workloads']
print('%-28s config3 %.0f | %s' % ('$lib', d['value'], '  '.join('%s %.0f' % (k.split('_')[0], v['mrays_s']) for k, v in w.items())))" || { tail -5 gpurun_out/ab_err.txt; exit 1; }
done
