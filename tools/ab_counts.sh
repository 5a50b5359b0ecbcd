#!/bin/bash
# traversal work per ray of the builds under variants/ and of the tree's library
set -u
cd "$(dirname "$0")/.."
for lib in variants/*.so mitsuba_amd/libmtsamd.so; do
  MTSAMD_LIB=$PWD/$lib timeout -k 10 200 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-workloads 2>gpurun_out/ab_err.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('$lib', d['value'], r['per_ray'], r.get('launch'))" || { tail -5 gpurun_out/ab_err.txt; exit 1; }
done
