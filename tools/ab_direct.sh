#!/bin/bash
# A/B run of library builds under variants/ with the direct integrator's benchmark
set -u
cd "$(dirname "$0")/.."
for lib in $(ls variants/*.so | sort); do
  MTSAMD_LIB=$PWD/$lib timeout -k 10 300 python3 tools/direct_bench.py --spp 64 2>gpurun_out/ab_direct.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); k = d['kernel_ms']
print(json.dumps({'lib': '$lib', 'mrays_per_s': round(d['mrays_per_s'], 1), 'shade': round(k['shade'], 1), 'extend': round(k['extend'], 1), 'shadow': round(k['shadow'], 1)}))" || { tail -5 gpurun_out/ab_direct.err; exit 1; }
done
