#!/bin/bash
# A/B run of library builds under variants/ on one workload with one forced shade variant:
#   tools/ab_variants.sh <workload> <MTSAMD_SHADE_VARIANT value or none>
set -u
cd "$(dirname "$0")/.."
for lib in $(ls variants/*.so | sort); do
  MTSAMD_LIB=$PWD/$lib MTSAMD_SHADE_VARIANT=${2:-none} timeout -k 10 300 python3 bench.py --workload ${1:-config3} --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>gpurun_out/ab_variants.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); k = d['kernel_ms_rank0']; n = d['steps']
print(json.dumps({'lib': '$lib', 'mrays_per_s': round(d['value'], 1), 'shade': round(k['shade']/n, 1), 'tail': round(k['tail']/n, 1)}))" || { tail -5 gpurun_out/ab_variants.err; exit 1; }
done
