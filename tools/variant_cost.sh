#!/bin/bash
# What the more general variants of the shading kernels cost on a scene that needs none of their features.
set -u
cd "$(dirname "$0")/.."
for v in none env env,rd mix env,rd,mix; do
  MTSAMD_SHADE_VARIANT=$v timeout -k 10 300 python3 bench.py --workload ${1:-config3} --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>gpurun_out/variant_cost.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); k = d['kernel_ms_rank0']; n = d['steps']
print(json.dumps({'variant': '$v', 'mrays_per_s': round(d['value'], 1), 'ms_per_step': round(d['ms_per_step'], 1), 'extend': round(k['extend']/n, 1), 'shade': round(k['shade']/n, 1), 'shadow': round(k['shadow']/n, 1), 'tail': round(k['tail']/n, 1)}))" || { tail -5 gpurun_out/variant_cost.err; exit 1; }
done
