#!/bin/bash
# bench lines (value, per-ray traversal work, kernel times) under different environment settings; each argument is one
# setting such as "MTSAMD_BVH_BUILDER=host" (several variables: separate them with spaces inside the quotes)
set -u
cd "$(dirname "$0")/.."
for setting in "$@"; do
  env $setting timeout -k 10 300 python3 bench.py --workload ${WORKLOAD:-config3} --steps ${STEPS:-2} --warmup 1 --no-cpu-baseline --no-extra-workloads 2>gpurun_out/ab_err.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']; k = d['kernel_ms_rank0']; p = r['per_ray']
print('%-40s %.1f Mrays/s extend %.1f shade %.1f shadow %.1f film %.1f | visits %.2f tris %.2f shadow visits %.2f tris %.2f | nodes %d depth %d' % ('$setting', d['value'], k['extend']/d['steps'], k['shade']/d['steps'], k['shadow']/d['steps'], k['film']/d['steps'], p['node_visits'], p['tri_tests'], p['shadow_node_visits'], p['shadow_tri_tests'], d['config'].get('bvh', {}).get('nodes', 0), d['config'].get('bvh', {}).get('depth', 0)))" || { tail -5 gpurun_out/ab_err.txt; exit 1; }
done
