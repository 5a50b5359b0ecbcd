#!/bin/bash
# Sweep of the BVH builder's triangle cost (env MTSAMD_COST_TRI); prints bench summary and per-ray traversal counts.
set -u
cd "$(dirname "$0")/.."
out=gpurun_out/sweep4_${1:-x}.txt
: > $out
while read -r ct; do
  [ -z "$ct" ] && continue
  r=$(MTSAMD_COST_TRI=$ct timeout -k 10 150 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>gpurun_out/sweep_err.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); k = d['kernel_ms_rank0']; p = d['roofline']['per_ray']
print('%.1f Mrays/s  ms/step %.1f  extend %.1f shade %.1f shadow %.1f | nodes/ray %.2f tris/ray %.2f shadow %.2f %.2f | bvh nodes %d' % (d['value'], d['ms_per_step'], k['extend']/d['steps'], k['shade']/d['steps'], k['shadow']/d['steps'], p['node_visits'], p['tri_tests'], p['shadow_node_visits'], p['shadow_tri_tests'], d['config']['bvh']['nodes']))")
  rc=$?
  echo "cost_tri=$ct : $r" | tee -a $out
  if [ $rc -ne 0 ]; then tail -3 gpurun_out/sweep_err.txt; exit 1; fi
done
