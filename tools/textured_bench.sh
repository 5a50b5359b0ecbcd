#!/bin/bash
# The benchmark scene with bitmap textures and a bump-mapped ground, one bench line per MIP filter, next to the plain scene.
set -u
cd "$(dirname "$0")/.."
for w in config3 config3tex:ewa config3tex:trilinear config3tex:bilinear; do
  timeout -k 10 300 python3 bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>gpurun_out/textured_bench.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); k = d['kernel_ms_rank0']; n = d['steps']
print(json.dumps({'workload': '$w', 'mrays_per_s': round(d['value'], 1), 'ms_per_step': round(d['ms_per_step'], 1), 'extend': round(k['extend']/n, 1), 'shade': round(k['shade']/n, 1), 'shadow': round(k['shadow']/n, 1), 'film': round(k['film']/n, 1), 'tail': round(k['tail']/n, 1)}))" || { tail -5 gpurun_out/textured_bench.err; exit 1; }
done
