#!/bin/bash
# Shade time of the every-feature kernel variant: forced on the plain benchmark scene, and on the textured one.
set -u
cd "$(dirname "$0")/.."
run() {
  timeout -k 10 300 python3 bench.py --workload $1 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>gpurun_out/mix_cost.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); k = d['kernel_ms_rank0']; n = d['steps']
print(json.dumps({'workload': '$1', 'forced': '${MTSAMD_SHADE_VARIANT:-}', 'mrays_per_s': round(d['value'], 1), 'shade': round(k['shade']/n, 1), 'tail': round(k['tail']/n, 1)}))" || { tail -5 gpurun_out/mix_cost.err; exit 1; }
}
MTSAMD_SHADE_VARIANT=env,rd,mix run config3 && run config3tex:ewa
