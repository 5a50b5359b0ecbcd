#!/bin/bash
# Sweep of the traversal kernel's two scheduling thresholds on the benchmark scene (one bench line per setting):
#   MTSAMD_REFILL_BELOW  refill idle lanes when fewer than this many lanes hold a ray
#   MTSAMD_TRI_SHARE     run the triangle phase once this many 64ths of the busy lanes wait for it
set -u
cd "$(dirname "$0")/.."
run() {
  env "$@" timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>gpurun_out/sweep_trace.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); k = d['kernel_ms_rank0']; n = d['steps']
print(json.dumps({'setting': '$*', 'mrays_per_s': round(d['value'], 1), 'extend': round(k['extend']/n, 1), 'shadow': round(k['shadow']/n, 1)}))" || { tail -5 gpurun_out/sweep_trace.err; exit 1; }
}
run MTSAMD_VERBOSE=0
for r in 44 50 54 58 61; do run MTSAMD_REFILL_BELOW=$r; done
for t in 6 10 14 20 26; do run MTSAMD_TRI_SHARE=$t; done
run MTSAMD_VERBOSE=0
