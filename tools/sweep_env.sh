#!/bin/bash
# One bench line per environment setting given on the command line: tools/sweep_env.sh NAME=value [NAME=value ...]
# (each argument is one run; the first and last runs are the defaults, to show the box's drift)
set -u
cd "$(dirname "$0")/.."
run() {
  env "$@" timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>gpurun_out/sweep_env.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); k = d['kernel_ms_rank0']; n = d['steps']
print(json.dumps({'setting': '$*', 'mrays_per_s': round(d['value'], 1), 'extend': round(k['extend']/n, 1), 'shade': round(k['shade']/n, 1), 'shadow': round(k['shadow']/n, 1), 'tail': round(k['tail']/n, 1)}))" || { tail -5 gpurun_out/sweep_env.err; exit 1; }
}
run MTSAMD_VERBOSE=0
for s in "$@"; do run "$s"; done
run MTSAMD_VERBOSE=0
