#!/bin/bash
# Does the size of the scratch ring limit the every-feature shade kernel? Same forced run with the runtime's limits raised.
set -u
cd "$(dirname "$0")/.."
run() {
  timeout -k 10 300 python3 bench.py --workload config3 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>gpurun_out/mix_cost.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); k = d['kernel_ms_rank0']; n = d['steps']
print(json.dumps({'case': '$1', 'mrays_per_s': round(d['value'], 1), 'shade': round(k['shade']/n, 1), 'tail': round(k['tail']/n, 1)}))" || { tail -5 gpurun_out/mix_cost.err; exit 1; }
}
export MTSAMD_SHADE_VARIANT=env,rd,mix
run default && HSA_SCRATCH_SINGLE_LIMIT=2000000000 run single_limit_2G && HSA_SCRATCH_SINGLE_LIMIT_ASYNC=4000000000 HSA_SCRATCH_SINGLE_LIMIT=2000000000 run both_limits
