#!/bin/bash
# Throughput against the path count below which one launch finishes the pass (MTSAMD_TAIL_BELOW).
cd "$(dirname "$0")/.."
for t in ${@:-0 65536 262144 1048576 4194304}; do
  MTSAMD_TAIL_BELOW=$t timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/ts.json 2>/dev/null || exit 1
  python3 -c "
import json; d=json.loads(open('gpurun_out/ts.json').read()); k=d['kernel_ms_rank0']; print($t, round(d['value'],1), round(d['ms_per_step'],1), {x: round(k[x]/d['steps'],1) for x in ('extend','shade','shadow')}, k['bounce_iterations']//d['steps'])"
done
