#!/bin/bash
# A/B of the ray hand-out: one range of rays per XCD (8) against a single range (1).
cd "$(dirname "$0")/.."
for r in 1 8 1 8; do
  MTSAMD_XCD_RANGES=$r timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/xcd.json 2>/dev/null || exit 1
  python3 -c "
import json; d=json.loads(open('gpurun_out/xcd.json').read()); k=d['kernel_ms_rank0']; print($r, round(d['value'],1), round(d['ms_per_step'],1), {x: round(k[x]/d['steps'],1) for x in ('extend','shade','shadow')})"
done
