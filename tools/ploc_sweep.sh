#!/bin/bash
# Tree quality against the neighbourhood radius of the GPU clustering (MTSAMD_PLOC_RADIUS) and the leaf cost (MTSAMD_COST_TRI).
cd "$(dirname "$0")/.."
for cfg in "${@:-16 0.5}"; do
  set -- $cfg
  MTSAMD_PLOC_RADIUS=$1 MTSAMD_COST_TRI=$2 timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/ploc.json 2>/dev/null || exit 1
  python3 -c "
import json; d=json.loads(open('gpurun_out/ploc.json').read()); k=d['kernel_ms_rank0']; r=d['roofline']['per_ray']; print('radius $1 cost_tri $2', round(d['value'],1), {x: round(k[x]/d['steps'],1) for x in ('extend','shadow')}, round(r['node_visits'],2), round(r['tri_tests'],2), d['config']['bvh']['build_s'], d['config']['bvh']['nodes'])"
done
