# Sweep of the SAH triangle cost used when the wide BVH is collapsed (MTSAMD_COST_TRI), headline workload.
set -u
for e in "MTSAMD_COST_TRI=0.5" "MTSAMD_COST_TRI=0.3" "MTSAMD_COST_TRI=0.8" "MTSAMD_COST_TRI=1.2" "MTSAMD_COST_TRI=0.2"; do
  env $e MTSAMD_COUNT_TRAVERSAL=0 timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); k = d['kernel_ms_rank0']; r = d.get('roofline', {}).get('per_ray', {})
print('$e: %.1f Mrays/s  ms/step %.1f  extend %.1f shade %.1f shadow %.1f  nodes/ray %.2f tris/ray %.2f  bvh nodes %d' % (d['value'], d['ms_per_step'], k['extend']/d['steps'], k['shade']/d['steps'], k['shadow']/d['steps'], r.get('node_visits', 0), r.get('tri_tests', 0), d['config']['bvh']['nodes']))" || exit 1
done
