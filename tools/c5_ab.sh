# A/B runs of the 10M-triangle workload under one environment switch at a time.
set -u
for e in "MTSAMD_CHUNK_MAX=512" "MTSAMD_CHUNK_MAX=128" "MTSAMD_CHUNK_MAX=64" "MTSAMD_CHUNK_MAX=2048" "MTSAMD_TOP_NODES=9" "MTSAMD_REFILL_BELOW=48" "MTSAMD_TRI_SHARE=8" "MTSAMD_TRI_SHARE=32"; do
  env $e timeout -k 10 300 python3 bench.py --workload config5 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); k = d['kernel_ms_rank0']
print('$e: %.1f Mrays/s  ms/step %.1f  extend %.1f shade %.1f shadow %.1f film %.1f tail %.1f' % (d['value'], d['ms_per_step'], k['extend']/d['steps'], k['shade']/d['steps'], k['shadow']/d['steps'], k['film']/d['steps'], k['tail']/d['steps']))" || exit 1
done
