# A/B runs of the 10M-triangle workload under one environment switch at a time.
set -u
for e in "MTSAMD_XCD_RANGES=8" "MTSAMD_XCD_RANGES=1" "MTSAMD_TAIL_BELOW=0" "MTSAMD_XCD_RANGES=8"; do
  env $e timeout -k 10 300 python3 bench.py --workload config5 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); k = d['kernel_ms_rank0']
print('$e: %.1f Mrays/s  ms/step %.1f  extend %.1f shade %.1f shadow %.1f film %.1f tail %.1f' % (d['value'], d['ms_per_step'], k['extend']/d['steps'], k['shade']/d['steps'], k['shadow']/d['steps'], k['film']/d['steps'], k['tail']/d['steps']))" || exit 1
done
