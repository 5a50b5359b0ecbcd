# Ray hand-out per XCD (8 ranges) against one shared range, on every bench workload.
set -u
for w in config3 cornell veach; do
for e in "MTSAMD_XCD_RANGES=8" "MTSAMD_XCD_RANGES=1" "MTSAMD_XCD_RANGES=8" "MTSAMD_XCD_RANGES=1"; do
  env $e timeout -k 10 300 python3 bench.py --workload $w --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); k = d['kernel_ms_rank0']
print('$w $e: %.1f Mrays/s  ms/step %.1f  extend %.1f shade %.1f shadow %.1f' % (d['value'], d['ms_per_step'], k['extend']/d['steps'], k['shade']/d['steps'], k['shadow']/d['steps']))" || exit 1
done
done
