# A/B of the shade kernel with and without its sweeps (MTSAMD_SHADE_ROUNDS=1: one path per thread, no large sort domain).
set -u
for r in 1 16 1 16; do
  MTSAMD_SHADE_ROUNDS=$r timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); k = d['kernel_ms_rank0']
print('rounds $r: %.1f Mrays/s  ms/step %.1f  extend %.1f shade %.1f shadow %.1f' % (d['value'], d['ms_per_step'], k['extend']/d['steps'], k['shade']/d['steps'], k['shadow']/d['steps']))" || exit 1
done
