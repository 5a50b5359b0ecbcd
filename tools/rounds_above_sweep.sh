# Queue size from which the shade kernel sorts 4096 paths per workgroup (below it: one path per thread).
set -u
for e in "MTSAMD_SHADE_ROUNDS_ABOVE=4194304" "MTSAMD_SHADE_ROUNDS_ABOVE=1048576" "MTSAMD_SHADE_ROUNDS_ABOVE=2097152" "MTSAMD_SHADE_ROUNDS_ABOVE=8388608" "MTSAMD_SHADE_ROUNDS_ABOVE=4194304"; do
  env $e timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); k = d['kernel_ms_rank0']
print('$e: %.1f Mrays/s  ms/step %.1f  extend %.1f shade %.2f shadow %.1f' % (d['value'], d['ms_per_step'], k['extend']/d['steps'], k['shade']/d['steps'], k['shadow']/d['steps']))" || exit 1
done
