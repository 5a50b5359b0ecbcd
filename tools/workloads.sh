set -u
timeout -k 10 400 python3 -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r16_tests.log 2>&1 || { tail -20 gpurun_out/r16_tests.log; exit 1; }
tail -1 gpurun_out/r16_tests.log
for w in config3 cornell veach config5; do
  timeout -k 10 300 python3 bench.py --workload $w --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); k = d['kernel_ms_rank0']
print('$w: %.1f Mrays/s  ms/step %.1f  extend %.1f shade %.1f shadow %.1f' % (d['value'], d['ms_per_step'], k['extend']/d['steps'], k['shade']/d['steps'], k['shadow']/d['steps']))" || exit 1
done
