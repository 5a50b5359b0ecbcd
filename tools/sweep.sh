#!/bin/bash
# Parameter sweep of the traversal launch shape: each line "<block> <top nodes> <refill below> <tri share>".
set -u
cd "$(dirname "$0")/.."
out=gpurun_out/sweep_${1:-x}.txt
: > $out
while read -r blk top refill tri chunk; do
  export MTSAMD_CHUNK_MAX=${chunk:-512}
  [ -z "$blk" ] && continue
  r=$(MTSAMD_TRACE_BLOCK=$blk MTSAMD_TOP_NODES=$top MTSAMD_REFILL_BELOW=$refill MTSAMD_TRI_SHARE=$tri MTSAMD_VERBOSE=1 timeout -k 10 120 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>gpurun_out/sweep_err.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); k = d['kernel_ms_rank0']
print('%.1f Mrays/s  ms/step %.1f  extend %.1f shade %.1f shadow %.1f' % (d['value'], d['ms_per_step'], k['extend']/d['steps'], k['shade']/d['steps'], k['shadow']/d['steps']))")
  rc=$?
  echo "block=$blk top=$top refill=$refill tri=$tri chunk=${chunk:-512} : $r $(grep -m1 'blocks/CU' gpurun_out/sweep_err.txt | sed 's/mtsamd: //')" | tee -a $out
  if [ $rc -ne 0 ]; then tail -3 gpurun_out/sweep_err.txt; exit 1; fi
done
