#!/bin/bash
# End-of-milestone measurement on the GPU box: bench line, rocprofv3 kernel stats of the same bench command, and the
# hardware-counter passes of the default workload (one pass per counter group, each in a run of its own).
# Usage: tools/profile_round.sh <tag> [tests]; results under gpurun_out/ (copy what is to be kept into profiles/).
set -u
tag=${1:-x}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
if [ "${2:-}" = "tests" ]; then
  timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/${tag}_tests.log 2>&1 || { tail -20 gpurun_out/${tag}_tests.log; exit 1; }
  tail -1 gpurun_out/${tag}_tests.log
fi
rm -rf gpurun_out/${tag}_stats
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-extra-workloads > gpurun_out/${tag}_stats.log 2>&1 || { tail -5 gpurun_out/${tag}_stats.log; exit 1; }
cp $(find gpurun_out/${tag}_stats -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_kernel_stats.csv
rm -rf gpurun_out/${tag}_stats
BENCH="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extra-workloads"
groups=(
  "FETCH_SIZE"
  "WRITE_SIZE"
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"
  "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAVES"
)
dirs=""
i=0
for g in "${groups[@]}"; do
  out=gpurun_out/${tag}_pmc_$i
  rm -rf "$out"
  timeout -s KILL 300 rocprofv3 --pmc $g -d "$out" --output-format csv -- $BENCH > gpurun_out/${tag}_pmc_$i.log 2>&1 || { echo "pass $i failed"; tail -5 gpurun_out/${tag}_pmc_$i.log; exit 1; }
  dirs="$dirs $out"
  i=$((i+1))
done
python3 tools/pmc_summary.py $dirs > gpurun_out/${tag}_pmc_summary.txt
for k in extend shadow shade; do python3 tools/pmc_json.py $k 256 $dirs > gpurun_out/${tag}_${k}_pmc.json; done
rm -rf $dirs
# the bench line proper comes last, so that it finds the counter profile of this very build
cp gpurun_out/${tag}_extend_pmc.json profiles/${tag}_extend_pmc.json
timeout -k 10 400 python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { tail -5 gpurun_out/${tag}_bench.err; exit 1; }
cat gpurun_out/${tag}_bench.json
