#!/bin/bash
# End-of-milestone measurement on the GPU box: parity tests, bench line, rocprofv3 kernel stats of the same bench
# command, and the two HBM-traffic counter passes. Usage: tools/profile_round.sh <tag>; results under gpurun_out/.
set -u
tag=${1:-x}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout -k 10 600 python3 -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/${tag}_tests.log 2>&1 || { tail -20 gpurun_out/${tag}_tests.log; exit 1; }
tail -1 gpurun_out/${tag}_tests.log
timeout -k 10 400 python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { tail -5 gpurun_out/${tag}_bench.err; exit 1; }
rm -rf gpurun_out/${tag}_stats
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_stats --output-format csv -- python3 bench.py --no-cpu-baseline > gpurun_out/${tag}_stats.log 2>&1 || { tail -5 gpurun_out/${tag}_stats.log; exit 1; }
cp $(find gpurun_out/${tag}_stats -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_kernel_stats.csv
rm -rf gpurun_out/${tag}_stats
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/${tag}_pmc_$c
  timeout -s KILL 300 rocprofv3 --pmc $c -d gpurun_out/${tag}_pmc_$c --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > gpurun_out/${tag}_pmc_$c.log 2>&1 || { tail -5 gpurun_out/${tag}_pmc_$c.log; exit 1; }
done
python3 tools/pmc_summary.py gpurun_out/${tag}_pmc_FETCH_SIZE gpurun_out/${tag}_pmc_WRITE_SIZE > gpurun_out/${tag}_traffic.txt
python3 tools/traffic_json.py gpurun_out/${tag}_pmc_FETCH_SIZE gpurun_out/${tag}_pmc_WRITE_SIZE > gpurun_out/${tag}_traffic.json
rm -rf gpurun_out/${tag}_pmc_FETCH_SIZE gpurun_out/${tag}_pmc_WRITE_SIZE
cat gpurun_out/${tag}_bench.json
