#!/bin/bash
# Counter profile of the closest-hit kernel on the 10M-triangle scene (structure larger than the Infinity Cache, so what
# leaves the L2s is HBM traffic): kernel stats, one rocprofv3 --pmc pass per counter group, then the bench line that
# reads the profile. Usage: tools/profile_config5.sh <tag>; results under gpurun_out/ and profiles/<tag>_c5_extend_pmc.json.
set -u
tag=${1:-x}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
W="--workload config5 --no-cpu-baseline"
rm -rf gpurun_out/${tag}_c5_stats
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_c5_stats --output-format csv -- python3 bench.py $W --steps 2 --warmup 1 --no-roofline > gpurun_out/${tag}_c5_stats.log 2>&1 || { tail -5 gpurun_out/${tag}_c5_stats.log; exit 1; }
cp $(find gpurun_out/${tag}_c5_stats -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_c5_kernel_stats.csv
rm -rf gpurun_out/${tag}_c5_stats
groups=(
  "FETCH_SIZE"
  "WRITE_SIZE"
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"
  "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAVES"
)
dirs=""
i=0
for g in "${groups[@]}"; do
  out=gpurun_out/${tag}_c5_pmc_$i
  rm -rf "$out"
  timeout -s KILL 300 rocprofv3 --pmc $g -d "$out" --output-format csv -- python3 bench.py $W --steps 1 --warmup 0 --no-roofline > gpurun_out/${tag}_c5_pmc_$i.log 2>&1 || { echo "pass $i failed"; tail -5 gpurun_out/${tag}_c5_pmc_$i.log; exit 1; }
  dirs="$dirs $out"
  i=$((i+1))
  echo "pass $i done"
done
python3 tools/pmc_summary.py $dirs > gpurun_out/${tag}_c5_pmc_summary.txt
python3 tools/pmc_json.py extend@config5 256 $dirs > gpurun_out/${tag}_c5_extend_pmc.json
rm -rf $dirs
cp gpurun_out/${tag}_c5_extend_pmc.json profiles/${tag}_c5_extend_pmc.json
timeout -k 10 400 python3 bench.py $W --steps 2 --warmup 1 > gpurun_out/${tag}_c5_bench.json 2> gpurun_out/${tag}_c5_bench.err || { tail -5 gpurun_out/${tag}_c5_bench.err; exit 1; }
cat gpurun_out/${tag}_c5_bench.json
