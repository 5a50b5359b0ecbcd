#!/bin/bash
# Two benchmark processes sharing one GPU: does overlapping two independent wavefront loops fill the gaps of one?
cd "$(dirname "$0")/.."
spp=${1:-64}
python3 bench.py --spp-step $spp --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/two_a.json 2>/dev/null &
pa=$!
python3 bench.py --spp-step $spp --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/two_b.json 2>/dev/null &
pb=$!
wait $pa; wait $pb
python3 - <<'PY'
import json
a = json.loads(open("gpurun_out/two_a.json").readline()); b = json.loads(open("gpurun_out/two_b.json").readline())
print("two processes: %.1f + %.1f = %.1f Mrays/s (ms/step %.1f, %.1f)" % (a["value"], b["value"], a["value"] + b["value"], a["ms_per_step"], b["ms_per_step"]))
PY
