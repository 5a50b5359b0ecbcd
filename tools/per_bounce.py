#!/usr/bin/env python3
"""Per-bounce kernel durations of one render pass from a rocprofv3 --kernel-trace CSV.

  rocprofv3 --kernel-trace -d gpurun_out/kt --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline
  python tools/per_bounce.py gpurun_out/kt

Prints, for the first bounces and in total, the duration of the closest-hit, shade and any-hit launches."""
import csv
import glob
import os
import sys

rows = []
for path in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
kinds = {"k_trace<0": "extend", "k_shade": "shade", "k_trace<1": "shadow", "k_shadow_accumulate": "accum"}
per = {k: [] for k in kinds.values()}
for s, e, name in rows:
    n = name.replace(" ", "").replace("(mtsamd::TraceMode)", "")
    for key, kind in kinds.items():
        if key in n:
            per[kind].append((e - s) * 1e-6)
nb = len(per["extend"])
print("bounces", nb, {k: round(sum(v), 2) for k, v in per.items()})
cum = 0.0
for i in range(min(nb, 40)):
    ext = per["extend"][i]
    sh = per["shade"][i] if i < len(per["shade"]) else 0
    sd = per["shadow"][i] if i < len(per["shadow"]) else 0
    cum += ext + sh + sd
    print("bounce %3d  extend %8.3f  shade %8.3f  shadow %8.3f  cumulative %8.2f ms" % (i, ext, sh, sd, cum))
tail = sum(per["extend"][40:]) + sum(per["shade"][40:]) + sum(per["shadow"][40:])
print("bounces 40.. : %.2f ms of kernel time in %d iterations" % (tail, max(0, nb - 40)))
