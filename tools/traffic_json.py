#!/usr/bin/env python3
"""HBM-side traffic of the closest-hit traversal launches from two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE).

  python tools/traffic_json.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> [spp per pass] > profiles/rNN_traffic.json

The counters are in KiB. On gfx950 FETCH_SIZE tallies 128-byte fabric requests at 64 bytes, so it is doubled; that
calibration was made on wide streaming reads and is only an estimate for the scattered 16-byte loads of a traversal.
Dispatches are attributed to the closest-hit instantiation of k_trace by its template argument in the kernel name."""
import csv
import glob
import json
import os
import sys


def collect(d, counter):
    tot, n = 0.0, 0
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"]
                if row["Counter_Name"] == counter and "k_trace<0" in k.replace(" ", "").replace("(mtsamd::TraceMode)", ""):
                    tot += float(row["Counter_Value"]); n += 1
    return tot, n


fetch, nf = collect(sys.argv[1], "FETCH_SIZE")
write, nw = collect(sys.argv[2], "WRITE_SIZE")
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 256
out = {"kernel": "k_trace<TRACE_CLOSEST>", "launches": nf, "spp_step": spp,
       "fetch_kib_raw": fetch, "write_kib": write,
       "traffic_bytes_per_launch": (2.0 * fetch + write) * 1024.0 / max(nf, 1),
       "note": "FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B); bench.py --steps 1 --warmup 0, one pass of spp_step samples per pixel"}
print(json.dumps(out))
