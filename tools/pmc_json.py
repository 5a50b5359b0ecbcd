#!/usr/bin/env python3
"""Per-launch hardware-counter figures of one kernel from rocprofv3 --pmc passes, as JSON for bench.py.

  python tools/pmc_json.py <kernel key>[@<workload>] <spp per step> <dir> [<dir> ...] > profiles/rNN_extend_pmc.json

<kernel key>: "extend" (closest-hit instantiation of k_trace), "shadow" (any-hit), "shade" (k_shade).
Every directory is one pass; counters are summed over the dispatches of the kernel and divided by their number.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-byte fabric requests at 64 bytes, so it is doubled
(calibrated on wide streaming reads; for scattered 16-byte loads it is an estimate). SQ_* cycle counters are in
quad-cycles per wave. The hash identifies the device sources the library was built from."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def matches(kernel, key):
    k = kernel.replace(" ", "").replace("(mtsamd::TraceMode)", "")
    if key == "extend":
        return "k_trace<0" in k
    if key == "shadow":
        return "k_trace<1" in k
    if key == "shade":
        return "k_shade<" in k
    return key in k


def main():
    key, spp = sys.argv[1], int(sys.argv[2])
    key, _, workload = key.partition("@")
    sums, counts = defaultdict(float), defaultdict(int)
    for d in sys.argv[3:]:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    if matches(row["Kernel_Name"], key):
                        sums[row["Counter_Name"]] += float(row["Counter_Value"])
                        counts[row["Counter_Name"]] += 1
    per = {c: sums[c] / counts[c] for c in sums}
    import bench
    out = {"kernel": key, "workload": workload or "config3", "spp_step": spp, "source_hash": bench.kernel_source_hash(), "launches": max(counts.values()) if counts else 0,
           "per_launch": per}
    if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
        out["fabric_bytes_per_launch"] = (2.0 * per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024.0
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in per and "TCP_TCC_READ_REQ_sum" in per:
        out["l1_hit"] = 1.0 - per["TCP_TCC_READ_REQ_sum"] / per["TCP_TOTAL_CACHE_ACCESSES_sum"]
    if "TCC_HIT_sum" in per and "TCC_MISS_sum" in per:
        out["l2_hit"] = per["TCC_HIT_sum"] / (per["TCC_HIT_sum"] + per["TCC_MISS_sum"])
    if "SQ_THREAD_CYCLES_VALU" in per and "SQ_ACTIVE_INST_VALU" in per and per["SQ_ACTIVE_INST_VALU"] > 0:
        out["valu_lane_utilisation"] = per["SQ_THREAD_CYCLES_VALU"] / (64.0 * per["SQ_ACTIVE_INST_VALU"])
    if "SQ_WAIT_ANY" in per and "SQ_WAVE_CYCLES" in per:
        out["wait_fraction"] = per["SQ_WAIT_ANY"] / per["SQ_WAVE_CYCLES"]
    if "SQ_INSTS_VALU" in per:
        # a SIMD issues one wave64 vector instruction per 2 cycles at best: 1 024 SIMDs at 2.4 GHz
        out["valu_issue_bound_ms"] = per["SQ_INSTS_VALU"] * 2.0 / (1024 * 2.4e9) * 1e3
    print(json.dumps(out))


if __name__ == "__main__":
    main()
