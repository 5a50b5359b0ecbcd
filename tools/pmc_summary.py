#!/usr/bin/env python3
"""Sums rocprofv3 counter-collection CSVs per (kernel, counter).

  python tools/pmc_summary.py gpurun_out/pmc_* > profiles/rNN_pmc_summary.txt

Every directory given is searched for *counter_collection.csv; kernel names are shortened to the text before the
first '(' or '<'. Values are summed over dispatches; the dispatch count is printed beside them."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    flat = name.replace(" ", "").replace("(mtsamd::TraceMode)", "")
    if "k_trace<0" in flat:
        return "k_trace<closest-hit>"
    if "k_trace<1" in flat:
        return "k_trace<any-hit>"
    for ch in "(<":
        i = name.find(ch)
        if i > 0:
            name = name[:i]
    return name.split("::")[-1].strip()


def main():
    sums = defaultdict(float)
    disp = defaultdict(set)
    for d in sys.argv[1:]:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    k = short(row["Kernel_Name"])
                    sums[(k, row["Counter_Name"])] += float(row["Counter_Value"])
                    disp[(k, row["Counter_Name"])].add((path, row["Dispatch_Id"]))
    kernels = sorted({k for k, _ in sums})
    for k in kernels:
        print(k)
        for (kk, c), v in sorted(sums.items()):
            if kk == k:
                print("    %-40s %20.0f   (%d dispatches)" % (c, v, len(disp[(kk, c)])))


if __name__ == "__main__":
    main()
