#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, per counter, the sum over
dispatches, the dispatch count and the summed duration; for the bounce kernels also the first
(largest) dispatch of each batch.

    python tools/pmc_summary.py gpurun_out/pmc > profiles/r01_pmc_summary.txt
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("(anonymous namespace)::", "")
    return name.replace("void ", "").strip()


def main(root):
    for path in sorted(glob.glob(os.path.join(root, "*", "*_counter_collection.csv"))):
        print(f"# pass {os.path.basename(os.path.dirname(path))}")
        tot = defaultdict(lambda: defaultdict(float))
        calls = defaultdict(set)
        dur = defaultdict(dict)
        first = {}
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                d = int(row["Dispatch_Id"])
                tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
                calls[k].add(d)
                dur[k][d] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
                key = (k, row["Counter_Name"])
                if key not in first or d < first[key][0]:
                    first[key] = (d, float(row["Counter_Value"]))
        for k in sorted(tot, key=lambda k: -sum(dur[k].values())):
            if k.startswith("__amd") or "at::" in k:
                continue
            n = len(calls[k])
            print(f"{k}  dispatches={n}  total_us={sum(dur[k].values()) / 1e3:.1f}  vgpr/sgpr see kernel trace")
            for c in sorted(tot[k]):
                print(f"    {c:34s} sum={tot[k][c]:.6g}   first_dispatch={first[(k, c)][1]:.6g}")
        print()


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc")
