#!/usr/bin/env python3
"""Timeline of the kernels of a few ticks from a rocprofv3 --kernel-trace CSV: start offset, duration and the gap to
the kernel before, per stream order.  usage: tick_timeline.py <kernel_trace.csv> [first_row] [rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
t0 = int(rows[first]["Start_Timestamp"])
prev_end = None
for r in rows[first:first + n]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:9.1f} us  gap {gap:8.1f} us  grid {r.get('Grid_Size', '?'):>8}  {name}")
    prev_end = max(prev_end or e, e)
