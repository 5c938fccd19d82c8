#!/usr/bin/env python3
"""Per-dispatch durations of one batch from a rocprofv3 kernel trace (csv) taken with one batch in flight.

    python tools/solo_trace.py gpurun_out/solo/solo_kernel_trace.csv [first_n]
"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first_n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_raygen(")]
st = idx[1] if len(idx) > 1 else idx[0]
end = idx[2] if len(idx) > 2 else len(rows)
t0 = int(rows[st]["Start_Timestamp"])
name = lambda r: re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
for r in rows[st:st + first_n]:
    print(f"{name(r):24s} start {(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  dur {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us")
tot = {}
for r in rows[st:end]:
    tot[name(r)] = tot.get(name(r), 0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print({k: round(v, 1) for k, v in tot.items()}, "batch wall us", (int(rows[end - 1]["End_Timestamp"]) - t0) / 1e3)
