#!/usr/bin/env python3
"""Counters of the first bounces of one batch for one kernel, from tools/pmc_passes.sh output.

    python tools/pmc_bounces.py gpurun_out/pmc k_extend_walk [n_bounces] [batch]
"""
import collections, csv, glob, os, sys
root, kern = sys.argv[1], sys.argv[2]
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 4
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 4
for path in sorted(glob.glob(os.path.join(root, "*", "*_counter_collection.csv"))):
    byd = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if kern in r["Kernel_Name"]:
            d = byd.setdefault(int(r["Dispatch_Id"]), {})
            d[r["Counter_Name"]] = float(r["Counter_Value"])
            d["dur_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    ds = sorted(byd)
    per_batch = 50
    for k in range(nb):
        i = batch * per_batch + k
        if i >= len(ds):
            break
        d = byd[ds[i]]
        extra = ""
        if "SQ_THREAD_CYCLES_VALU" in d and d.get("SQ_ACTIVE_INST_VALU"):
            extra = f"  lanes_active={d['SQ_THREAD_CYCLES_VALU'] / (64 * d['SQ_ACTIVE_INST_VALU']):.2f}"
        print(os.path.basename(os.path.dirname(path)), "bounce", k, " ".join(f"{a.replace('SQ_', '').replace('_sum', '')}={b:.3g}" for a, b in sorted(d.items())) + extra)
