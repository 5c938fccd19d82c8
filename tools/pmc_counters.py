#!/usr/bin/env python3
"""Per-ray hardware-counter figures of the closest-hit kernels (summed over both) from the passes of tools/pmc_passes.sh.

Every pass is one rocprofv3 --pmc run of the same short bench.py command (kernels run one at a time
under the counters); the bench line in each pass's log gives the rays that run traced.  Output: one
JSON object with instructions, lane utilisation, waiting share, cache hit rates and HBM bytes per ray,
which bench.py reads to put counter-based fractions beside the algorithmic one.

FETCH_SIZE and WRITE_SIZE count 1024 bytes per unit as rocprofv3 reports them; on gfx950 FETCH_SIZE
tallies a 128-byte request as 64 bytes, so it is doubled, which makes the figure an upper bound for
this kernel's scattered 64-byte node fetches.

    python tools/pmc_counters.py gpurun_out/pmc profiles/r02_counters.json
"""
import csv
import json
import os
import sys
from collections import defaultdict

root, out = sys.argv[1], sys.argv[2]
kernels = ("k_extend_walk", "k_extend_packet")  # the two closest-hit kernels of a step: queued rays, new paths
PASSES = ("sq_time", "sq_insts", "cache", "fetch", "write")


def bench_line(name):
    with open(os.path.join(root, name + ".log")) as f:
        for line in f:
            if line.startswith("{") and '"metric"' in line:
                return json.loads(line)
    raise SystemExit(f"no bench line in {name}.log")


def totals(name, only=None):
    tot = defaultdict(float)
    dur = {}
    with open(os.path.join(root, name, f"{name}_counter_collection.csv"), newline="") as f:
        for r in csv.DictReader(f):
            if any(k in r["Kernel_Name"] for k in (kernels if only is None else (only,))):
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
                dur[int(r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return tot, len(dur), sum(dur.values()) * 1e-9


c = {}
rays = {}
secs = {}
disp = {}
lines = {}
for p in PASSES:
    t, n, s = totals(p)
    b = bench_line(p)
    if b["warmup"] != 0:
        raise SystemExit("run the passes with --warmup 0 so the bench line counts every traced ray")
    lines[p] = b
    rays[p], secs[p], disp[p] = float(b["rays"]), s, n
    for k, v in t.items():
        c[k] = (v, p)


def per_ray(name):
    v, p = c[name]
    return v / rays[p]


def ratio(a, b):
    return c[a][0] / c[b][0] if c[b][0] else None


b0 = lines["sq_time"]
res = {
    "kernel": " + ".join(kernels),
    "command": "bench.py " + " ".join(sys.argv[3:]) + " under rocprofv3 --pmc, one run per pass (tools/pmc_passes.sh)",
    "workload": b0["config"]["workload"],
    "dispatches": disp["sq_time"],
    "rays": rays["sq_time"],
    "kernel_seconds_serialised": secs["sq_time"],
    "grays_per_s_serialised": rays["sq_time"] / secs["sq_time"] / 1e9,
    "valu_per_ray": per_ray("SQ_INSTS_VALU") ,
    "salu_per_ray": per_ray("SQ_INSTS_SALU"),
    "smem_per_ray": per_ray("SQ_INSTS_SMEM"),
    "vmem_rd_per_ray": per_ray("SQ_INSTS_VMEM_RD"),
    "lds_per_ray": per_ray("SQ_INSTS_LDS"),
    "note_per_ray": "wave-level instructions divided by rays: one wave instruction serves up to 64 rays",
    "lane_util": c["SQ_THREAD_CYCLES_VALU"][0] / (64.0 * c["SQ_ACTIVE_INST_VALU"][0]),
    "wait_share_of_wave_cycles": ratio("SQ_WAIT_ANY", "SQ_WAVE_CYCLES"),
    "valu_share_of_wave_cycles": ratio("SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES"),
    "l1_hit": 1.0 - c["TCP_TCC_READ_REQ_sum"][0] / c["TCP_TOTAL_CACHE_ACCESSES_sum"][0],
    "l2_hit": c["TCC_HIT_sum"][0] / (c["TCC_HIT_sum"][0] + c["TCC_MISS_sum"][0]),
    "l2_read_latency_cycles": ratio("TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCC_READ_REQ_sum"),
    "fetch_size_reported": c["FETCH_SIZE"][0],
    "write_size_reported": c["WRITE_SIZE"][0],
    "fetch_bytes_per_ray": 2.0 * c["FETCH_SIZE"][0] * 1024.0 / rays["fetch"],
    "write_bytes_per_ray": c["WRITE_SIZE"][0] * 1024.0 / rays["write"],
    "algorithmic_bytes_per_ray": b0["roofline"]["bytes_per_ray"],
    "note_traffic": "FETCH_SIZE doubled per the gfx950 correction for 128-byte requests; an upper bound for scattered 64-byte reads",
    "raw": {k: v for k, (v, _) in sorted(c.items())},
}
res["hbm_bytes_per_ray"] = res["fetch_bytes_per_ray"] + res["write_bytes_per_ray"]

# each kernel on its own: the camera-ray kernel traces the paths that start (bench line: "paths"), the ray-drawing
# kernel everything else (everything, where the camera rays have no kernel of their own)
per_kernel = {}
for kname in kernels:
    k = {}
    ok = True
    for p in PASSES:
        t, n, secs_k = totals(p, kname)
        if n == 0:
            ok = False
            break
        b = lines[p]
        paths = float(b.get("paths", b["rays"] / b["mean_path_length"]))
        has_packet = totals(p, "k_extend_packet")[1] > 0
        r_k = paths if kname == "k_extend_packet" else (float(b["rays"]) - paths if has_packet else float(b["rays"]))
        k[p] = (t, n, secs_k, r_k)
    if not ok:
        continue
    g = lambda p, name: k[p][0][name]
    per_kernel[kname] = {
        "dispatches": k["sq_time"][1], "rays": k["sq_time"][3], "kernel_seconds_serialised": k["sq_time"][2],
        "grays_per_s_serialised": k["sq_time"][3] / k["sq_time"][2] / 1e9,
        "valu_per_ray": g("sq_insts", "SQ_INSTS_VALU") / k["sq_insts"][3], "salu_per_ray": g("sq_insts", "SQ_INSTS_SALU") / k["sq_insts"][3],
        "vmem_rd_per_ray": g("sq_insts", "SQ_INSTS_VMEM_RD") / k["sq_insts"][3], "smem_per_ray": g("sq_insts", "SQ_INSTS_SMEM") / k["sq_insts"][3],
        "lds_per_ray": g("sq_insts", "SQ_INSTS_LDS") / k["sq_insts"][3],
        "lane_util": g("sq_time", "SQ_THREAD_CYCLES_VALU") / (64.0 * g("sq_time", "SQ_ACTIVE_INST_VALU")),
        "wait_share_of_wave_cycles": g("sq_time", "SQ_WAIT_ANY") / g("sq_time", "SQ_WAVE_CYCLES"),
        "l1_hit": 1.0 - g("cache", "TCP_TCC_READ_REQ_sum") / g("cache", "TCP_TOTAL_CACHE_ACCESSES_sum"),
        "l2_hit": g("cache", "TCC_HIT_sum") / (g("cache", "TCC_HIT_sum") + g("cache", "TCC_MISS_sum")),
        "l2_read_latency_cycles": g("cache", "TCP_TCC_READ_REQ_LATENCY_sum") / max(1.0, g("cache", "TCP_TCC_READ_REQ_sum")),
        "hbm_bytes_per_ray": 2.0 * g("fetch", "FETCH_SIZE") * 1024.0 / k["fetch"][3] + g("write", "WRITE_SIZE") * 1024.0 / k["write"][3],
    }
res["per_kernel"] = per_kernel
res["csrc_sha256"] = b0.get("csrc_sha256")
try:
    import subprocess
    res["git_commit"] = subprocess.check_output(["git", "-C", os.path.dirname(os.path.abspath(__file__)), "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    res["git_commit"] = None
with open(out, "w") as f:
    json.dump(res, f, indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "raw"}, indent=1))
