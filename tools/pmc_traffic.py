#!/usr/bin/env python3
"""HBM traffic of the closest-hit kernel per ray, from the `fetch` and `write` passes of tools/pmc_passes.sh.

FETCH_SIZE and WRITE_SIZE are in KiB-like units of 1024 bytes per count as rocprofv3 reports them; on gfx950
FETCH_SIZE tallies a 128-byte request as 64 bytes for wide streaming reads, so it is doubled here, which makes
the figure an upper bound for this kernel's scattered 64-byte node fetches.

    python tools/pmc_traffic.py gpurun_out/pmc profiles/r01_traffic.json
"""
import csv, json, os, sys

root, out = sys.argv[1], sys.argv[2]
kernel = "k_extend_walk"


def total(pass_name, counter):
    s = 0.0
    n = 0
    with open(os.path.join(root, pass_name, f"{pass_name}_counter_collection.csv"), newline="") as f:
        for r in csv.DictReader(f):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
                s += float(r["Counter_Value"])
                n += 1
    return s, n


def bench_line(pass_name):
    with open(os.path.join(root, pass_name + ".log")) as f:
        for line in f:
            if line.startswith("{") and '"metric"' in line:
                return json.loads(line)
    raise SystemExit(f"no bench line in {pass_name}.log")


fetch_kb, n_disp = total("fetch", "FETCH_SIZE")
write_kb, _ = total("write", "WRITE_SIZE")
b = bench_line("fetch")
rays_timed = b["value"] * 1e6 * b["ms_per_step"] * b["steps"] / 1e3
rays = rays_timed * (b["steps"] + b["warmup"]) / b["steps"]  # warm-up steps run the same kernel on as many strata
res = {
    "kernel": kernel,
    "command": f"bench.py --steps {b['steps']} --warmup {b['warmup']} --no-cpu-baseline under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate runs)",
    "dispatches": n_disp,
    "rays": rays,
    "fetch_size_kb_reported": fetch_kb,
    "write_size_kb_reported": write_kb,
    "fetch_bytes_per_ray": 2.0 * fetch_kb * 1024.0 / rays,
    "write_bytes_per_ray": write_kb * 1024.0 / rays,
    "hbm_bytes_per_ray": (2.0 * fetch_kb + write_kb) * 1024.0 / rays,
    "algorithmic_bytes_per_ray": b["roofline"]["bytes_per_ray"],
    "note": "FETCH_SIZE doubled per the gfx950 correction for 128-byte requests; uncalibrated for scattered 64-byte reads, so an upper bound",
}
with open(out, "w") as f:
    json.dump(res, f, indent=1)
print(json.dumps(res, indent=1))
