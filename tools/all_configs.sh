#!/bin/bash
# The five benchmark configurations at their own sizes, one bench.py line each into gpurun_out/cfg_<name>.json.
set -e
mkdir -p gpurun_out
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" > gpurun_out/cfg_$name.json 2> gpurun_out/cfg_$name.err; echo "$name done"; }
run c1_cornell_256 --scene cornell --width 256 --strata-per-step 64 --steps 40 --warmup 3
run c2_next_week_800 --scene next_week_like --extensions --width 800 --steps 5 --warmup 2
run c3_knob_768 --scene knob --width 768 --steps 20 --warmup 3
run c4_dragon_1024 --scene dragon_box --width 1024 --steps 5 --warmup 2
run c5_interior_1920_p05 --scene interior --width 1920 --strata-per-step 64 --bsdf-probability 0.5 --steps 5 --warmup 2
run c5_interior_1920_nee --scene interior --width 1920 --strata-per-step 64 --integrator nee --steps 5 --warmup 2
python3 - <<'PY'
import glob, json
rows = {}
for f in sorted(glob.glob("gpurun_out/cfg_*.json")):
    with open(f) as fh:
        rows[f.split("cfg_")[1][:-5]] = json.loads(fh.read().strip().splitlines()[-1])
with open("gpurun_out/all_configs.json", "w") as fh:
    json.dump(rows, fh, indent=1)
PY
