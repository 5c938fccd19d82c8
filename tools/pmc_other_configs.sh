#!/bin/bash
# The counter passes of tools/pmc_passes.sh on the configurations other than the headline one:
#   gpurun_out/pmc_<name>/ and gpurun_out/counters_<name>.json
set -e
R="${GRAFT_REPO_ROOT:-$(pwd)}"
one() {
    name=$1; shift
    bash "$R/tools/pmc_passes.sh" "$R/gpurun_out/pmc_$name" "$@" > "$R/gpurun_out/pmc_$name.log" 2>&1
    python3 "$R/tools/pmc_counters.py" "$R/gpurun_out/pmc_$name" "$R/gpurun_out/counters_$name.json" "$@" > "$R/gpurun_out/counters_$name.log"
    echo "$name done"
}
one c1_cornell --steps 4 --warmup 0 --no-cpu-baseline --scene cornell --width 256 --strata-per-step 64
one c4_dragon_box --steps 2 --warmup 0 --no-cpu-baseline --scene dragon_box --width 1024
one c5_interior --steps 2 --warmup 0 --no-cpu-baseline --scene interior --width 1920 --strata-per-step 64 --bsdf-probability 0.5
