#!/bin/bash
# The counter passes of tools/pmc_passes.sh for the benchmark configurations named on the command line (default: all),
# each on bench.py's own command for that configuration: gpurun_out/pmc_<tag>/ and gpurun_out/r03_counters_<tag>.json,
# <tag> as bench.py's config_tag() names the workload (bench.py reads profiles/r03_counters_<tag>.json).
set -e
R="${GRAFT_REPO_ROOT:-$(pwd)}"
one() {
    tag=$1; shift
    bash "$R/tools/pmc_passes.sh" "$R/gpurun_out/pmc_$tag" "$@" > "$R/gpurun_out/pmc_$tag.log" 2>&1
    python3 "$R/tools/pmc_counters.py" "$R/gpurun_out/pmc_$tag" "$R/gpurun_out/r03_counters_$tag.json" "$@" > "$R/gpurun_out/r03_counters_$tag.log"
    python3 "$R/tools/pmc_summary.py" "$R/gpurun_out/pmc_$tag" > "$R/gpurun_out/r03_pmc_summary_$tag.txt"
    echo "$tag done"
}
want=" ${*:-knob_768 cornell_256 dragon_box_1024 interior_1920_p0.5 interior_1920_nee next_week_like_800_ext} "
case "$want" in *" knob_768 "*) one knob_768 --steps 2 --warmup 0 --no-cpu-baseline;; esac
case "$want" in *" cornell_256 "*) one cornell_256 --steps 4 --warmup 0 --no-cpu-baseline --scene cornell --width 256 --strata-per-step 64;; esac
case "$want" in *" dragon_box_1024 "*) one dragon_box_1024 --steps 2 --warmup 0 --no-cpu-baseline --scene dragon_box --width 1024;; esac
case "$want" in *" interior_1920_p0.5 "*) one interior_1920_p0.5 --steps 2 --warmup 0 --no-cpu-baseline --scene interior --width 1920 --strata-per-step 64 --bsdf-probability 0.5;; esac
case "$want" in *" interior_1920_nee "*) one interior_1920_nee --steps 2 --warmup 0 --no-cpu-baseline --scene interior --width 1920 --strata-per-step 64 --integrator nee;; esac
case "$want" in *" next_week_like_800_ext "*) one next_week_like_800_ext --steps 2 --warmup 0 --no-cpu-baseline --scene next_week_like --extensions --width 800;; esac
