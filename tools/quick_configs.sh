#!/bin/bash
# Rates of the benchmark configurations in short runs, one line each (no CPU baseline): for comparing two builds.
run() { name=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline "$@" > gpurun_out/q_$name.json 2> gpurun_out/q_$name.err || exit 1; python - "$name" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/q_{sys.argv[1]}.json").read().strip().splitlines()[-1])
pk = d["roofline"]["per_kernel"]
print(f"{sys.argv[1]:22s} {d['value']:8.0f} Mrays/s {d['spp_per_s']:9.1f} spp/s  packet {pk.get('k_extend_packet', {}).get('ms_per_launch', 0):.3f} ms  walk {pk.get('k_extend_walk', {}).get('ms_per_launch', 0):.3f} ms  extend {d['kernel_ms']['extend']:.0f} shade {d['kernel_ms']['shade']:.0f}")
PY
}
run c3_knob_768
run c1_cornell_256 --scene cornell --width 256 --strata-per-step 64 --steps 40 --warmup 3
run c2_next_week_800 --scene next_week_like --extensions --width 800 --steps 5 --warmup 2
run c4_dragon_1024 --scene dragon_box --width 1024 --steps 5 --warmup 2
run c5_interior_p05 --scene interior --width 1920 --strata-per-step 64 --bsdf-probability 0.5 --steps 5 --warmup 2
run c5_interior_nee --scene interior --width 1920 --strata-per-step 64 --integrator nee --steps 5 --warmup 2
