#!/bin/bash
# One pipe against two by the number of paths a launch starts (bench.py --pipes 1 / 2 on small stratum ranges): where a second
# pipe of the rolling schedule pays.  Prints one line per run.
run() { label=$1; shift
for pipes in 1 2; do
timeout -k 10 200 python bench.py --no-cpu-baseline --pipes $pipes "$@" > gpurun_out/c1p.json 2>/dev/null
python3 -c "
import json,sys; d=json.loads(open('gpurun_out/c1p.json').read().strip().splitlines()[-1]); print('probe', sys.argv[1], 'pipes', sys.argv[2], round(d['value']), round(d['device_bytes']/1e9,1), 'GB')" "$label" $pipes
done; }
run "cornell256x64(4.2M)" --scene cornell --width 256 --strata-per-step 64 --steps 40 --warmup 3
run "cornell256x128(8.4M)" --scene cornell --width 256 --strata-per-step 128 --steps 20 --warmup 3
run "cornell256x256(16.8M)" --scene cornell --width 256 --strata-per-step 256 --steps 10 --warmup 3
run "knob768x8(4.7M)" --scene knob --strata-per-step 8 --steps 100 --warmup 5
run "knob768x16(9.4M)" --scene knob --strata-per-step 16 --steps 60 --warmup 5
run "knob768x32(18.9M)" --scene knob --strata-per-step 32 --steps 40 --warmup 5
run "dragon1024x8(8.4M)" --scene dragon_box --width 1024 --strata-per-step 8 --steps 20 --warmup 3
run "interior1920x4(8.3M)" --scene interior --width 1920 --strata-per-step 4 --bsdf-probability 0.5 --steps 20 --warmup 3
