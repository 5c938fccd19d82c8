#!/bin/bash
# ab_all.sh with the device bytes beside the rate
tag=$1; shift
run() { cfg=$1; shift; for lib in $LIBS; do name=$(basename $lib .so); PTCORE_LIB=$PWD/$lib timeout -k 10 200 python bench.py --no-cpu-baseline "$@" > gpurun_out/${tag}_${cfg}_${name}.json 2> gpurun_out/${tag}_${cfg}_${name}.err || { echo "$cfg $name failed"; tail -3 gpurun_out/${tag}_${cfg}_${name}.err; exit 1; }
python3 - gpurun_out/${tag}_${cfg}_${name}.json $cfg $name <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:18s} {sys.argv[3]:24s} {d['value']:8.0f} Mrays/s  {d['device_bytes']/1e9:6.1f} GB  host wait {d['host_wait_ms']:.0f} ms  extend {d['kernel_ms']['extend']:.0f} shade {d['kernel_ms']['shade']:.0f}", flush=True)
PY
done; }
LIBS="$@"
run knob --steps 20
run cornell --scene cornell --width 256 --strata-per-step 64 --steps 40 --warmup 3
run next_week --scene next_week_like --extensions --width 800 --steps 5 --warmup 2
run dragon --scene dragon_box --width 1024 --steps 5 --warmup 2
run interior_p05 --scene interior --width 1920 --strata-per-step 64 --bsdf-probability 0.5 --steps 5 --warmup 2
run interior_nee --scene interior --width 1920 --strata-per-step 64 --integrator nee --steps 5 --warmup 2
