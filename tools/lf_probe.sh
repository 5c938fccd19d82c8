#!/bin/bash
run() { echo "# $*"; timeout -k 10 200 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %.0f Mrays/s  %.2f GB host_wait %.0f ms of %.0f' % (d['value'], d['device_bytes']/1e9, d['host_wait_ms'], d['ms_per_step']*d['steps']))"; }
for lf in 4 5 6; do run --scene cornell --width 256 --strata-per-step 64 --steps 40 --live-factor $lf; done
for lf in 4 5 6; do run --scene dragon_box --width 1024 --steps 5 --warmup 2 --live-factor $lf; done
for lf in 4 5 6; do run --scene interior --width 1920 --strata-per-step 64 --bsdf-probability 0.5 --steps 5 --warmup 2 --live-factor $lf; done
for lf in 4 5; do run --scene next_week_like --extensions --width 800 --steps 5 --warmup 2 --live-factor $lf; done
