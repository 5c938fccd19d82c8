#!/bin/bash
# Rate and device bytes against the two knobs that size the rolling schedule's state: paths started per launch
# (--rolling) and the room for rays in flight as a multiple of it (--live-factor).  One JSON line per run.
out=gpurun_out/state_size_probe.jsonl
: > $out
run() { echo "# $*" >> $out; timeout -k 10 150 python bench.py --no-cpu-baseline "$@" >> $out 2>> gpurun_out/state_size_probe.err || exit 1; }
for lf in 0 3 2.5 2; do run --live-factor $lf; done
for rp in 24000000 16000000; do for lf in 0 2.5; do run --rolling $rp --live-factor $lf; done; done
for lf in 0 3 2.5; do run --scene interior --width 1920 --strata-per-step 64 --bsdf-probability 0.5 --steps 5 --warmup 2 --live-factor $lf; done
for lf in 0 3 2.5; do run --scene cornell --width 256 --strata-per-step 64 --steps 40 --live-factor $lf; done
for lf in 0 3 2.5; do run --scene dragon_box --width 1024 --steps 5 --warmup 2 --live-factor $lf; done
python - <<'PY'
import json
for line in open('gpurun_out/state_size_probe.jsonl'):
    if line.startswith('#'): print(line.strip()); continue
    d = json.loads(line)
    print(f"   {d['value']:.0f} Mrays/s  {d.get('spp_per_s', 0):.1f} spp/s  {(d.get('device_bytes') or d.get('config', {}).get('device_bytes') or 0) / 1e9:.2f} GB")
PY
