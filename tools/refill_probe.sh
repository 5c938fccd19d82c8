#!/bin/bash
# PT_OPT_REFILL_BELOW x PT_OPT_LEAF_BATCH of the walking kernel: SCENE, W and BENCH_ARGS choose the workload
mkdir -p gpurun_out
SCENE=${SCENE:-dragon_box}; W=${W:-1024}
CFGS=${CFGS:-32_16 40_16 48_16 32_24 40_24 32_12 24_16}
for cfg in $CFGS; do
  a=${cfg%%_*}; b=${cfg##*_}
  timeout -k 10 200 python bench.py --no-cpu-baseline --scene $SCENE --width $W --refill-below $a --leaf-batch $b $BENCH_ARGS > gpurun_out/rf_${a}_$b.json 2> gpurun_out/rf_${a}_$b.err || { echo "$cfg failed"; tail -3 gpurun_out/rf_${a}_$b.err; exit 1; }
  python3 - "$a" "$b" "$SCENE" <<'PY'
import json,sys
a,b,sc=sys.argv[1:4]
d=json.loads(open(f"gpurun_out/rf_{a}_{b}.json").read().strip().splitlines()[-1])
print(sc,a,b, round(d["value"]), "Mrays/s", {k:round(v,1) for k,v in d["kernel_ms"].items()}, flush=True)
PY
done
