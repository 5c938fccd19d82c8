#!/bin/bash
# PT_OPT_REFILL_BELOW x PT_OPT_LEAF_BATCH of the walking kernel on a scene whose rays are mostly bounce rays
mkdir -p gpurun_out
SCENE=${SCENE:-dragon_box}; W=${W:-1024}
for cfg in "32 16" "40 16" "48 16" "56 16" "48 24" "40 24" "32 24" "48 12"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --no-cpu-baseline --scene $SCENE --width $W --refill-below $1 --leaf-batch $2 $BENCH_ARGS > gpurun_out/rf_$1_$2.json 2> gpurun_out/rf_$1_$2.err || { echo "$cfg failed"; tail -3 gpurun_out/rf_$1_$2.err; exit 1; }
  python3 - "$1" "$2" <<'PY'
import json,sys
a,b=sys.argv[1:3]
d=json.loads(open(f"gpurun_out/rf_{a}_{b}.json").read().strip().splitlines()[-1])
print(a,b, round(d["value"]), "Mrays/s", {k:round(v,1) for k,v in d["kernel_ms"].items()}, flush=True)
PY
done
