#!/bin/bash
# bench.py (knob, 1 M-triangle box) with other builds of libptcore.so: bash tools/ab_variants.sh <tag> <lib> [<lib> ...]
tag=$1; shift
for lib in "$@"; do
  name=$(basename $lib .so)
  PTCORE_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 20 --no-cpu-baseline > gpurun_out/${tag}_${name}_knob.json 2>gpurun_out/${tag}_${name}_knob.err
  PTCORE_LIB=$PWD/$lib timeout -k 10 200 python bench.py --scene dragon_box --width 1024 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_${name}_dragon.json 2>gpurun_out/${tag}_${name}_dragon.err
done
python3 - "$tag" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(f"gpurun_out/{sys.argv[1]}_*.json")):
    try:
        b = json.loads(open(f).read().strip().splitlines()[-1]); pk = b["roofline"]["per_kernel"]
        print(f, round(b["value"]), {k: (round(v["Grays_per_s"], 2), round(v["ms_per_launch"], 2)) for k, v in pk.items()}, {k: round(v) for k, v in b["kernel_ms"].items()})
    except Exception as e:
        print(f, "ERR", e)
PY
