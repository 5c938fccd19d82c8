#!/bin/bash
# Two rolling pipes against one, at several sizes and walking-kernel grids (knob scene): one bench.py line each.
mkdir -p gpurun_out
run() { name=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 "$@" > gpurun_out/pp_$name.json 2> gpurun_out/pp_$name.err || { echo "$name failed"; exit 1; }; 
python3 - "$name" <<'PY'
import json,sys
n=sys.argv[1]
d=json.loads(open(f"gpurun_out/pp_{n}.json").read().strip().splitlines()[-1])
print(n, round(d["value"]), "Mrays/s", {k:round(v,1) for k,v in d["kernel_ms"].items()}, round(d["device_bytes"]/1e9,1),"GB", flush=True)
PY
}
run p1_32 --pipes 1
run p2_16 --pipes 2 --rolling 16000000
run p2_24 --pipes 2 --rolling 24000000
run p2_32 --pipes 2 --rolling 32000000
run p2_16_b4 --pipes 2 --rolling 16000000 --blocks-per-cu 4
run p2_16_b3 --pipes 2 --rolling 16000000 --blocks-per-cu 3
run p3_12 --pipes 3 --rolling 12000000
