#!/bin/bash
# bench.py over builds of the core library x PT_PACKET_PER_CU: arguments are build:per_cu pairs
mkdir -p gpurun_out
for vp in "$@"; do
  v=${vp%%:*}; pc=${vp##*:}
  PT_PACKET_PER_CU=$pc PTCORE_LIB=$PWD/pathtracer_amd/libptcore_$v.so timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 $BENCH_ARGS > gpurun_out/var_${v}_$pc.json 2> gpurun_out/var_${v}_$pc.err || { echo "$vp failed"; tail -3 gpurun_out/var_${v}_$pc.err; exit 1; }
  python3 - "${v}_$pc" <<'PY'
import json,sys
n=sys.argv[1]
d=json.loads(open(f"gpurun_out/var_{n}.json").read().strip().splitlines()[-1])
print(n, round(d["value"]), "Mrays/s", {k:round(v,1) for k,v in d["kernel_ms"].items()}, flush=True)
PY
done
