#!/bin/bash
# bench.py over several builds of the core library (PTCORE_LIB), knob scene; extra arguments go to bench.py
mkdir -p gpurun_out
for v in "$@"; do
  PTCORE_LIB=$PWD/pathtracer_amd/libptcore_$v.so timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 $BENCH_ARGS > gpurun_out/var_$v.json 2> gpurun_out/var_$v.err || { echo "$v failed"; tail -3 gpurun_out/var_$v.err; exit 1; }
  python3 - "$v" <<'PY'
import json,sys
n=sys.argv[1]
d=json.loads(open(f"gpurun_out/var_{n}.json").read().strip().splitlines()[-1])
print(n, round(d["value"]), "Mrays/s", {k:round(v,1) for k,v in d["kernel_ms"].items()}, flush=True)
PY
done
