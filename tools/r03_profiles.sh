#!/bin/bash
# What profiles/ holds for the final build besides the counter files (tools/r03_counters.sh), collected on the GPU box:
# the default bench line, rocprofv3 kernel trace + stats of the same command, one bench line per benchmark
# configuration, the whole knob render.  Usage: bash tools/r03_profiles.sh   (writes gpurun_out/r03/)
set -e
R="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$R/gpurun_out/r03"
mkdir -p "$OUT"
timeout -k 10 400 python3 "$R/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench done"
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o knob -- python3 "$R/bench.py" --no-cpu-baseline > "$OUT/bench_under_rocprofv3.json" 2> "$OUT/trace.err")
find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/knob_kernel_stats.csv" \;
find "$OUT/trace" -name "*kernel_trace.csv" -delete
echo "trace done"
cd "$R"
bash tools/all_configs.sh
cp gpurun_out/all_configs.json "$OUT/all_configs.json"
echo "configs done"
timeout -k 10 600 python3 tools/c3_full.py "$OUT/c3_full.json" > "$OUT/c3_full.log" 2>&1
echo "full render done"
