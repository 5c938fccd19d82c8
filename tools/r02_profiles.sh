#!/bin/bash
# Everything the profiles/ directory holds for the default bench command, collected on the GPU box:
#   bench line, rocprofv3 kernel trace + stats of the same command, the counter passes.
# Usage: bash tools/r02_profiles.sh <out_dir under gpurun_out>
set -e
R="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$R/${1:-gpurun_out/prof}"
mkdir -p "$OUT"
timeout -k 10 300 python3 "$R/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench done"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o knob -- python3 "$R/bench.py" --no-cpu-baseline > "$OUT/bench_under_rocprofv3.json" 2> "$OUT/trace.err"
echo "trace done"
cd "$R"
bash tools/pmc_passes.sh "$OUT/pmc"
python3 tools/pmc_counters.py "$OUT/pmc" "$OUT/counters.json" --steps 2 --warmup 0 --no-cpu-baseline > "$OUT/counters.log"
python3 tools/pmc_summary.py "$OUT/pmc" > "$OUT/pmc_summary.txt"
echo "counters done"
