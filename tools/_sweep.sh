mkdir -p gpurun_out/r3i
run() { name=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 "$@" > gpurun_out/r3i/$name.json 2> gpurun_out/r3i/$name.err || exit 1; }
run r24_l8 &&
run r16_l8 --refill-below 16 &&
run r32_l8 --refill-below 32 &&
run r40_l8 --refill-below 40 &&
run r48_l8 --refill-below 48 &&
run r32_l4 --refill-below 32 --leaf-batch 4 &&
run r32_l16 --refill-below 32 --leaf-batch 16 &&
run r24_l4 --leaf-batch 4 &&
run r24_l16 --leaf-batch 16 &&
run r40_l16 --refill-below 40 --leaf-batch 16
