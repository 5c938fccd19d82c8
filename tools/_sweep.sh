mkdir -p gpurun_out/r3b
run() { name=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline "$@" > gpurun_out/r3b/$name.json 2> gpurun_out/r3b/$name.err || exit 1; }
run c3_t0 --batch-ticks 0 &&
run c3_t24 --batch-ticks 24 &&
run c3_t16 --batch-ticks 16 &&
run c3_t12 --batch-ticks 12 &&
run c3_t8 --batch-ticks 8 &&
run c3_t6 --batch-ticks 6 &&
run c2_t0 --scene next_week_like --extensions --width 800 --batch-ticks 0 &&
run c2_t16 --scene next_week_like --extensions --width 800 --batch-ticks 16 &&
run c2_t10 --scene next_week_like --extensions --width 800 --batch-ticks 10 &&
run c5_t0 --scene interior --width 1920 --strata-per-step 64 --bsdf-probability 0.5 --batch-ticks 0 &&
run c5_t10 --scene interior --width 1920 --strata-per-step 64 --bsdf-probability 0.5 --batch-ticks 10 &&
run c3_t12_16M --batch-ticks 12 --rolling 16000000 &&
run c3_t12_16M_f3 --batch-ticks 12 --rolling 16000000 --live-factor 3 &&
run c3_t12_20M_f3 --batch-ticks 12 --rolling 20000000 --live-factor 3
