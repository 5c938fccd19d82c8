mkdir -p gpurun_out/r3p
run() { name=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline "$@" > gpurun_out/r3p/$name.json 2> gpurun_out/r3p/$name.err || exit 1; }
run c3 --steps 10 &&
run c3_r32 --steps 10 --rolling 32000000 &&
PT_PACKET_BLOCKS=32 run c3_p32 --steps 10 &&
run c4 --scene dragon_box --width 1024 --steps 5 &&
run c1 --scene cornell --width 256 --strata-per-step 64 --steps 20 --warmup 3 &&
run c5 --scene interior --width 1920 --strata-per-step 64 --bsdf-probability 0.5 --steps 5 &&
run c5n --scene interior --width 1920 --strata-per-step 64 --integrator nee --steps 5 &&
run c2 --scene next_week_like --extensions --width 800 --steps 5
