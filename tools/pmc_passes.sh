#!/bin/bash
# Hardware-counter passes over one short bench run, one rocprofv3 run per pass (the counters of a
# block do not split over passes).  Run on the GPU box from the repository root:
#     bash tools/pmc_passes.sh [out_dir] [bench args...]
# Output: <out_dir>/<pass>/..._counter_collection.csv per pass.
set -e
R="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="${1:-$R/gpurun_out/pmc}"
shift || true
ARGS="${@:---steps 2 --warmup 0 --no-cpu-baseline}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pass() {
    name="$1"; shift
    case " ${PASSES:-sq_time sq_insts cache fetch write} " in *" $name "*) ;; *) return 0;; esac
    echo "== pass $name: $*"
    timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o "$name" -- python3 "$R/bench.py" $ARGS > "$OUT/$name.log" 2>&1
    echo "   done"
}
pass sq_time SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU
pass sq_insts SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM
pass cache TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
