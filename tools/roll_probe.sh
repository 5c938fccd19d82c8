cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "12000000 50" "4000000 50" "12000000 8" "0 50"; do set -- $cfg; echo "== rolling $1 depth $2"; ROLLING=$1 timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/probe_$1_$2 -o p -- python3 $R/tools/depth_cost.py $2 2>&1 | grep depth; grep -E "k_shade|k_extend_walk|k_raygen|k_acc" $R/gpurun_out/probe_$1_$2/p_kernel_stats.csv | cut -d, -f1-4 | sed 's/(ptd[^"]*"/"/' ; done
